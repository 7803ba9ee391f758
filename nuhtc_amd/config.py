"""Config surface of the reference kept as is: the standalone mmcv-style Python config files
(configs/nuhtc/htc_lite_swin_pytorch_fpn_*_seasaw_CAS.py) are exec'd and read as attribute dicts
(the reference: mmcv `Config.fromfile` + `patch_config`, tools/infer.py:43-46, nuhtc/utils/patch.py:69-81).
Only the keys the tile-inference path reads are interpreted; unsupported model types are rejected loudly."""
import copy
import inspect
import os


class ConfigDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _wrap(o):
    if isinstance(o, dict):
        return ConfigDict({k: _wrap(v) for k, v in o.items()})
    if isinstance(o, (list, tuple)):
        return type(o)(_wrap(v) for v in o)
    return o


class Config(ConfigDict):
    @staticmethod
    def fromfile(path):
        ns = {'__file__': os.path.abspath(path)}
        with open(path) as f:
            exec(compile(f.read(), path, 'exec'), ns)
        cfg = Config(_wrap({k: v for k, v in ns.items() if not k.startswith('__') and not inspect.ismodule(v) and not callable(v)}))
        cfg['filename'] = path
        return cfg

    def merge_from_dict(self, options):
        for key, v in (options or {}).items():
            d = self
            parts = key.split('.')
            for p in parts[:-1]:
                d = d[int(p)] if isinstance(d, (list, tuple)) else d.setdefault(p, ConfigDict())
            if isinstance(d, list):
                d[int(parts[-1])] = v
            else:
                d[parts[-1]] = v


def patch_config(cfg):
    """nuhtc/utils/patch.py:69-81 resolves `${a.b}` strings and sets cfg_name; the htc_lite configs contain no `${}`."""
    cfg = copy.deepcopy(cfg)
    if 'filename' in cfg:
        cfg['cfg_name'] = os.path.splitext(os.path.basename(cfg['filename']))[0]
    return cfg


def set_test_scale_factor(cfg, mag):
    """tools/infer_wsi.py:416-419 (same in infer_patch.py): every MultiScaleFlipAug step of the test pipeline gets
    `scale_factor = float(80 / mag)` (40x -> 2.0, 20x -> 4.0)."""
    sf = float(80 / mag)
    pipes = []
    if 'data' in cfg and 'test' in cfg.data and 'pipeline' in cfg.data.test:
        pipes.append(cfg.data.test.pipeline)
    if 'test_pipeline' in cfg:
        pipes.append(cfg.test_pipeline)
    for pipe in pipes:
        for step in pipe:
            if step.type == 'MultiScaleFlipAug':
                step['scale_factor'] = sf
    return sf


def _expect(cond, what):
    if not cond:
        raise ValueError(f'unsupported config for the MI355X htc_lite_swin engine: {what}')


def engine_options(cfg):
    """Translate `cfg.model` / `cfg.data.test.pipeline` into nuhtc_config fields (include/nuhtc_hip.h)."""
    m = cfg.model
    _expect(m.type == 'HybridTaskCascade_Cus', f'model.type={m.type}')
    bb = m.backbone
    _expect(bb.type == 'SwinTransformer' and bb.embed_dims == 96 and list(bb.depths) == [2, 2, 6, 2] and
            list(bb.num_heads) == [3, 6, 12, 24] and bb.window_size == 7 and bb.get('mlp_ratio', 4) == 4, 'backbone must be Swin-T (96, [2,2,6,2], window 7)')
    nk = m.neck
    _expect(nk.type == 'FPN' and list(nk.in_channels) == [96, 192, 384, 768] and nk.out_channels == 64 and nk.num_outs == 4, f'neck={dict(nk)}')
    rp = m.rpn_head
    ag = rp.anchor_generator
    _expect(rp.type == 'RPNHead' and list(ag.scales) == [4] and list(ag.ratios) == [0.5, 1.0, 2.0] and list(ag.strides) == [4, 8, 16, 32], 'rpn_head / anchor_generator')
    _expect(list(rp.bbox_coder.target_stds) == [1.0, 1.0, 1.0, 1.0] and list(rp.bbox_coder.target_means) == [0.0] * 4, 'rpn bbox_coder')
    rh = m.roi_head
    _expect(rh.type == 'HybridTaskCascadeRoIHead_Lite' and rh.num_stages == 3, f'roi_head.type={rh.type}')
    ex = rh.bbox_roi_extractor
    _expect(ex.type == 'AttentionRoIExtractor' and ex.start_level == 2 and ex.roi_layer.output_size == 7 and ex.roi_layer.sampling_ratio == 2, 'bbox_roi_extractor')
    mx = rh.mask_roi_extractor
    _expect(mx.type == 'AttentionRoIExtractor' and mx.roi_layer.output_size == 14 and mx.roi_layer.sampling_ratio == 0, 'mask_roi_extractor')
    heads = rh.bbox_head
    _expect(len(heads) == 3 and all(h.type == 'Shared2FCBBoxHeadWithProb' and h.reg_class_agnostic and h.fc_out_channels == 256 and
                                    h.cls_predictor_cfg.type == 'NormedLinear' and h.cls_predictor_cfg.get('tempearture', 20) == 20 for h in heads), 'bbox_head')
    mh = rh.mask_head[0] if isinstance(rh.mask_head, (list, tuple)) else rh.mask_head
    _expect(mh.type == 'HTCMaskHead' and mh.class_agnostic and mh.num_convs == 4, 'mask_head')
    _expect(rh.semantic_head.type == 'FusedSemanticHead' and rh.semantic_head.num_classes == 1 and rh.semantic_head.fusion_level == 0, 'semantic_head')
    t = m.test_cfg
    opts = dict(
        num_classes=int(heads[0].num_classes),
        rpn_nms_pre=int(t.rpn.nms_pre), rpn_max_per_img=int(t.rpn.max_per_img), rpn_nms_iou=float(t.rpn.nms.iou_threshold),
        rpn_min_bbox_size=float(t.rpn.min_bbox_size),
        score_thr=float(t.rcnn.score_thr), nms_iou=float(t.rcnn.nms.iou_threshold), max_per_img=int(t.rcnn.max_per_img),
        mask_thr_binary=float(t.rcnn.mask_thr_binary),
        att_thres=float(ex.thres), watershed_proposal=int(bool(rh.get('watershed_proposal', True))),
        stage_stds=[[float(v) for v in h.bbox_coder.target_stds] for h in heads],
    )
    pipe = None
    if 'data' in cfg and 'test' in cfg.data and 'pipeline' in cfg.data.test:
        pipe = cfg.data.test.pipeline
    elif 'test_pipeline' in cfg:
        pipe = cfg.test_pipeline
    if pipe is not None:
        for step in pipe:
            if step.type == 'MultiScaleFlipAug':
                opts['scale_factor'] = float(step.get('scale_factor', 1.0))
                _expect(not step.get('flip', False), 'flip TTA')
                for tr in step.transforms:
                    if tr.type == 'Normalize':
                        opts['mean'] = [float(v) for v in tr.mean]
                        opts['std'] = [float(v) for v in tr.std]
                        _expect(tr.get('to_rgb', True), 'Normalize(to_rgb=False)')
    return opts
