"""Drop-in mirror of the reference's inference API:

    init_detector(config, checkpoint=None, device='cuda:0', cfg_options=None)   nuhtc/apis/inference.py:11-57
    inference_detector(model, imgs)                                             mmdet/apis/inference.py:90-153

Same names, argument meaning and result format — `(bbox_results, segm_results)` per image with per-class (k,5) float32
arrays in original-tile pixels and per-class lists of (H,W) bool masks.  Everything between the uint8 pixels and
those results runs in libnuhtc_hip.so on the MI355X; there is no CPU path (device='cpu' is an error).
"""
import warnings

import numpy as np

from . import hip, weights
from .config import Config, engine_options, patch_config


class Detector:
    """Stands in for the nn.Module `init_detector` returns: callers set `.CLASSES` and read `.cfg`
    (tools/infer.py:49, tools/infer_wsi.py:424-428)."""

    def __init__(self, cfg, state_dict, device, max_batch=16, max_cc_proposals=512, bind_host=None):
        self.cfg = cfg
        self.bind_host = bind_host
        self.state_dict = state_dict
        self.device = device
        self.max_batch = max_batch
        self.max_cc_proposals = max_cc_proposals
        self.opts = engine_options(cfg)
        self.CLASSES = tuple(str(i) for i in range(self.opts['num_classes']))
        self._engines = {}          # (tile size[, depth]) -> engine, least recently used first
        self.max_engines = 4        # each holds the weights + a workspace of ~0.3 GB per tile of max_batch

    def _cached(self, key, make):
        if key in self._engines:
            self._engines[key] = self._engines.pop(key)         # move to the most-recent end
            return self._engines[key]
        while len(self._engines) >= self.max_engines:           # evict the least recently used size
            old = self._engines.pop(next(iter(self._engines)))
            old.close()
        self._engines[key] = make()
        return self._engines[key]

    def engine(self, tile_hw):
        from .engine import Engine
        key = (int(tile_hw[0]), int(tile_hw[1]))

        def make():
            opts = dict(self.opts)
            nc = opts.pop('num_classes')
            return Engine(self.state_dict, device=self.device, max_batch=self.max_batch, tile=key, num_classes=nc,
                          max_cc_proposals=self.max_cc_proposals, bind_host=self.bind_host, **opts)
        return self._cached(key, make)

    def pipeline(self, tile_hw, depth=4):
        """`depth` engines on their own streams for the streaming (WSI) path: nuhtc_amd.pipeline.EnginePipeline."""
        from .pipeline import EnginePipeline
        key = (int(tile_hw[0]), int(tile_hw[1]), int(depth))

        def make():
            opts = dict(self.opts)
            nc = opts.pop('num_classes')
            return EnginePipeline(self.state_dict, device=self.device, depth=depth, max_batch=self.max_batch, tile=key[:2],
                                  num_classes=nc, max_cc_proposals=self.max_cc_proposals, bind_host=self.bind_host, **opts)
        return self._cached(key, make)

    def eval(self):
        return self


def _device_index(device):
    d = str(device)
    if d == 'cpu' or d.startswith('cpu'):
        raise ValueError("nuhtc_amd has no CPU path: device must be 'cuda:N' (an MI355X). The CPU oracle under oracle/ is test infrastructure only.")
    if d == 'cuda':
        return 0
    if d.startswith('cuda:'):
        return int(d.split(':')[1])
    raise ValueError(f'unsupported device {device!r}')


def init_detector(config, checkpoint=None, device='cuda:0', cfg_options=None, max_batch=16, bind_host=None, att_pool_fp16=0):
    """nuhtc/apis/inference.py:11-57.  Beyond the reference's arguments: `max_batch` (capacity of the engines the detector creates) and
    `bind_host` -- True places the thread that creates an engine on the CPUs of the GPU's NUMA node (nuhtc_bind_host_thread: worth
    8 % on the dense launches of a two-socket host; the thread's mask is restored when the engine is closed).  A
    library does not change its caller's CPU affinity unasked: the default (None) follows NUHTC_HOST_AFFINITY, which defaults to off;
    the entry points that own their process (bench.py, tools/infer_wsi.py, tools/bench_wsi.py) switch it on.
    `att_pool_fp16=1`: the attention-pool branch of the RoI extractor in the fp16 arithmetic the reference uses when its maps are on a CUDA
    device (roi_extractors_cus.py:203,231); the default is the fp32 arithmetic of its CPU path."""
    if isinstance(config, str):
        config = Config.fromfile(config)
    elif not isinstance(config, dict):
        raise TypeError(f'config must be a filename or Config object, but got {type(config)}')
    if cfg_options is not None:
        config.merge_from_dict(cfg_options)
    config = patch_config(config)
    if 'pretrained' in config.model:
        config.model.pretrained = None
    config.model.train_cfg = None
    dev = _device_index(device)
    opts = engine_options(config)   # validates the model description before touching the GPU
    classes = None
    if checkpoint is not None:
        sd, meta = weights.load_checkpoint(checkpoint, opts['num_classes'], return_meta=True)
        if 'CLASSES' in meta:                       # nuhtc/apis/inference.py:45-46
            classes = tuple(meta['CLASSES'])
        else:
            # :47-53: the reference falls back to the 80 COCO names here (wrong for every nuclei model; both tools overwrite
            # CLASSES right after, tools/infer.py:49, tools/infer_wsi.py:425); class indices are kept instead
            warnings.warn("Class names are not saved in the checkpoint's meta data, use the class indices '0'..'N-1' by default.")
    else:
        warnings.warn('init_detector called without a checkpoint: using seeded synthetic weights (the reference would keep its random init)')
        sd = weights.seeded_state_dict(0, opts['num_classes'])
    model = Detector(config, sd, dev, max_batch=max_batch, bind_host=bind_host)
    if att_pool_fp16:
        model.opts['att_pool_fp16'] = 1
    if classes is not None:
        model.CLASSES = classes
    return model


def _load_image_rgb(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert('RGB'))


def inference_detector(model, imgs):
    """imgs: str | ndarray | list of either.  File paths follow tools/infer.py (mmcv.imread BGR -> to_rgb: the network
    sees true RGB); ndarrays follow the LoadImageFromWebcam branch (mmdet/apis/inference.py:112-115): the array is
    taken as BGR and channel-swapped before normalisation, whatever it really holds."""
    is_batch = isinstance(imgs, (list, tuple))
    if not is_batch:
        imgs = [imgs]
    if len(imgs) == 0:
        return []
    if isinstance(imgs[0], np.ndarray):
        arrs, mode = [np.asarray(i) for i in imgs], hip.CH_SWAP
    else:
        arrs, mode = [_load_image_rgb(p) for p in imgs], hip.CH_AS_IS
    for a in arrs:
        if a.ndim != 3 or a.shape[2] != 3 or a.dtype != np.uint8:
            raise ValueError('images must be uint8 HxWx3')
    # A list may mix sizes (mmdet/apis/inference.py:118-139 collates whatever it is given).  Images are grouped by size, every size
    # runs on its own engine (the Detector keeps the most recently used `max_engines` sizes) and the results go back in input order.
    # Deviation, stated: the reference pads a mixed batch to its largest member, so there an image's features near its right / bottom
    # edge depend on its batch mates; here every image is computed as the reference computes it in a batch of its own size.
    by_shape = {}
    for i, a in enumerate(arrs):
        by_shape.setdefault(a.shape[:2], []).append(i)
    results = [None] * len(arrs)
    if len(by_shape) > model.max_engines:
        warnings.warn(f'inference_detector: {len(by_shape)} image sizes in one call but the detector keeps {model.max_engines} engines '
                      '(Detector.max_engines): engines are rebuilt (seconds each) within this call; raise max_engines or group the images by size')
    # sizes whose engine is already cached run first, so that this call evicts none of the engines it is about to use
    cached = {k for k in model._engines if len(k) == 2}
    for hw, idx in sorted(by_shape.items(), key=lambda kv: (int(kv[0][0]), int(kv[0][1])) not in cached):
        eng = model.engine(hw)
        for i, r in zip(idx, eng(np.stack([arrs[i] for i in idx]), mode)):
            results[i] = r
    return results if is_batch else results[0]


def concat_results(result):
    """tools/infer_wsi.py:486-494: per-class lists -> flat arrays (class-major order)."""
    bbox, segm = result
    boxes = np.concatenate(bbox, 0) if len(bbox) else np.zeros((0, 5), np.float32)
    labels = np.concatenate([np.full(len(b), c, np.int32) for c, b in enumerate(bbox)]) if len(bbox) else np.zeros(0, np.int32)
    masks = [m for cl in segm for m in cl]
    return boxes, labels, (np.stack(masks) if masks else np.zeros((0, 0, 0), bool))


def save_result(model, img, result, score_thr=0.3, out_file=None, **kw):
    """nuhtc/apis/inference.py:60-82 (matplotlib overlay in the reference): writes a plain mask/box overlay PNG."""
    from PIL import Image, ImageDraw
    arr = _load_image_rgb(img) if isinstance(img, str) else np.asarray(img)
    boxes, labels, masks = concat_results(result)
    out = arr.astype(np.float32).copy()
    palette = np.array([[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 0], [255, 0, 255], [0, 255, 255]], np.float32)
    for b, l, m in zip(boxes, labels, masks):
        if b[4] >= score_thr:
            out[m] = 0.5 * out[m] + 0.5 * palette[l % len(palette)]
    im = Image.fromarray(out.clip(0, 255).astype(np.uint8))
    dr = ImageDraw.Draw(im)
    for b, l in zip(boxes, labels):
        if b[4] >= score_thr:
            dr.rectangle([float(b[0]), float(b[1]), float(b[2]), float(b[3])], outline=tuple(int(v) for v in palette[l % len(palette)]))
    if out_file:
        im.save(out_file)
    return im
