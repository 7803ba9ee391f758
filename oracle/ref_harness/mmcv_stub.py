"""Import shim that lets the *reference's own Python files* run on CPU in the
build container (TEST INFRASTRUCTURE ONLY — never imported by the product).

The reference (boyden/NuHTC + vendored mmdetection 2.18) is pure Python but
depends on packages absent from this image (mmcv-full 1.7.2, torchvision,
scikit-image, pycocotools, cv2 ...).  This module fabricates permissive stub
packages for those names and fills in real implementations for the ~25
symbols that sit on the htc_lite_swin inference path, so that
`/root/reference` can be imported *unmodified* and driven to produce golden
vectors (see make_golden.py).  Nothing here is copied from the reference.

What the stand-ins restate (third-party code that is NOT under
/root/reference, so these are "parity unpinned" by the reference itself):
  * mmcv.ops.RoIAlign       -> oracle.ops_np.roi_align  (mmcv-full 1.7.2 semantics)
  * mmcv.ops.nms/batched_nms-> oracle.ops_np.nms / batched_nms
  * torchvision gaussian_blur, skimage watershed (identity on markers)
  * thin mmcv.cnn wrappers (ConvModule, FFN, build_*_layer) over torch.nn
"""
import copy
import importlib.abc
import importlib.machinery
import inspect
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REF_ROOT = '/root/reference'
STUB_TOPLEVEL = ('mmcv', 'torchvision', 'skimage', 'pycocotools', 'cv2', 'terminaltables', 'wandb',
                 'seaborn', 'shapely', 'openslide', 'h5py', 'imagecorruptions', 'albumentations',
                 'cityscapesscripts', 'lvis', 'prettytable', 'histomicstk', 'tifffile')


class Dummy:
    """Absorbs any use: attribute access, call, decoration, subclassing, iteration."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        if len(a) == 1 and not k and (inspect.isclass(a[0]) or inspect.isfunction(a[0])):
            return a[0]  # used as a bare decorator
        return Dummy()

    def __getattr__(self, name):
        if name.startswith('__') and name.endswith('__'):
            raise AttributeError(name)
        return Dummy()

    def __iter__(self):
        return iter(())

    def __mro_entries__(self, bases):
        return ()

    def __contains__(self, x):
        return False

    def __bool__(self):
        return False


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith('__') and name.endswith('__'):
            raise AttributeError(name)
        return Dummy()


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split('.')[0] in STUB_TOPLEVEL:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


# --------------------------------------------------------------------------- mmcv.utils
class ConfigDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def copy(self):
        return ConfigDict(dict.copy(self))

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def to_cfg(obj):
    if isinstance(obj, dict):
        return ConfigDict({k: to_cfg(v) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return type(obj)(to_cfg(v) for v in obj)
    return obj


_SHARED_TABLE = {}


class Registry:
    """All registries share one table (mmdet's registries have disjoint names)."""

    def __init__(self, name, build_func=None, parent=None, scope=None):
        self.name = name
        self._module_dict = _SHARED_TABLE

    def __contains__(self, k):
        return k in self._module_dict

    def get(self, k):
        return self._module_dict.get(k)

    @property
    def module_dict(self):
        return self._module_dict

    def register_module(self, name=None, force=False, module=None):
        if module is not None:
            self._module_dict[name if isinstance(name, str) else module.__name__] = module
            return module
        if inspect.isclass(name) or inspect.isfunction(name):
            self._module_dict[name.__name__] = name
            return name

        def _reg(cls):
            names = [name] if isinstance(name, str) else (name or [cls.__name__])
            for n in names:
                self._module_dict[n] = cls
            return cls
        return _reg

    def build(self, cfg, default_args=None, **kw):
        return build_from_cfg(cfg, self, default_args)


def build_from_cfg(cfg, registry, default_args=None):
    args = dict(cfg)
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    t = args.pop('type')
    cls = registry.get(t) if isinstance(t, str) else t
    if cls is None:
        raise KeyError(f'{t} is not registered')
    return cls(**args)


def to_2tuple(x):
    return tuple(x) if isinstance(x, (list, tuple)) else (x, x)


def digit_version(s):
    return [int(x) for x in s.split('.') if x.isdigit()]


# --------------------------------------------------------------------------- mmcv.runner
class BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self._is_init = False
        self.init_cfg = copy.deepcopy(init_cfg)

    def init_weights(self):
        pass


class ModuleList(BaseModule, nn.ModuleList):
    def __init__(self, modules=None, init_cfg=None):
        BaseModule.__init__(self, init_cfg)
        nn.ModuleList.__init__(self, modules)


class Sequential(BaseModule, nn.Sequential):
    def __init__(self, *args, init_cfg=None):
        BaseModule.__init__(self, init_cfg)
        nn.Sequential.__init__(self, *args)


def _noop_decorator_factory(*a, **k):
    if len(a) == 1 and not k and callable(a[0]) and not isinstance(a[0], (tuple, list, str)):
        return a[0]
    return lambda f: f


# --------------------------------------------------------------------------- mmcv.cnn
def build_norm_layer(cfg, num_features, postfix=''):
    t = cfg['type']
    if t == 'LN':
        return 'ln' + str(postfix), nn.LayerNorm(num_features, eps=cfg.get('eps', 1e-5))
    raise NotImplementedError(t)


def build_conv_layer(cfg, *args, **kwargs):
    assert cfg is None or cfg.get('type', 'Conv2d') in ('Conv2d', 'Conv'), cfg
    return nn.Conv2d(*args, **kwargs)


def build_upsample_layer(cfg, *args, **kwargs):
    cfg = dict(cfg)
    t = cfg.pop('type')
    assert t == 'deconv', t
    return nn.ConvTranspose2d(*args, **cfg, **kwargs)


def build_activation_layer(cfg):
    t = cfg['type']
    if t == 'ReLU':
        return nn.ReLU(inplace=cfg.get('inplace', False))
    if t == 'GELU':
        return nn.GELU()
    raise NotImplementedError(t)


class ConvModule(nn.Module):
    """conv -> (no norm on this path) -> optional ReLU; parameters live under `.conv`."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias='auto', conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'), inplace=True, **kw):
        super().__init__()
        assert norm_cfg is None, 'norm is never configured on the htc_lite_swin path'
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                              bias=(bias is True or bias == 'auto'))
        self.with_activation = act_cfg is not None
        if self.with_activation:
            self.activate = build_activation_layer(dict(act_cfg, inplace=inplace)
                                                   if act_cfg['type'] == 'ReLU' else act_cfg)

    def forward(self, x):
        x = self.conv(x)
        return self.activate(x) if self.with_activation else x


class FFN(BaseModule):
    """mmcv FFN: layers = Sequential(Sequential(Linear, act, drop), Linear, drop); out = identity + layers(x)."""

    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2, act_cfg=dict(type='ReLU', inplace=True),
                 ffn_drop=0., dropout_layer=None, add_identity=True, init_cfg=None, **kw):
        super().__init__(init_cfg)
        assert num_fcs == 2
        self.layers = nn.Sequential(
            nn.Sequential(nn.Linear(embed_dims, feedforward_channels), build_activation_layer(act_cfg), nn.Dropout(ffn_drop)),
            nn.Linear(feedforward_channels, embed_dims), nn.Dropout(ffn_drop))
        self.add_identity = add_identity

    def forward(self, x, identity=None):
        out = self.layers(x)
        if not self.add_identity:
            return out
        return (x if identity is None else identity) + out


def build_dropout(cfg, default_args=None):
    return nn.Identity()


# --------------------------------------------------------------------------- mmcv.ops
class RoIAlign(nn.Module):
    def __init__(self, output_size, spatial_scale=1.0, sampling_ratio=0, pool_mode='avg', aligned=True, use_torchvision=False):
        super().__init__()
        assert pool_mode == 'avg' and aligned
        self.output_size = to_2tuple(output_size)
        self.spatial_scale = float(spatial_scale)
        self.sampling_ratio = int(sampling_ratio)

    def forward(self, feat, rois):
        from oracle import ops_np
        out = ops_np.roi_align(feat.detach().numpy(), rois.detach().numpy(), self.output_size[0],
                               self.spatial_scale, self.sampling_ratio)
        return torch.from_numpy(out)


def nms(boxes, scores, iou_threshold, offset=0, score_threshold=0, max_num=-1):
    from oracle import ops_np
    assert offset == 0 and score_threshold == 0 and max_num == -1
    keep = ops_np.nms(boxes.detach().numpy(), scores.detach().numpy(), float(iou_threshold))
    keep = torch.from_numpy(keep)
    return torch.cat([boxes[keep], scores[keep, None]], 1), keep


def batched_nms(boxes, scores, idxs, nms_cfg, class_agnostic=False):
    cfg = dict(nms_cfg)
    assert cfg.pop('type', 'nms') == 'nms' and not cfg.pop('class_agnostic', class_agnostic)
    assert boxes.shape[0] < cfg.pop('split_thr', 10000)
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
    dets, keep = nms(boxes + offsets[:, None], scores, **cfg)
    return torch.cat([boxes[keep], dets[:, 4:5]], -1), keep


# --------------------------------------------------------------------------- torchvision / skimage
def gaussian_blur(img, kernel_size, sigma=None):
    k = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
    s = k * 0.15 + 0.35 if sigma is None else sigma
    half = (k - 1) * 0.5
    x = torch.linspace(-half, half, steps=k)
    pdf = torch.exp(-0.5 * (x / s).pow(2))
    k1 = pdf / pdf.sum()
    k2 = torch.mm(k1[:, None], k1[None, :]).to(img.dtype)
    c = img.shape[-3]
    pad = k // 2
    out = F.conv2d(F.pad(img, [pad] * 4, mode='reflect'), k2.expand(c, 1, k, k), groups=c)
    return out


def watershed(image, markers=None, connectivity=1, offset=None, mask=None, **kw):
    # every foreground pixel already carries a marker on this path => flooding is the identity
    assert mask is not None and ((markers > 0) == (np.asarray(mask) > 0)).all()
    return markers * (np.asarray(mask) > 0)


# --------------------------------------------------------------------------- install
def install():
    if getattr(install, '_done', False):
        return
    install._done = True
    sys.meta_path.insert(0, _StubFinder())
    import mmcv
    import mmcv.cnn
    import mmcv.cnn.bricks.transformer as mt
    import mmcv.ops
    import mmcv.ops.nms as mnms
    import mmcv.runner
    import mmcv.runner.base_module as mbm
    import mmcv.utils
    import skimage.segmentation
    import torchvision.transforms.functional as TF

    mmcv.__version__ = '1.7.2'
    mmcv.ConfigDict = ConfigDict
    mmcv.Config = ConfigDict
    mmcv.jit = _noop_decorator_factory
    mmcv.is_tuple_of = lambda seq, t: isinstance(seq, tuple) and all(isinstance(s, t) for s in seq)
    mmcv.utils.Registry = Registry
    mmcv.utils.build_from_cfg = build_from_cfg
    mmcv.utils.to_2tuple = to_2tuple
    mmcv.utils.ConfigDict = ConfigDict
    mmcv.utils.Config = ConfigDict
    mmcv.utils.digit_version = digit_version
    mmcv.runner.BaseModule = BaseModule
    mbm.BaseModule = BaseModule
    mmcv.runner.ModuleList = ModuleList
    mmcv.runner.Sequential = Sequential
    mmcv.runner.force_fp32 = _noop_decorator_factory
    mmcv.runner.auto_fp16 = _noop_decorator_factory
    mmcv.cnn.MODELS = Registry('model')
    mmcv.cnn.CONV_LAYERS = Registry('conv layer')
    mmcv.cnn.ConvModule = ConvModule
    mmcv.cnn.build_norm_layer = build_norm_layer
    mmcv.cnn.build_conv_layer = build_conv_layer
    mmcv.cnn.build_upsample_layer = build_upsample_layer
    mmcv.cnn.build_activation_layer = build_activation_layer
    mt.FFN = FFN
    mt.build_dropout = build_dropout
    mmcv.ops.RoIAlign = RoIAlign
    mmcv.ops.nms = mnms  # module attr; functions below
    mmcv.ops.batched_nms = batched_nms
    mnms.batched_nms = batched_nms
    mnms.nms = nms
    TF.gaussian_blur = gaussian_blur
    skimage.segmentation.watershed = watershed

    for p in (REF_ROOT, REF_ROOT + '/thirdparty/mmdetection'):
        if p not in sys.path:
            sys.path.insert(0, p)


def load_config(path):
    """exec() a standalone mmcv-style config file into an attr-dict."""
    ns = {}
    with open(path) as f:
        exec(compile(f.read(), path, 'exec'), ns)
    return to_cfg({k: v for k, v in ns.items() if not k.startswith('__') and not inspect.ismodule(v)})


def build_reference_detector(config_path, seed=0, std=0.02):
    """Instantiate the reference's HybridTaskCascade_Cus with seeded random weights."""
    install()
    import mmdet.models  # noqa: F401
    import nuhtc.models  # noqa: F401
    from mmdet.models import build_detector
    cfg = load_config(config_path)
    cfg.model.train_cfg = None
    cfg.model.backbone.init_cfg = None
    cfg.model.pretrained = None
    model = build_detector(cfg.model)
    model.eval()
    seed_weights(model, seed, std)
    return model, cfg


def seed_weights(model, seed=0, std=0.02):
    """Deterministic weights shared by reference, oracle and HIP engine.

    N(0,std) for every weight/table, small N(0,std) biases (non-zero so bias paths are exercised),
    LayerNorm gamma = 1 + N(0,std).
    """
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.state_dict().items():
            if not p.dtype.is_floating_point or name.endswith('cum_samples') or name == 'roi_head.kernel':
                continue
            v = torch.randn(p.shape, generator=g) * std
            if ('norm' in name) and name.endswith('.weight'):
                v = v + 1.0
            p.copy_(v)
