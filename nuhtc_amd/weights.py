"""state_dict schema of the htc_lite_swin detector, seeded synthetic weights and checkpoint loading.

The schema mirrors what `build_detector(cfg.model)` creates in the reference
(configs/nuhtc/htc_lite_swin_pytorch_fpn_PanNuke_seasaw_CAS.py:29-267):
273 entries / 30.75 M parameters for PanNuke (num_classes=5).  `models/pannuke.pth` is not
distributed with the reference tree, so tests and benches use `seeded_state_dict`.
"""
from collections import OrderedDict

import numpy as np
import torch

EMBED = 96
DEPTHS = (2, 2, 6, 2)
HEADS = (3, 6, 12, 24)
WINDOW = 7
FPN_C = 64
FC_C = 256


def schema(num_classes=5):
    """Ordered {name: shape} of every float tensor the engine consumes (buffers it recomputes are omitted)."""
    s = OrderedDict()
    s['backbone.patch_embed.projection.weight'] = (EMBED, 3, 4, 4)
    s['backbone.patch_embed.projection.bias'] = (EMBED,)
    s['backbone.patch_embed.norm.weight'] = (EMBED,)
    s['backbone.patch_embed.norm.bias'] = (EMBED,)
    for st, (d, nh) in enumerate(zip(DEPTHS, HEADS)):
        C = EMBED << st
        for b in range(d):
            p = f'backbone.stages.{st}.blocks.{b}.'
            s[p + 'norm1.weight'] = (C,)
            s[p + 'norm1.bias'] = (C,)
            s[p + 'attn.w_msa.relative_position_bias_table'] = ((2 * WINDOW - 1) ** 2, nh)
            s[p + 'attn.w_msa.qkv.weight'] = (3 * C, C)
            s[p + 'attn.w_msa.qkv.bias'] = (3 * C,)
            s[p + 'attn.w_msa.proj.weight'] = (C, C)
            s[p + 'attn.w_msa.proj.bias'] = (C,)
            s[p + 'norm2.weight'] = (C,)
            s[p + 'norm2.bias'] = (C,)
            s[p + 'ffn.layers.0.0.weight'] = (4 * C, C)
            s[p + 'ffn.layers.0.0.bias'] = (4 * C,)
            s[p + 'ffn.layers.1.weight'] = (C, 4 * C)
            s[p + 'ffn.layers.1.bias'] = (C,)
        if st < 3:
            p = f'backbone.stages.{st}.downsample.'
            s[p + 'norm.weight'] = (4 * C,)
            s[p + 'norm.bias'] = (4 * C,)
            s[p + 'reduction.weight'] = (2 * C, 4 * C)
    for st in range(4):
        s[f'backbone.norm{st}.weight'] = (EMBED << st,)
        s[f'backbone.norm{st}.bias'] = (EMBED << st,)
    for i in range(4):
        s[f'neck.lateral_convs.{i}.conv.weight'] = (FPN_C, EMBED << i, 1, 1)
        s[f'neck.lateral_convs.{i}.conv.bias'] = (FPN_C,)
    for i in range(4):
        s[f'neck.fpn_convs.{i}.conv.weight'] = (FPN_C, FPN_C, 3, 3)
        s[f'neck.fpn_convs.{i}.conv.bias'] = (FPN_C,)
    s['rpn_head.rpn_conv.weight'] = (FPN_C, FPN_C, 3, 3)
    s['rpn_head.rpn_conv.bias'] = (FPN_C,)
    s['rpn_head.rpn_cls.weight'] = (3, FPN_C, 1, 1)
    s['rpn_head.rpn_cls.bias'] = (3,)
    s['rpn_head.rpn_reg.weight'] = (12, FPN_C, 1, 1)
    s['rpn_head.rpn_reg.bias'] = (12,)
    for k in range(3):
        p = f'roi_head.bbox_head.{k}.'
        s[p + 'shared_fcs.0.weight'] = (FC_C, FPN_C * 49)
        s[p + 'shared_fcs.0.bias'] = (FC_C,)
        s[p + 'shared_fcs.1.weight'] = (FC_C, FC_C)
        s[p + 'shared_fcs.1.bias'] = (FC_C,)
        s[p + 'fc_cls.weight'] = (num_classes + 2, FC_C)
        s[p + 'fc_cls.bias'] = (num_classes + 2,)
        s[p + 'fc_reg.weight'] = (4, FC_C)
        s[p + 'fc_reg.bias'] = (4,)
    p = 'roi_head.mask_head.0.'
    for j in range(4):
        s[p + f'convs.{j}.conv.weight'] = (FPN_C, FPN_C, 3, 3)
        s[p + f'convs.{j}.conv.bias'] = (FPN_C,)
    s[p + 'upsample.weight'] = (FPN_C, FPN_C, 2, 2)  # ConvTranspose2d: [in, out, kh, kw]
    s[p + 'upsample.bias'] = (FPN_C,)
    s[p + 'conv_logits.weight'] = (1, FPN_C, 1, 1)
    s[p + 'conv_logits.bias'] = (1,)
    s[p + 'conv_res.conv.weight'] = (FPN_C, FPN_C, 1, 1)  # unused at test time (res_feat is None)
    s[p + 'conv_res.conv.bias'] = (FPN_C,)
    p = 'roi_head.semantic_head.'
    for i in range(4):
        s[p + f'lateral_convs.{i}.conv.weight'] = (FPN_C, FPN_C, 1, 1)
        s[p + f'lateral_convs.{i}.conv.bias'] = (FPN_C,)
    for j in range(4):
        s[p + f'convs.{j}.conv.weight'] = (FPN_C, FPN_C, 3, 3)
        s[p + f'convs.{j}.conv.bias'] = (FPN_C,)
    s[p + 'conv_embedding.conv.weight'] = (FPN_C, FPN_C, 1, 1)
    s[p + 'conv_embedding.conv.bias'] = (FPN_C,)
    s[p + 'conv_logits.weight'] = (1, FPN_C, 1, 1)
    s[p + 'conv_logits.bias'] = (1,)
    return s


def _fan_in(name, shape):
    if name.endswith('upsample.weight'):
        return shape[0]  # transposed conv k2 s2: one tap per output pixel
    if name.endswith('relative_position_bias_table'):
        return 4
    n = 1
    for d in shape[1:]:
        n *= d
    return max(n, 1)


def seeded_state_dict(seed=0, num_classes=5, gain=1.0):
    """Deterministic synthetic weights (CPU generator -> identical on every box).

    Weights ~ N(0, gain²·g/fan_in) so activations stay O(1) through the network (g=2 in ReLU stacks),
    biases ~ N(0, 0.1²), LayerNorm gamma = 1 + N(0, 0.1²).  Not a trained model: detections are
    arbitrary but every branch of the path (proposals, cascade, NMS, masks) is exercised.
    """
    gen = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for name, shape in schema(num_classes).items():
        v = torch.randn(shape, generator=gen, dtype=torch.float32)
        if len(shape) == 1:
            v = v * 0.1
            if name.endswith('.weight'):  # LayerNorm gamma
                v = v + 1.0
        else:
            relu_stack = name.startswith(('neck.', 'rpn_head.', 'roi_head.'))
            g = 2.0 if relu_stack else 1.0
            v = v * (gain * (g / _fan_in(name, shape)) ** 0.5)
        # keep box regression well-conditioned like a trained model (|delta| < ~1): otherwise proposals blow up to
        # tile-sized boxes and fp32 rounding differences are amplified through the cascade
        # (and keep objectness logits out of sigmoid saturation: exact score ties are ordered by an unstable sort
        # in the reference, which no re-implementation can reproduce)
        if name.endswith('rpn_cls.weight'):
            v = v * 0.2
        elif name.endswith('rpn_reg.weight'):
            v = v * 0.03
        elif name.endswith('rpn_reg.bias'):
            v = v * 0.5
        elif name.endswith('fc_reg.weight'):
            v = v * 0.02
        sd[name] = v
    # make the synthetic model produce work for every stage: positive semantic logit bias (foreground
    # blobs for the connected-component proposals) is tuned in tests via `bias_overrides`
    return sd


def relative_position_index(ws=WINDOW):
    """idx[a][b] = (i_a - i_b + ws-1)·(2ws-1) + (j_a - j_b + ws-1)  (mmdet swin.py:57-66 / SURVEY A.2)."""
    c = np.arange(ws)
    ii, jj = np.meshgrid(c, c, indexing='ij')
    ii, jj = ii.reshape(-1), jj.reshape(-1)
    return ((ii[:, None] - ii[None, :] + ws - 1) * (2 * ws - 1) + (jj[:, None] - jj[None, :] + ws - 1)).astype(np.int64)


def load_checkpoint(path, num_classes=5, return_meta=False):
    """Load an mmdet checkpoint ({'meta','state_dict',...} or a bare state_dict), non-strict like
    nuhtc/apis/inference.py:44: EMA buffers, optimizer state and recomputable buffers are ignored;
    a missing or mis-shaped tensor on the inference path is an error (the reference would silently
    keep random init there, which is never what a user wants).  return_meta: also the checkpoint's
    `meta` dict ({} when it has none; `init_detector` takes CLASSES from it, :45-46)."""
    meta = {}
    if path.endswith('.npz'):
        raw = {k: torch.from_numpy(v) for k, v in np.load(path).items()}
    else:
        raw = torch.load(path, map_location='cpu', weights_only=False)
    if isinstance(raw, dict) and 'state_dict' in raw:
        meta = raw.get('meta') or {}
        raw = raw['state_dict']
    raw = {(k[7:] if k.startswith('module.') else k): v for k, v in raw.items()}
    sd = OrderedDict()
    for name, shape in schema(num_classes).items():
        if name not in raw:
            if 'conv_res' in name:  # unused at test time
                sd[name] = torch.zeros(shape)
                continue
            raise KeyError(f'checkpoint {path} lacks {name}')
        t = raw[name].detach().to(torch.float32).contiguous()
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f'{name}: checkpoint shape {tuple(t.shape)} != expected {tuple(shape)}')
        sd[name] = t
    return (sd, meta) if return_meta else sd


def bench_state_dict(seed=0, num_classes=5, obj_bias=-0.2):
    """Synthetic weights for bench.py / smoke(): seeded_state_dict plus fixed offsets on every stage's fc_cls.bias
    that give the path a realistic load: two classes compete around the 0.35 score threshold, and the objectness
    offset `obj_bias` sets how many (roi, class) pairs pass it.  obj_bias=-0.2 was calibrated with the oracle on
    synth.nuclei_tiles to ~64 detections per 256x256 tile out of ~1020 RoIs (the load SURVEY §8d quotes);
    obj_bias=3.0 saturates max_per_img=500."""
    sd = seeded_state_dict(seed, num_classes)
    add = torch.zeros(num_classes + 2)
    add[:2] = 1.2
    add[-2] = obj_bias
    for k in range(3):
        sd[f'roi_head.bbox_head.{k}.fc_cls.bias'] = sd[f'roi_head.bbox_head.{k}.fc_cls.bias'] + add
    return sd
