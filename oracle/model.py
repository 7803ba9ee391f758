"""ORACLE — test infrastructure, NOT product code.

fp32 PyTorch-CPU restatement of the reference's htc_lite_swin tile-inference path
(`inference_detector` -> `HybridTaskCascade_Cus.simple_test`), written from the
reference's behaviour, one function per hot-path row of SURVEY §8(a).  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import it.

Pinning: `tests/test_oracle_golden.py` checks every stage of this file against golden
vectors produced by running the reference's *own* Python files in the build container
(oracle/ref_harness/make_golden.py; fixtures in tests/golden/).  The two mmcv native ops
(RoIAlign, NMS), torchvision's gaussian_blur and cv2's uint8 resize are third-party code
absent from /root/reference: they are restated (oracle/ops_np.py and below) and are
"parity unpinned" by the reference; hand-derived known answers pin them instead.

Reference citations use paths relative to /root/reference; `mmdet/` abbreviates
`thirdparty/mmdetection/mmdet/`.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F
from scipy import ndimage as ndi

from . import ops as ops_np  # plain-C restatement (oracle/ops_c.c); oracle/ops_np.py is the numpy cross-check

MEAN = (123.675, 116.28, 103.53)   # configs/nuhtc/htc_lite_swin_pytorch_fpn_PanNuke_seasaw_CAS.py:8
STD = (58.395, 57.12, 57.375)
WS = 7
DEPTHS = (2, 2, 6, 2)
HEADS = (3, 6, 12, 24)
ATT_THRES = 0.965926               # config:4  (cos 15 deg)
STAGE_STDS = ((0.1, 0.1, 0.2, 0.2), (0.05, 0.05, 0.1, 0.1), (0.033, 0.033, 0.067, 0.067))  # config:97,115,133
MAX_RATIO = abs(math.log(16 / 1000))  # mmdet/core/bbox/coder/delta_xywh_bbox_coder.py:39,239


# ----------------------------------------------------------------------------- a1 pre-processing
def _cv_linear_tables(ssize, dsize):
    """Per-axis source offsets and 11-bit fixed-point weights exactly as cv::resize builds them for INTER_LINEAR
    (opencv-python 4.x `imgproc/src/resize.cpp`, `cv::hal::resize` -> the `xofs/ialpha` loop): `scale = 1/(dsize/ssize)` in
    double, `f = (float)((d+0.5)*scale-0.5)`, `s = floor(f)`, `f -= s` in float, weights
    `saturate_cast<short>((1-f)*2048)`, `saturate_cast<short>(f*2048)` (round half to even).  Returned unclamped:
    the two passes treat the borders differently (see `cv2_resize_linear_u8`)."""
    scale = 1.0 / (float(dsize) / float(ssize))
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    return s, f


def cv2_resize_linear_u8(img, dsize_w, dsize_h):
    """cv2.resize(img, (dsize_w, dsize_h), interpolation=cv2.INTER_LINEAR) for uint8 images, bit for bit.

    Reference call chain: mmdet/datasets/pipelines/transforms.py:207-236 (`Resize._resize_img`) -> mmcv.imrescale ->
    cv2.resize(INTER_LINEAR).  OpenCV's 8-bit linear path (`resizeGeneric_<HResizeLinear<uchar,int,short,2048,...>,
    VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>,...>>`; the IPP branch is not taken for 8u linear unless
    `useIPP_NotExact`) is third-party code absent from /root/reference and from this image: PARITY UNPINNED, restated
    from the published source, pinned by the hand-worked cases in tests/test_oracle_ops.py.

      horizontal: sx<0 -> (sx,fx)=(0,0); sx>=W-1 -> (W-1,0);  H[dx] = S[sx]*a0 + S[sx+1]*a1   (int32, 11 fractional bits)
      vertical:   rows clip(sy), clip(sy+1) to [0,H-1], weights NOT reset at the border;
                  dst = ((( b0*(H0>>4) )>>16) + (( b1*(H1>>4) )>>16) + 2) >> 2      -- two separately truncated products
    """
    img = np.asarray(img, np.uint8)
    sq = img.ndim == 2
    if sq:
        img = img[:, :, None]
    Hs, Ws = img.shape[:2]
    if (dsize_w, dsize_h) == (Ws, Hs):     # cv::resize: "dsize == ssize -> src.copyTo(dst)"
        return img[:, :, 0].copy() if sq else img.copy()
    sx, fx = _cv_linear_tables(Ws, dsize_w)
    lo, hi = sx < 0, sx >= Ws - 1
    fx = np.where(lo | hi, np.float32(0), fx)
    sx = np.where(lo, 0, np.where(hi, Ws - 1, sx))
    a0 = np.rint((np.float32(1) - fx) * np.float32(2048)).astype(np.int32)
    a1 = np.rint(fx * np.float32(2048)).astype(np.int32)
    sx1 = np.minimum(sx + 1, Ws - 1)       # weight 0 wherever the clamp acts
    sy, fy = _cv_linear_tables(Hs, dsize_h)
    b0 = np.rint((np.float32(1) - fy) * np.float32(2048)).astype(np.int32)
    b1 = np.rint(fy * np.float32(2048)).astype(np.int32)
    y0 = np.clip(sy, 0, Hs - 1)
    y1 = np.clip(sy + 1, 0, Hs - 1)
    s = img.astype(np.int32)
    h = s[:, sx] * a0[None, :, None] + s[:, sx1] * a1[None, :, None]
    h4 = h >> 4
    v = ((b0[:, None, None] * h4[y0]) >> 16) + ((b1[:, None, None] * h4[y1]) >> 16)
    out = np.clip((v + 2) >> 2, 0, 255).astype(np.uint8)
    return out[:, :, 0] if sq else out


def resize2x_u8(img):
    """The x2 case of `cv2_resize_linear_u8` (tools/infer_wsi.py:416-419: scale_factor = 80/mag = 2 at 40x).  For an exact
    x2 factor the weights are 512/1536 (2048 at the left/right border) and the formula collapses to
    `(((a+3b)>>2) + ((3c+9d)>>2) + 2) >> 2` with (a,b) the far row's (far,near) columns and (c,d) the near row's."""
    img = np.asarray(img)
    return cv2_resize_linear_u8(img, 2 * img.shape[1], 2 * img.shape[0])


def preprocess(tiles_u8, channel_mode=0, scale=2):
    """(B,H,W,3) u8 -> (B,3,2H,2W) f32 network input (mmdet/apis/inference.py:112-139 + test pipeline).

    channel_mode 0 = `tools/infer.py` (file -> BGR ndarray -> to_rgb swap => true RGB meets RGB means):
                     caller passes **RGB** tiles, no swap.
    channel_mode 1 = `tools/infer_wsi.py` (RGB ndarray treated as BGR and swapped):
                     caller passes RGB tiles, channels are reversed before normalisation."""
    out = []
    mean = np.array(MEAN, np.float32)
    istd = (1.0 / np.array(STD, np.float64)).astype(np.float32) if False else None
    for t in np.asarray(tiles_u8):
        r = cv2_resize_linear_u8(t, int(t.shape[1] * scale + 0.5), int(t.shape[0] * scale + 0.5)).astype(np.float32)
        if channel_mode == 1:
            r = r[:, :, ::-1]
        # mmcv.imnormalize: cv2.subtract(img, mean) ; cv2.multiply(img, 1/std) with float64 scalars on f32 data
        stdinv = 1.0 / np.array(STD, np.float64)
        r = ((r - mean) * stdinv.astype(np.float32)).astype(np.float32)
        out.append(np.ascontiguousarray(r.transpose(2, 0, 1)))
    x = torch.from_numpy(np.stack(out))
    H, W = x.shape[-2:]
    ph, pw = (32 - H % 32) % 32, (32 - W % 32) % 32   # Pad(size_divisor=32), transforms.py:570-
    if ph or pw:
        x = F.pad(x, (0, pw, 0, ph))
    return x


# ----------------------------------------------------------------------------- a2-a6 Swin-T
def rel_pos_index():
    c = torch.arange(WS)
    ii, jj = torch.meshgrid(c, c, indexing='ij')
    ii, jj = ii.reshape(-1), jj.reshape(-1)
    return (ii[:, None] - ii[None, :] + WS - 1) * (2 * WS - 1) + (jj[:, None] - jj[None, :] + WS - 1)


def shift_mask(Hp, Wp):
    """(nW,49,49) 0/-100 mask on the padded grid (mmdet/models/backbones/swin.py:197-218)."""
    ids = torch.zeros(Hp, Wp)
    sl = (slice(0, -WS), slice(-WS, -3), slice(-3, None))
    c = 0
    for h in sl:
        for w in sl:
            ids[h, w] = c
            c += 1
    mw = ids.view(Hp // WS, WS, Wp // WS, WS).permute(0, 2, 1, 3).reshape(-1, WS * WS)
    d = mw[:, None, :] - mw[:, :, None]
    return torch.where(d != 0, torch.tensor(-100.0), torch.tensor(0.0))


def window_attention(sd, p, x, H, W, nh, shifted):
    """ShiftWindowMSA.forward + WindowMSA.forward (swin.py:178-252, 79-117). x: (B,H*W,C) already LN1'd."""
    B, L, C = x.shape
    x = x.view(B, H, W, C)
    pr, pb = (WS - W % WS) % WS, (WS - H % WS) % WS
    x = F.pad(x, (0, 0, 0, pr, 0, pb))
    Hp, Wp = H + pb, W + pr
    mask = None
    if shifted:
        x = torch.roll(x, (-3, -3), (1, 2))
        mask = shift_mask(Hp, Wp)
    xw = x.view(B, Hp // WS, WS, Wp // WS, WS, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, WS * WS, C)
    Bn, N = xw.shape[0], WS * WS
    qkv = F.linear(xw, sd[p + 'qkv.weight'], sd[p + 'qkv.bias']).reshape(Bn, N, 3, nh, C // nh).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (C // nh) ** -0.5, qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)
    bias = sd[p + 'relative_position_bias_table'][rel_pos_index().view(-1)].view(N, N, -1).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = (attn.view(Bn // nW, nW, nh, N, N) + mask[None, :, None]).view(-1, nh, N, N)
    attn = attn.softmax(-1)
    o = (attn @ v).transpose(1, 2).reshape(Bn, N, C)
    o = F.linear(o, sd[p + 'proj.weight'], sd[p + 'proj.bias'])
    o = o.view(B, Hp // WS, Wp // WS, WS, WS, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
    if shifted:
        o = torch.roll(o, (3, 3), (1, 2))
    return o[:, :H, :W].reshape(B, H * W, C)


def swin_block(sd, p, x, H, W, nh, shifted):
    """SwinBlock.forward (swin.py:356-376) + mmcv FFN (exact-erf GELU)."""
    C = x.shape[-1]
    h = F.layer_norm(x, (C,), sd[p + 'norm1.weight'], sd[p + 'norm1.bias'], 1e-5)
    x = x + window_attention(sd, p + 'attn.w_msa.', h, H, W, nh, shifted)
    h = F.layer_norm(x, (C,), sd[p + 'norm2.weight'], sd[p + 'norm2.bias'], 1e-5)
    h = F.gelu(F.linear(h, sd[p + 'ffn.layers.0.0.weight'], sd[p + 'ffn.layers.0.0.bias']))
    return x + F.linear(h, sd[p + 'ffn.layers.1.weight'], sd[p + 'ffn.layers.1.bias'])


def patch_embed(sd, img):
    """PatchEmbed.forward (mmdet/models/utils/transformer.py:236-257): conv4x4 s4 + LN(96)."""
    B, _, H, W = img.shape
    img = F.pad(img, (0, (4 - W % 4) % 4, 0, (4 - H % 4) % 4))
    x = F.conv2d(img, sd['backbone.patch_embed.projection.weight'], sd['backbone.patch_embed.projection.bias'], stride=4)
    h, w = x.shape[-2:]
    x = x.flatten(2).transpose(1, 2)
    x = F.layer_norm(x, (96,), sd['backbone.patch_embed.norm.weight'], sd['backbone.patch_embed.norm.bias'], 1e-5)
    return x, h, w


def patch_merge(sd, p, x, H, W):
    """PatchMerging.forward (transformer.py:340-385): Unfold 2x2 (channel-major c*4+kh*2+kw), LN(4C), Linear(4C->2C)."""
    B, L, C = x.shape
    x = x.view(B, H, W, C).permute(0, 3, 1, 2)
    x = F.pad(x, (0, W % 2, 0, H % 2))
    x = F.unfold(x, 2, stride=2).transpose(1, 2)      # (B, L/4, 4C)
    x = F.layer_norm(x, (4 * C,), sd[p + 'norm.weight'], sd[p + 'norm.bias'], 1e-5)
    return F.linear(x, sd[p + 'reduction.weight']), (H + 1) // 2, (W + 1) // 2


def backbone(sd, img, return_tokens=False):
    """SwinTransformer.forward (swin.py:746-764) -> 4 NCHW maps."""
    x, H, W = patch_embed(sd, img)
    outs, toks = [], {'embed': x}
    for s in range(4):
        for b in range(DEPTHS[s]):
            x = swin_block(sd, f'backbone.stages.{s}.blocks.{b}.', x, H, W, HEADS[s], b % 2 == 1)
            toks[f's{s}b{b}'] = x
        C = x.shape[-1]
        o = F.layer_norm(x, (C,), sd[f'backbone.norm{s}.weight'], sd[f'backbone.norm{s}.bias'], 1e-5)
        outs.append(o.view(-1, H, W, C).permute(0, 3, 1, 2).contiguous())
        if s < 3:
            x, H, W = patch_merge(sd, f'backbone.stages.{s}.downsample.', x, H, W)
    return (outs, toks) if return_tokens else outs


# ----------------------------------------------------------------------------- a7 FPN, a8 RPN convs, a13 semantic head
def fpn(sd, feats):
    """FPN.forward (mmdet/models/necks/fpn.py:152-179): 1x1 laterals, nearest top-down, 3x3, no act."""
    lat = [F.conv2d(f, sd[f'neck.lateral_convs.{i}.conv.weight'], sd[f'neck.lateral_convs.{i}.conv.bias'])
           for i, f in enumerate(feats)]
    for i in range(3, 0, -1):
        lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode='nearest')
    return [F.conv2d(l, sd[f'neck.fpn_convs.{i}.conv.weight'], sd[f'neck.fpn_convs.{i}.conv.bias'], padding=1)
            for i, l in enumerate(lat)]


def rpn_convs(sd, feats):
    """RPNHead.forward_single (mmdet/models/dense_heads/rpn_head.py:62-68), shared over levels."""
    cls, reg = [], []
    for f in feats:
        h = F.relu(F.conv2d(f, sd['rpn_head.rpn_conv.weight'], sd['rpn_head.rpn_conv.bias'], padding=1))
        cls.append(F.conv2d(h, sd['rpn_head.rpn_cls.weight'], sd['rpn_head.rpn_cls.bias']))
        reg.append(F.conv2d(h, sd['rpn_head.rpn_reg.weight'], sd['rpn_head.rpn_reg.bias']))
    return cls, reg


def semantic_head(sd, feats):
    """FusedSemanticHead.forward (mmdet/models/roi_heads/mask_heads/fused_semantic_head.py:97-111)."""
    p = 'roi_head.semantic_head.'

    def lat(i, f):
        return F.relu(F.conv2d(f, sd[p + f'lateral_convs.{i}.conv.weight'], sd[p + f'lateral_convs.{i}.conv.bias']))
    x = lat(0, feats[0])
    size = tuple(x.shape[-2:])
    for i in range(1, 4):
        x = x + lat(i, F.interpolate(feats[i], size=size, mode='bilinear', align_corners=True))
    for j in range(4):
        x = F.relu(F.conv2d(x, sd[p + f'convs.{j}.conv.weight'], sd[p + f'convs.{j}.conv.bias'], padding=1))
    pred = F.conv2d(x, sd[p + 'conv_logits.weight'], sd[p + 'conv_logits.bias'])
    feat = F.relu(F.conv2d(x, sd[p + 'conv_embedding.conv.weight'], sd[p + 'conv_embedding.conv.bias']))
    return pred, feat


# ----------------------------------------------------------------------------- a9-a12 RPN proposals
def anchors_for_level(h, w, stride):
    """AnchorGenerator(scales=[4], ratios=[.5,1,2], strides=...) (mmdet/core/anchor/anchor_generator.py:151-194,241-281)."""
    ratios = torch.tensor([0.5, 1.0, 2.0])
    hr = torch.sqrt(ratios)
    wr = 1 / hr
    ws = (stride * wr[:, None] * torch.tensor([4.0])[None, :]).view(-1)
    hs = (stride * hr[:, None] * torch.tensor([4.0])[None, :]).view(-1)
    base = torch.stack([-0.5 * ws, -0.5 * hs, 0.5 * ws, 0.5 * hs], -1)   # center_offset = 0
    sx = torch.arange(0, w).float() * stride
    sy = torch.arange(0, h).float() * stride
    xx = sx.repeat(h)
    yy = sy.view(-1, 1).repeat(1, w).view(-1)
    shifts = torch.stack([xx, yy, xx, yy], -1)
    return (base[None] + shifts[:, None]).view(-1, 4)


def delta2bbox(rois, deltas, stds, max_hw):
    """mmdet/core/bbox/coder/delta_xywh_bbox_coder.py:230-260 (means 0)."""
    if rois.shape[0] == 0:
        return deltas.clone()
    d = deltas * torch.tensor(stds, dtype=torch.float32).view(1, 4)
    pxy = (rois[:, :2] + rois[:, 2:]) * 0.5
    pwh = rois[:, 2:] - rois[:, :2]
    dwh = d[:, 2:].clamp(min=-MAX_RATIO, max=MAX_RATIO)
    gxy = pxy + pwh * d[:, :2]
    gwh = pwh * dwh.exp()
    b = torch.cat([gxy - gwh * 0.5, gxy + gwh * 0.5], -1)
    b[:, 0::2] = b[:, 0::2].clamp(min=0, max=max_hw[1])
    b[:, 1::2] = b[:, 1::2].clamp(min=0, max=max_hw[0])
    return b


def rpn_proposals(cls, reg, img_hw, nms_pre=3000, max_per_img=1000, iou=0.7, min_size=10):
    """RPNHead._get_bboxes_single + _bbox_post_process (rpn_head.py:103-236) per image -> list[(n,5)].

    Sort ties are broken by lower index (stable), the deterministic rule the HIP engine follows too."""
    B = cls[0].shape[0]
    out = []
    for b in range(B):
        sc, dl, an, ids = [], [], [], []
        for lvl, (c, r) in enumerate(zip(cls, reg)):
            h, w = c.shape[-2:]
            s = c[b].permute(1, 2, 0).reshape(-1).sigmoid()
            d = r[b].permute(1, 2, 0).reshape(-1, 4)
            a = anchors_for_level(h, w, 4 << lvl)
            if 0 < nms_pre < s.shape[0]:
                rs, ri = s.sort(descending=True, stable=True)
                ri = ri[:nms_pre]
                s, d, a = rs[:nms_pre], d[ri], a[ri]
            sc.append(s); dl.append(d); an.append(a); ids.append(torch.full((s.shape[0],), lvl))
        sc, dl, an, ids = torch.cat(sc), torch.cat(dl), torch.cat(an), torch.cat(ids)
        props = delta2bbox(an, dl, (1.0, 1.0, 1.0, 1.0), img_hw)
        wv, hv = props[:, 2] - props[:, 0], props[:, 3] - props[:, 1]
        valid = (wv > min_size) & (hv > min_size)
        props, sc, ids = props[valid], sc[valid], ids[valid]
        if props.shape[0] == 0:
            out.append(props.new_zeros(0, 5))
            continue
        dets, _ = ops_np.batched_nms(props.numpy(), sc.numpy(), ids.numpy(), iou)
        out.append(torch.from_numpy(dets[:max_per_img]))
    return out


# ----------------------------------------------------------------------------- a14 connected-component ("watershed") proposals
def gaussian_kernel5():
    """torchvision gaussian_blur(kernel_size=5): sigma = 0.15*5+0.35 = 1.1 (absent third-party)."""
    x = torch.linspace(-2, 2, 5)
    pdf = torch.exp(-0.5 * (x / 1.1).pow(2))
    return pdf / pdf.sum()


def semantic_binary_mask(pred, img_hw):
    """steps 1-4 of _watershed_proposal (nuhtc/models/htc_roi_head_cus.py:284-300): upsample, blur, >0, open(5x5,2)."""
    m = F.interpolate(pred, size=img_hw, mode='bilinear', align_corners=True)
    k1 = gaussian_kernel5()
    k2 = torch.mm(k1[:, None], k1[None, :])[None, None]
    m = F.conv2d(F.pad(m, (2, 2, 2, 2), mode='reflect'), k2)
    m = (m > 0).float()
    ones = torch.ones(1, 1, 5, 5)
    for _ in range(2):   # binary_erosion :238-243
        m = torch.clamp(F.conv2d(m, ones, padding=2) - 25 + 1, 0, 1)
    for _ in range(2):   # binary_dilate :245-250
        m = torch.clamp(F.conv2d(m, ones, padding=2), 0, 1)
    return m[:, 0]


def cc_proposals(pred, img_hw, min_area=10):
    """_watershed_proposal steps 5-6 (:301-335): fill holes, 4-connected labels in raster order of first
    pixel (== watershed output), area filter, boxes [xmin,ymin,xmax+1,ymax+1,1]."""
    masks = semantic_binary_mask(pred, img_hw).numpy()
    max_area = img_hw[0] * img_hw[1] / 4
    out = []
    for m in masks:
        filled = ndi.binary_fill_holes(m)
        lab, n = ndi.label(filled)
        boxes = []
        if n:
            areas = np.bincount(lab.reshape(-1), minlength=n + 1)
            for i, sl in enumerate(ndi.find_objects(lab), start=1):
                if sl is None or not (min_area < areas[i] < max_area):
                    continue
                boxes.append([sl[1].start, sl[0].start, sl[1].stop, sl[0].stop, 1.0])
        out.append(torch.tensor(boxes, dtype=torch.float32).view(-1, 5))
    return out


# ----------------------------------------------------------------------------- a16-a18 RoI features + bbox head
ATT_POOL_FP16 = False      # module switch read by attention_pool: the reference's arithmetic on a CUDA device (see attention_pool)


def attention_pool(feat, rois, stride, thres=ATT_THRES, fp16=None):
    """levels 2,3 of AttentionRoIExtractor.forward (nuhtc/models/roi_extractors_cus.py:220-238) -> (R,C) vectors.

    fp16 (default: the module switch ATT_POOL_FP16, off): what the reference computes when its feature maps are on a CUDA device --
    `roi_dtype = torch.float16 if feats[0].is_cuda` (:203), `feat = feats[i].to(roi_dtype)` (:231): the very same tensor expressions on
    fp16 tensors (each operation rounds to fp16, reductions accumulate in fp32), with cosine_similarity written out as torch 1.13.1 -- the
    version the reference pins, README.md:86 -- computes it (w12 / sqrt(clamp_min(w1 * w2, eps^2)), ATen/native/Distance.cpp); the fp16
    result is added into the fp32 RoI features (:247).  The engine's twin is nuhtc_config.att_pool_fp16."""
    fp16 = ATT_POOL_FP16 if fp16 is None else fp16
    N, C, H, W = feat.shape
    b = rois[:, 0].long()
    cx = torch.div(rois[:, 1] + rois[:, 3], 2 * stride, rounding_mode='floor').clamp(0, W - 1).long()
    cy = torch.div(rois[:, 2] + rois[:, 4], 2 * stride, rounding_mode='floor').clamp(0, H - 1).long()
    key = (b * H + cy) * W + cx
    uk, inv = torch.unique(key, return_inverse=True)
    ub, ucy, ucx = uk // (H * W), (uk // W) % H, uk % W
    if fp16:
        fh = feat.to(torch.float16)
        x1 = fh[ub, :, ucy, ucx][:, None, :]                     # (U,1,C)
        fv = fh.permute(0, 2, 3, 1).reshape(N, H * W, C)[ub]      # (U,HW,C)
        w12, w1, w2 = (x1 * fv).sum(2), (x1 * x1).sum(2), (fv * fv).sum(2)
        cos = w12 / (w1 * w2).clamp_min(1e-8 * 1e-8).sqrt()
        sim = F.relu(cos - thres) + thres
        return (fv * sim[..., None]).mean(1).float()[inv]
    q = feat[ub, :, ucy, ucx]                                    # (U,C)
    fv = feat.permute(0, 2, 3, 1).reshape(N, H * W, C)[ub]       # (U,HW,C)
    sim = F.relu(F.cosine_similarity(q[:, None, :], fv, dim=2) - thres) + thres
    g = (fv * sim[..., None]).mean(1)                            # (U,C)
    return g[inv]


def roi_extract(x, rois, out_size, sampling_ratio):
    """AttentionRoIExtractor.forward with 4 levels, start_level>=2, aggregation='sum' (:194-259)."""
    R = rois.shape[0]
    out = torch.zeros(R, x[0].shape[1], out_size, out_size)
    if R == 0:
        return out
    rn = rois.numpy()
    for i in range(4):
        if i < 2:
            out += torch.from_numpy(ops_np.roi_align(x[i].numpy(), rn, out_size, 1.0 / (4 << i), sampling_ratio))
        else:
            out += attention_pool(x[i], rois, 4 << i)[:, :, None, None]
    return out


def semantic_roi(sem_feat, rois):
    """single-level call of the extractor == plain RoIAlign(14, sr=0, scale 1/4) (:197-198)."""
    return torch.from_numpy(ops_np.roi_align(sem_feat.numpy(), rois.numpy(), 14, 0.25, 0))


def bbox_feats(x, sem_feat, rois):
    """_bbox_forward feature part (nuhtc/models/htc_roi_head_cus.py:187-199)."""
    f = roi_extract(x, rois, 7, 2)
    return f + F.adaptive_avg_pool2d(semantic_roi(sem_feat, rois), (7, 7))


def bbox_head(sd, k, feats):
    """ConvFCBBoxHead.forward (mmdet/models/roi_heads/bbox_heads/convfc_bbox_head.py:158-196) with NormedLinear
    cls predictor (mmdet/models/utils/normed_predictor.py:33-38)."""
    p = f'roi_head.bbox_head.{k}.'
    h = feats.flatten(1)
    h = F.relu(F.linear(h, sd[p + 'shared_fcs.0.weight'], sd[p + 'shared_fcs.0.bias']))
    h = F.relu(F.linear(h, sd[p + 'shared_fcs.1.weight'], sd[p + 'shared_fcs.1.bias']))
    w = sd[p + 'fc_cls.weight']
    w_ = w / (w.norm(dim=1, keepdim=True) + 1e-6)
    x_ = h / (h.norm(dim=1, keepdim=True) + 1e-6) * 20
    cls = F.linear(x_, w_, sd[p + 'fc_cls.bias'])
    reg = F.linear(h, sd[p + 'fc_reg.weight'], sd[p + 'fc_reg.bias'])
    return cls, reg


# ----------------------------------------------------------------------------- a19-a22 cascade + detection post-processing
def seesaw_scores(cls):
    """SeesawLoss.get_activation (mmdet/models/losses/seesaw_loss.py:157-175)."""
    nc = cls.shape[1] - 2
    sc = F.softmax(cls[:, :nc], -1)
    so = F.softmax(cls[:, nc:], -1)
    return torch.cat([sc * so[:, :1], so[:, 1:2]], -1)


def detect_post(rois_xyxy, cls_mean, reg3, img_hw, scale, score_thr=0.35, iou=0.5, max_per_img=500):
    """Shared2FCBBoxHeadWithProb.get_bboxes + multiclass_nms (nuhtc/models/bbox_head.py:230-292,12-102)."""
    scores = seesaw_scores(cls_mean)
    boxes = delta2bbox(rois_xyxy, reg3, STAGE_STDS[2], img_hw)
    boxes = boxes / torch.tensor([scale] * 4, dtype=torch.float32)
    nc = scores.shape[1] - 1
    s = scores[:, :nc].reshape(-1)
    bx = boxes[:, None, :].expand(-1, nc, 4).reshape(-1, 4)
    lb = torch.arange(nc).view(1, -1).expand(scores.shape[0], nc).reshape(-1)
    inds = (s > score_thr).nonzero().squeeze(1)
    bx, s, lb = bx[inds], s[inds], lb[inds]
    if bx.shape[0] == 0:
        return torch.zeros(0, 5), torch.zeros(0, dtype=torch.long)
    dets, keep = ops_np.batched_nms(bx.numpy(), s.numpy(), lb.numpy(), iou)
    return torch.from_numpy(dets[:max_per_img]), lb[torch.from_numpy(keep[:max_per_img])]


# ----------------------------------------------------------------------------- a23-a26 mask branch
def mask_head(sd, feats):
    """HTCMaskHead.forward, res_feat=None (mmdet/models/roi_heads/mask_heads/htc_mask_head.py:22-39) + sigmoid (:2349)."""
    p = 'roi_head.mask_head.0.'
    x = feats
    for j in range(4):
        x = F.relu(F.conv2d(x, sd[p + f'convs.{j}.conv.weight'], sd[p + f'convs.{j}.conv.bias'], padding=1))
    x = F.relu(F.conv_transpose2d(x, sd[p + 'upsample.weight'], sd[p + 'upsample.bias'], stride=2))
    return F.conv2d(x, sd[p + 'conv_logits.weight'], sd[p + 'conv_logits.bias']).sigmoid()


def paste_masks(prob, boxes, H, W, thr=0.5, return_values=False):
    """FCNMaskHead.get_seg_masks/_do_paste_mask (mmdet/models/roi_heads/mask_heads/fcn_mask_head.py:229-307,344-412).
    prob (D,1,28,28), boxes (D,4) in output-pixel space -> bool (D,H,W).

    CPU semantics of the reference: one instance per chunk with skip_empty=True, i.e. only pixels inside the
    integer hull [floor(x0)-1, ceil(x1)+1) x [floor(y0)-1, ceil(y1)+1) (clamped to the canvas) are sampled,
    everything else stays False (this matters for boxes wider than ~112 px, whose bilinear tail leaves the hull).
    return_values=True (tests only) also returns the sampled probabilities (D,H,W), so that a pixel that differs between
    two implementations can be shown to sit on the threshold."""
    D = prob.shape[0]
    if D == 0:
        return (np.zeros((0, H, W), bool), np.zeros((0, H, W), np.float32)) if return_values else np.zeros((0, H, W), bool)
    x0, y0, x1, y1 = torch.split(boxes, 1, dim=1)
    iy = torch.arange(0, H).float() + 0.5
    ix = torch.arange(0, W).float() + 0.5
    hx0 = torch.clamp(x0.floor() - 1, min=0).to(torch.int32)
    hy0 = torch.clamp(y0.floor() - 1, min=0).to(torch.int32)
    hx1 = torch.clamp(x1.ceil() + 1, max=W).to(torch.int32)
    hy1 = torch.clamp(y1.ceil() + 1, max=H).to(torch.int32)
    px = torch.arange(0, W)[None, :]
    py = torch.arange(0, H)[None, :]
    in_hull = ((py >= hy0) & (py < hy1))[:, :, None] & ((px >= hx0) & (px < hx1))[:, None, :]
    iy = (iy - y0) / (y1 - y0) * 2 - 1
    ix = (ix - x0) / (x1 - x0) * 2 - 1
    ix[torch.isinf(ix)] = 0
    iy[torch.isinf(iy)] = 0
    grid = torch.stack([ix[:, None, :].expand(D, H, W), iy[:, :, None].expand(D, H, W)], 3)
    m = F.grid_sample(prob.float(), grid, align_corners=False)[:, 0]
    if return_values:
        return ((m >= thr) & in_hull).numpy(), m.numpy()
    return ((m >= thr) & in_hull).numpy()


# ----------------------------------------------------------------------------- whole path
class Oracle:
    """`simple_test` of HybridTaskCascade_Cus (nuhtc/models/htc_cus.py:110-121) for a batch of tiles."""

    def __init__(self, state_dict, num_classes=5, score_thr=0.35, max_per_img=500, scale=2.0):
        self.sd = state_dict
        self.nc = num_classes
        self.score_thr = score_thr
        self.max_per_img = max_per_img
        self.scale = scale

    @torch.no_grad()
    def forward_tensor(self, img, ori_hw, fixed_rois=None, keep=False, img_hw=None, timing=None):
        """img: (B,3,Hn,Wn) f32 network input (the padded batch tensor). Returns list of (bbox_results, segm_results) like
        the reference (+ dict of intermediates when keep=True).

        img_hw = `img_meta['img_shape']`, the resized image BEFORE Pad(size_divisor=32) (default: the tensor's own size, i.e. no
        padding; `timing`: a dict that receives the seconds spent per stage of BASELINE.md section 3 -- backbone / fpn_rpn_semantic /
        proposals / cascade / mask / post -- accumulated over calls, for bench.py's CPU baseline): the reference clips RPN proposals (rpn_head.py:141,219), refined RoIs and detections (bbox_head.py:358,533) to
        it and interpolates the semantic logits to it for the component proposals (htc_roi_head_cus.py:285,397), while anchors,
        feature maps and RoI features live on the padded tensor; masks are pasted into `ori_shape` = ori_hw."""
        import time as _time
        sd = self.sd
        B = img.shape[0]
        img_hw = tuple(img.shape[-2:]) if img_hw is None else tuple(int(v) for v in img_hw)
        _t = [_time.perf_counter()]

        def lap(stage):
            now = _time.perf_counter()
            if timing is not None:
                timing[stage] = timing.get(stage, 0.0) + now - _t[0]
            _t[0] = now
        c = backbone(sd, img)
        lap('backbone')
        x = fpn(sd, c)
        rcls, rreg = rpn_convs(sd, x)
        sem_pred, sem_feat = semantic_head(sd, x)
        lap('fpn_rpn_semantic')
        if fixed_rois is None:
            rpn = rpn_proposals(rcls, rreg, img_hw)
            ws = cc_proposals(sem_pred, img_hw)
            props = [torch.cat([w[:, :4], r[:, :4]], 0) for w, r in zip(ws, rpn)]
        else:
            rpn, ws = None, None
            props = [torch.as_tensor(r, dtype=torch.float32) for r in fixed_rois]
        lap('proposals')
        n_per = [p.shape[0] for p in props]
        rois = torch.cat([torch.cat([torch.full((p.shape[0], 1), float(i)), p], 1) for i, p in enumerate(props)], 0)
        inter = dict(c=c, x=x, rpn_cls=rcls, rpn_reg=rreg, sem_pred=sem_pred, sem_feat=sem_feat, rpn=rpn, ws=ws,
                     rois0=rois.clone(), stage_cls=[], stage_reg=[], stage_rois=[])
        if rois.shape[0] == 0:
            res = [([np.zeros((0, 5), np.float32) for _ in range(self.nc)], [[] for _ in range(self.nc)]) for _ in range(B)]
            return (res, inter) if keep else res
        ms = []
        for k in range(3):
            cls, reg = bbox_head(sd, k, bbox_feats(x, sem_feat, rois))
            ms.append(cls)
            inter['stage_cls'].append(cls); inter['stage_reg'].append(reg); inter['stage_rois'].append(rois.clone())
            if k < 2:   # regress_by_class, class-agnostic (mmdet/.../bbox_head.py:459-496)
                rois = torch.cat([rois[:, :1], delta2bbox(rois[:, 1:], reg, STAGE_STDS[k], img_hw)], 1)
        cls_mean = sum(ms) / 3.0
        lap('cascade')
        dets, labels, off = [], [], 0
        for i in range(B):
            sl = slice(off, off + n_per[i]); off += n_per[i]
            d, l = detect_post(rois[sl, 1:], cls_mean[sl], reg[sl], img_hw, self.scale, self.score_thr, 0.5, self.max_per_img)
            dets.append(d); labels.append(l)
        inter.update(dets=dets, labels=labels)
        lap('post')
        # mask branch (htc_roi_head_cus.py:2310-2367)
        mrois = torch.cat([torch.cat([torch.full((d.shape[0], 1), float(i)), d[:, :4] * self.scale], 1)
                           for i, d in enumerate(dets)], 0)
        results = []
        if mrois.shape[0]:
            mf = roi_extract(x, mrois, 14, 0) + semantic_roi(sem_feat, mrois)
            prob = mask_head(sd, mf)
        else:
            prob = torch.zeros(0, 1, 28, 28)
        inter.update(mask_prob=prob, mask_rois=mrois)
        lap('mask')
        off = 0
        for i in range(B):
            d, l = dets[i], labels[i]
            pm = paste_masks(prob[off:off + d.shape[0]], (d[:, :4] * self.scale) / self.scale, ori_hw[0], ori_hw[1])
            off += d.shape[0]
            bbox_res = [d[l == c].numpy() for c in range(self.nc)]
            segm_res = [[pm[j] for j in range(d.shape[0]) if int(l[j]) == c] for c in range(self.nc)]
            results.append((bbox_res, segm_res))
        lap('post')
        return (results, inter) if keep else results

    def __call__(self, tiles_u8, channel_mode=0, **kw):
        tiles_u8 = np.asarray(tiles_u8)
        h, w = tiles_u8.shape[1:3]
        img_hw = (int(h * self.scale + 0.5), int(w * self.scale + 0.5))       # mmcv.imrescale: new size = int(size * scale + 0.5)
        import time as _time
        t0 = _time.perf_counter()
        img = preprocess(tiles_u8, channel_mode, self.scale)
        if kw.get('timing') is not None:
            kw['timing']['preprocess'] = kw['timing'].get('preprocess', 0.0) + _time.perf_counter() - t0
        return self.forward_tensor(img, (h, w), img_hw=img_hw, **kw)


# ----------------------------------------------------------------------------- a27-a28 per-tile filter + mask-NMS
def tile_filter_and_mask_nms(bbox_res, segm_res, size=256, margin=2, min_area=10, thr=0.05):
    """tools/infer_wsi.py:486-531 + mask_nms :60-84 (pycocotools RLE IoU == exact integer popcount IoU)."""
    boxes = np.concatenate(bbox_res, 0) if len(bbox_res) else np.zeros((0, 5), np.float32)
    labels = np.concatenate([np.full(len(b), c, np.int64) for c, b in enumerate(bbox_res)]) if len(bbox_res) else np.zeros(0, np.int64)
    masks = [m for cl in segm_res for m in cl]
    if not len(masks):
        return boxes[:0], labels[:0], np.zeros((0, size, size), bool)
    masks = np.stack(masks)
    area = masks.reshape(len(masks), -1).sum(1)
    keep = ((boxes[:, 0] >= margin) & (boxes[:, 1] >= margin) & (boxes[:, 2] <= size - margin) &
            (boxes[:, 3] <= size - margin) & (area >= min_area))
    boxes, labels, masks, area = boxes[keep], labels[keep], masks[keep], area[keep]
    order = np.argsort(boxes[:, 4])[::-1]
    flat = masks.reshape(len(masks), -1).astype(np.int32)
    inter = flat @ flat.T
    union = area[:, None] + area[None, :] - inter
    iou = inter / np.maximum(union, 1)
    sup = np.zeros(len(masks), bool)
    kept = []
    for a, i in enumerate(order):
        if sup[i]:
            continue
        kept.append(i)
        for j in order[a + 1:]:
            if iou[i, j] > thr:
                sup[j] = True
    kept = np.array(kept, np.int64)
    return boxes[kept], labels[kept], masks[kept]
