"""ORACLE (test infrastructure, not product code).

CPU restatements of the two native ops the reference path takes from the
un-vendored dependency **mmcv-full==1.7.2** (absent from /root/reference):

  * `roi_align`  — mmcv `ops.RoIAlign` forward, pool_mode='avg', aligned=True.
      reference call sites: thirdparty/mmdetection/mmdet/models/roi_heads/roi_extractors/base_roi_extractor.py:53-58,
      nuhtc/models/roi_extractors_cus.py:198,218
  * `nms` / `batched_nms` — mmcv `ops.nms`, `ops.batched_nms`.
      reference call sites: thirdparty/mmdetection/mmdet/models/dense_heads/rpn_head.py:232,
      nuhtc/models/bbox_head.py:93

PARITY UNPINNED by the reference: no reference test holds numbers for these
ops; the published mmcv algorithm is restated here and pinned by
hand-derived known answers in tests/test_oracle_ops.py.

All arithmetic is float32, as in mmcv's `float` kernels.
"""
import numpy as np

f32 = np.float32


def _axis_terms(c, size):
    """Per-axis bilinear pieces of mmcv `bilinear_interpolate` (float32).

    c: sample coordinates (any shape, f32).  Returns (valid, low, high, w_low, w_high)."""
    valid = (c >= f32(-1.0)) & (c <= f32(size))
    c = np.maximum(c, f32(0.0))
    lo = c.astype(np.int32)  # truncation == floor for c >= 0
    edge = lo >= size - 1
    lo = np.where(edge, size - 1, lo)
    hi = np.where(edge, size - 1, lo + 1)
    c = np.where(edge, lo.astype(f32), c)
    l = (c - lo.astype(f32)).astype(f32)
    h = (f32(1.0) - l).astype(f32)
    return valid, lo, hi, h, l


def roi_align(feat, rois, out_size, spatial_scale, sampling_ratio, chunk=256):
    """feat (N,C,H,W) f32; rois (R,5) f32 [batch, x1, y1, x2, y2] -> (R,C,P,P) f32."""
    feat = np.ascontiguousarray(feat, dtype=f32)
    rois = np.asarray(rois, dtype=f32)
    N, C, H, W = feat.shape
    R, P = rois.shape[0], int(out_size)
    out = np.zeros((R, C, P, P), dtype=f32)
    if R == 0:
        return out
    s = f32(spatial_scale)
    b = rois[:, 0].astype(np.int64)
    x1 = rois[:, 1] * s - f32(0.5)
    y1 = rois[:, 2] * s - f32(0.5)
    x2 = rois[:, 3] * s - f32(0.5)
    y2 = rois[:, 4] * s - f32(0.5)
    rw = (x2 - x1).astype(f32)
    rh = (y2 - y1).astype(f32)
    bw = (rw / f32(P)).astype(f32)
    bh = (rh / f32(P)).astype(f32)
    if sampling_ratio > 0:
        gh = np.full(R, sampling_ratio, np.int64)
        gw = np.full(R, sampling_ratio, np.int64)
    else:
        gh = np.ceil(rh / f32(P)).astype(np.int64)
        gw = np.ceil(rw / f32(P)).astype(np.int64)
    pidx = np.arange(P, dtype=f32)
    for key in np.unique(np.stack([gh, gw], 1), axis=0):
        kh, kw = int(key[0]), int(key[1])
        sel_all = np.nonzero((gh == kh) & (gw == kw))[0]
        if kh <= 0 or kw <= 0:
            continue  # count = max(gh*gw,1), empty sum -> zeros
        count = f32(max(kh * kw, 1))
        iy = np.arange(kh, dtype=f32)
        ix = np.arange(kw, dtype=f32)
        for c0 in range(0, len(sel_all), chunk):
            sel = sel_all[c0:c0 + chunk]
            r = len(sel)
            # y = y1 + ph*bin_h + (iy+.5)*bin_h/gh   (float32 ops in this order)
            ys = (y1[sel, None, None] + pidx[None, :, None] * bh[sel, None, None]
                  + (iy[None, None, :] + f32(0.5)) * bh[sel, None, None] / f32(kh)).astype(f32)
            xs = (x1[sel, None, None] + pidx[None, :, None] * bw[sel, None, None]
                  + (ix[None, None, :] + f32(0.5)) * bw[sel, None, None] / f32(kw)).astype(f32)
            vy, yl, yh, hy, ly = _axis_terms(ys, H)   # (r,P,kh)
            vx, xl, xh, hx, lx = _axis_terms(xs, W)   # (r,P,kw)
            fb = feat[b[sel]]                          # (r,C,H,W)
            ar = np.arange(r)[:, None, None, None, None]

            def g(yi, xi):
                # -> (r, P, kh, P, kw, C)
                return fb.transpose(0, 2, 3, 1)[ar, yi[:, :, :, None, None], xi[:, None, None, :, :]]
            w1 = (hy[:, :, :, None, None] * hx[:, None, None, :, :])[..., None]
            w2 = (hy[:, :, :, None, None] * lx[:, None, None, :, :])[..., None]
            w3 = (ly[:, :, :, None, None] * hx[:, None, None, :, :])[..., None]
            w4 = (ly[:, :, :, None, None] * lx[:, None, None, :, :])[..., None]
            val = w1 * g(yl, xl) + w2 * g(yl, xh) + w3 * g(yh, xl) + w4 * g(yh, xh)
            val = val * (vy[:, :, :, None, None] & vx[:, None, None, :, :])[..., None]
            # accumulate iy outer, ix inner, like the scalar kernel
            acc = np.zeros((r, P, P, C), dtype=f32)
            for a in range(kh):
                for c in range(kw):
                    acc += val[:, :, a, :, c, :]
            out[sel] = (acc / count).transpose(0, 3, 1, 2)
    return out


def nms(boxes, scores, iou_thr):
    """Greedy NMS, offset 0, suppress when IoU > thr (strict). Order: score desc, ties by lower index.

    Returns kept indices (int64) in descending-score order."""
    boxes = np.asarray(boxes, dtype=f32)
    scores = np.asarray(scores, dtype=f32)
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0,), np.int64)
    order = np.argsort(-scores, kind='stable')
    b = boxes[order]
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = ((x2 - x1) * (y2 - y1)).astype(f32)
    suppressed = np.zeros(n, bool)
    keep = []
    thr = f32(iou_thr)
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(i)
        j = slice(i + 1, n)
        iw = np.maximum(f32(0), np.minimum(x2[i], x2[j]) - np.maximum(x1[i], x1[j]))
        ih = np.maximum(f32(0), np.minimum(y2[i], y2[j]) - np.maximum(y1[i], y1[j]))
        inter = (iw * ih).astype(f32)
        ovr = inter / (areas[i] + areas[j] - inter)
        suppressed[j] |= ovr > thr
    return order[np.asarray(keep, np.int64)]


def batched_nms(boxes, scores, idxs, iou_thr):
    """mmcv batched_nms (class_agnostic=False, N < split_thr): coordinate-offset trick in float32.

    Returns (dets (K,5), keep (K,))."""
    boxes = np.asarray(boxes, dtype=f32)
    scores = np.asarray(scores, dtype=f32)
    if boxes.shape[0] == 0:
        return np.zeros((0, 5), f32), np.zeros((0,), np.int64)
    off = np.asarray(idxs).astype(f32) * (boxes.max() + f32(1))
    keep = nms(boxes + off[:, None], scores, iou_thr)
    return np.concatenate([boxes[keep], scores[keep, None]], 1), keep
