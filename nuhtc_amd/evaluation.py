"""Instance-segmentation metrics and dataset export formats for the engine's detections (host side, numpy).

Mirrors the evaluation surface the reference wraps around its detector:

* per-image statistics on lists of binary instance masks -- AJI, AJI+, DQ/SQ/PQ, ensemble Dice
  (reference: nuhtc/utils/stats_utils.py:34-240,438-455; driver nuhtc/datasets/WSI_coco.py:545-637 `stat_calc`,
  :639-658 `mutlti_stat_calc`, :487-522 the `multi_pq+` / `multi_pq` aggregation, :733-748 confusion matrix);
* the PanNuke evaluation protocol on label maps -- `remap_label`, `binarize`, `get_fast_pq_map`, per-class / per-tissue
  bPQ and mPQ (reference: tools/analysis_tools/pannuke/utils.py:7-160, compute_stats.py:66-170);
* the export formats of `WSIDataset.convert_format` (WSI_coco.py:863-906): 'pannuke' (H,W,C+1 instance maps),
  'consep' (inst_map / inst_type / centroids) and 'conic' (H,W,2).

Everything here consumes what `nuhtc_amd.engine.Engine.results` returns (boolean masks, labels, scores); nothing runs
on the GPU and nothing imports the oracle.
"""
import numpy as np
from scipy.optimize import linear_sum_assignment

PANNUKE_TISSUES = ['Adrenal_gland', 'Bile-duct', 'Bladder', 'Breast', 'Cervix', 'Colon', 'Esophagus', 'HeadNeck', 'Kidney',
                   'Liver', 'Lung', 'Ovarian', 'Pancreatic', 'Prostate', 'Skin', 'Stomach', 'Testis', 'Thyroid', 'Uterus']


# ----------------------------------------------------------------------------- mask-list statistics
def _flat(masks):
    m = np.asarray(masks)
    return m.reshape(m.shape[0], -1).astype(np.float64) if m.size else np.zeros((len(masks), 0))


def pairwise_inter_union(true_masks, pred_masks):
    """Intersection and union pixel counts of every (true, pred) pair (stats_utils.py:438-455), as one matrix product."""
    t, p = _flat(true_masks), _flat(pred_masks)
    inter = t @ p.T
    union = t.sum(1)[:, None] + p.sum(1)[None, :] - inter
    return inter, union


def _pairing(pairwise_iou, paired_true, paired_pred):
    if paired_true is None or paired_pred is None:
        paired_true, paired_pred = linear_sum_assignment(-pairwise_iou)
    return np.asarray(paired_true), np.asarray(paired_pred)


def get_fast_aji(true_masks, pred_masks, pairwise_inter=None, pairwise_union=None):
    """AJI as distributed by MoNuSeg: every true instance takes its best-IoU prediction, predictions may be reused
    (stats_utils.py:34-77)."""
    if pairwise_inter is None or pairwise_union is None:
        pairwise_inter, pairwise_union = pairwise_inter_union(true_masks, pred_masks)
    iou = pairwise_inter / (pairwise_union + 1.0e-6)
    best = np.argmax(iou, axis=1)
    best_iou = np.max(iou, axis=1)
    pt = np.nonzero(best_iou > 0.0)[0]
    pp = best[pt]
    inter = pairwise_inter[pt, pp].sum()
    union = pairwise_union[pt, pp].sum()
    t, p = _flat(true_masks), _flat(pred_masks)
    unpaired_t = np.setdiff1d(np.arange(len(t)), pt)
    unpaired_p = np.setdiff1d(np.arange(len(p)), pp)
    union += t[unpaired_t].sum() + p[unpaired_p].sum()
    return inter / union


def get_fast_aji_plus(true_masks, pred_masks, pairwise_inter=None, pairwise_union=None, paired_true=None, paired_pred=None):
    """AJI+ : one-to-one pairing (stats_utils.py:80-125)."""
    if pairwise_inter is None or pairwise_union is None:
        pairwise_inter, pairwise_union = pairwise_inter_union(true_masks, pred_masks)
    iou = pairwise_inter / (pairwise_union + 1.0e-6)
    pt, pp = _pairing(iou, paired_true, paired_pred)
    sel = iou[pt, pp] > 0.0
    pt, pp = pt[sel], pp[sel]
    inter = pairwise_inter[pt, pp].sum()
    union = pairwise_union[pt, pp].sum()
    t, p = _flat(true_masks), _flat(pred_masks)
    union += t[np.setdiff1d(np.arange(len(t)), pt)].sum() + p[np.setdiff1d(np.arange(len(p)), pp)].sum()
    return inter / union


def get_fast_pq(true_masks, pred_masks, pairwise_inter=None, pairwise_union=None, paired_true=None, paired_pred=None,
                match_iou=0.5):
    """[dq, sq, pq], [paired_true, paired_pred, unpaired_true, unpaired_pred] on mask lists (stats_utils.py:128-198)."""
    assert match_iou >= 0.0
    if pairwise_inter is None or pairwise_union is None:
        pairwise_inter, pairwise_union = pairwise_inter_union(true_masks, pred_masks)
    iou = pairwise_inter / (pairwise_union + 1.0e-6)
    pt, pp = _pairing(iou, paired_true, paired_pred)
    piou = iou[pt, pp]
    sel = piou > match_iou
    pt, pp, piou = list(pt[sel]), list(pp[sel]), piou[sel]
    unpaired_true = [i for i in range(len(true_masks)) if i not in pt]
    unpaired_pred = [i for i in range(len(pred_masks)) if i not in pp]
    tp, fp, fn = len(pt), len(unpaired_pred), len(unpaired_true)
    dq = tp / (tp + 0.5 * fp + 0.5 * fn)
    sq = piou.sum() / (tp + 1.0e-6)
    return [dq, sq, dq * sq], [pt, pp, unpaired_true, unpaired_pred]


def get_fast_dice(true_masks, pred_masks, pairwise_inter=None, pairwise_union=None, paired_true=None, paired_pred=None):
    """Ensemble Dice over the paired instances (stats_utils.py:202-240)."""
    if pairwise_inter is None or pairwise_union is None:
        pairwise_inter, pairwise_union = pairwise_inter_union(true_masks, pred_masks)
    iou = pairwise_inter / (pairwise_union + 1.0e-6)
    pt, pp = _pairing(iou, paired_true, paired_pred)
    ok = iou[pt, pp] >= 1e-4
    pt, pp = pt[ok], pp[ok]
    if len(pt) + len(pp) == 0:
        return 1
    inter = pairwise_inter[pt, pp].sum()
    total = (pairwise_union[pt, pp] + pairwise_inter[pt, pp]).sum()
    return 2 * inter / total


def stat_calc(true_masks, pred_masks, match_iou=0.5):
    """Per-image statistics dictionary of `WSIDataset.stat_calc` (WSI_coco.py:545-637). Masks are (n, H, W) arrays or
    lists of (H, W) arrays; returns None when both sides are empty."""
    nt, npred = len(true_masks), len(pred_masks)
    if nt == 0 and npred == 0:
        return None
    zero = dict(aji=0, aji_plus=0, dq=0, sq=0, pq=0, dice=0, precision=0, recall=0, tp=0, fp=0, fn=0, iou=0)
    if nt == 0:
        return dict(zero, fp=npred)
    if npred == 0:
        return dict(zero, fn=nt)
    inter, union = pairwise_inter_union(true_masks, pred_masks)
    iou = inter / union                     # maskUtils.iou: exact ratio (no epsilon) decides the pairing
    iou[iou <= match_iou] = 0.0
    paired_true, paired_pred = np.nonzero(iou)
    aji = get_fast_aji(true_masks, pred_masks, inter, union)
    aji_plus = get_fast_aji_plus(true_masks, pred_masks, inter, union, paired_true, paired_pred)
    pq = get_fast_pq(true_masks, pred_masks, inter, union, paired_true, paired_pred, match_iou)
    tp, fp, fn = len(pq[1][0]), len(pq[1][3]), len(pq[1][2])
    dice = get_fast_dice(true_masks, pred_masks, inter, union, paired_true, paired_pred)
    return dict(aji=aji, aji_plus=aji_plus, dq=pq[0][0], sq=pq[0][1], pq=pq[0][2], dice=dice,
                precision=tp / (tp + fp + 1e-9), recall=tp / (tp + fn + 1e-9), tp=tp, fp=fp, fn=fn,
                iou=pq[0][1] * (tp + 1e-6))


def multi_stat_calc(true_masks, pred_masks, gt_labels, pred_labels, num_classes, match_iou=0.5):
    """[tp, fp, fn, iou_sum] per class, NaN rows for classes absent on both sides (WSI_coco.py:639-658)."""
    true_masks, pred_masks = np.asarray(true_masks), np.asarray(pred_masks)
    gt_labels, pred_labels = np.asarray(gt_labels), np.asarray(pred_labels)
    out = []
    for c in range(num_classes):
        t = true_masks[gt_labels == c] if len(true_masks) else true_masks
        p = pred_masks[pred_labels == c] if len(pred_masks) else pred_masks
        info = stat_calc(t, p, match_iou)
        out.append([info['tp'], info['fp'], info['fn'], info['iou']] if info else [float('nan')] * 4)
    return out


def aggregate_mpq(mpq_info_list):
    """Dataset-level multi-class PQ from the per-image [tp, fp, fn, iou_sum] tables (WSI_coco.py:487-522):
    'multi_pq+' pools the counts over images before forming DQ·SQ per class, 'multi_pq' averages per-image PQ."""
    info = np.array(mpq_info_list, dtype='float')          # (images, classes, 4)
    res = {}
    tot = np.nansum(info, axis=0)
    plus = []
    for c in range(tot.shape[0]):
        tp, fp, fn, siou = tot[c]
        dq = tp / ((tp + 0.5 * fp + 0.5 * fn) + 1.0e-6)
        sq = siou / (tp + 1.0e-6)
        plus.append(dq * sq)
        res[f'multi_pq+_{c}'] = dq * sq
    res['multi_pq+'] = np.mean(plus)
    dq = info[:, :, 0] / (info[:, :, 0] + 0.5 * info[:, :, 1] + 0.5 * info[:, :, 2] + 1.0e-6)
    sq = info[:, :, 3] / (info[:, :, 0] + 1.0e-6)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', category=RuntimeWarning)   # a class absent from every image stays NaN
        per_class = np.nanmean(dq * sq, axis=0)
    for c in range(per_class.shape[0]):
        res[f'multi_pq_{c}'] = per_class[c]
    res['multi_pq'] = np.mean(per_class)
    return res


def update_confusion_matrix(confusion_matrix, true_masks, pred_masks, gt_labels, pred_labels, tp_iou_thr=0.5):
    """Adds one image to a (C+1, C+1) confusion matrix whose last row/column is background (WSI_coco.py:733-748)."""
    gt_labels = np.asarray(gt_labels, dtype=int)
    pred_labels = np.asarray(pred_labels, dtype=int)
    if len(true_masks) and len(pred_masks):
        inter, union = pairwise_inter_union(true_masks, pred_masks)
        ious = inter / union
    else:
        ious = np.zeros((len(true_masks), len(pred_masks)))
    matched = np.zeros(len(gt_labels), dtype=int)
    for i, dl in enumerate(pred_labels):
        hit = np.nonzero(ious[:, i] >= tp_iou_thr)[0]
        for j in hit:
            matched[j] += 1
            confusion_matrix[gt_labels[j], dl] += 1
        if len(hit) == 0:
            confusion_matrix[-1, dl] += 1
    for n, gl in zip(matched, gt_labels):
        if n == 0:
            confusion_matrix[gl, -1] += 1
    return confusion_matrix


def mask_nms(masks, scores, thr=0.9):
    """Greedy mask NMS in descending score order, IoU > thr suppresses (stats_utils.py:10-32 / WSI_coco.py:708-731;
    the dataset evaluation calls it with thr=0.05). Returns (kept masks in score order, their indices into the input)."""
    masks = np.asarray(masks)
    order = np.argsort(scores)[::-1]
    m = masks[order]
    if len(m) == 0:
        return m, order
    inter, union = pairwise_inter_union(m, m)
    iou = inter / np.maximum(union, 1.0)
    keep = np.ones(len(m), dtype=bool)
    for i in range(len(m)):
        if keep[i]:
            keep[i + 1:] &= ~(iou[i, i + 1:] > thr)
    return m[keep], order[keep]


def mask_post_process(pred_masks, min_area):
    """`_mask_post_process` (WSI_coco.py:246-276): drop instances below `min_area`, then instances that are (almost)
    entirely covered by another one. Returns (kept masks, boolean selection over the area-filtered list)."""
    pred_masks = np.asarray(pred_masks)
    if len(pred_masks) == 0:
        return pred_masks, None
    area = pred_masks.reshape(len(pred_masks), -1).sum(1)
    sel = area >= min_area
    if sel.sum() == 0:
        return pred_masks[sel], sel
    pm, area = pred_masks[sel], area[sel]
    flat = pm.reshape(len(pm), -1).astype(np.float64)
    overlap = flat @ flat.T - np.diag(area)
    ratio = (overlap / area).max(axis=1)
    keep = ratio < 0.999
    return pm[keep], keep


# ----------------------------------------------------------------------------- export formats
def convert_format(masks, labels, height, width, num_classes, data_format='conic'):
    """`WSIDataset.convert_format` (WSI_coco.py:863-906) on decoded masks: `masks` (n, H, W) bool/0-1, `labels` (n,).

    'pannuke' -> int (H, W, C+1): channel c holds the 1-based index (within class c) of the instance covering a pixel,
                 the last channel is background; 'consep' -> dict(inst_map, inst_type[, inst_centroid, inst_uid]);
                 anything else ('conic') -> int (H, W, 2) = (instance id, class + 1). Later instances win overlaps (max).
    """
    masks = np.asarray(masks).astype(int).reshape(-1, height, width)
    labels = np.asarray(labels, dtype=int)
    n = len(masks)
    if data_format == 'pannuke':
        out = np.zeros((height, width, num_classes + 1), dtype=int)
        if n == 0:
            return out
        for c in range(num_classes):
            m = masks[labels == c]
            if len(m) == 0:
                continue
            out[:, :, c] = np.max(m * np.arange(1, len(m) + 1).reshape(-1, 1, 1), axis=0)
        out[:, :, -1] = 1 - np.max(masks, axis=0)
        return out
    out = np.zeros((height, width, 2), dtype=int)
    if n:
        out[:, :, 0] = np.max(masks * np.arange(1, n + 1).reshape(-1, 1, 1), axis=0)
        out[:, :, 1] = np.max(masks * (labels + 1).reshape(-1, 1, 1), axis=0)
    if data_format != 'consep':
        return out
    mat = {'inst_map': out[:, :, 0], 'inst_type': out[:, :, 1]}
    if n:
        cent = np.zeros((n, 2))
        for i, m in enumerate(masks):      # centre of the (x, y, w, h) box, as maskUtils.toBbox gives it
            ys, xs = np.nonzero(m)
            if len(xs):
                cent[i] = (xs.min() + (xs.max() + 1 - xs.min()) / 2, ys.min() + (ys.max() + 1 - ys.min()) / 2)
        mat['inst_centroid'] = cent
        mat['inst_uid'] = np.array(range(1, n))     # (sic) the reference stops one short
    return mat


# ----------------------------------------------------------------------------- PanNuke protocol on label maps
def remap_label(pred, by_size=False):
    """Contiguous instance ids 1..n, order preserved unless `by_size` (pannuke/utils.py:107-137)."""
    ids = list(np.unique(pred))
    if 0 in ids:
        ids.remove(0)
    if len(ids) == 0:
        return pred
    if by_size:
        sizes = [(pred == i).sum() for i in ids]
        ids = [i for i, _ in sorted(zip(ids, sizes), key=lambda x: x[1], reverse=True)]
    out = np.zeros(pred.shape, np.int32)
    for k, i in enumerate(ids):
        out[pred == i] = k + 1
    return out


def binarize(x):
    """(H, W, C) per-class instance maps -> one instance map; later channels / ids overwrite (pannuke/utils.py:141-162)."""
    out = np.zeros([x.shape[0], x.shape[1]])
    count = 1
    for c in range(x.shape[2]):
        ch = x[:, :, c]
        vals = np.unique(ch).tolist()
        if 0 in vals:
            vals.remove(0)
        for v in vals:
            m = ch == v
            out *= 1 - m
            out += count * m
            count += 1
    return out.astype('int32')


def get_fast_pq_map(true, pred, match_iou=0.5):
    """PQ on two contiguous-id label maps (pannuke/utils.py:7-104). Returns [dq, sq, pq], pairing lists (1-based ids)."""
    assert match_iou >= 0.0
    true_ids = list(np.unique(true))
    pred_ids = list(np.unique(pred))
    nt, npd = len(true_ids) - 1, len(pred_ids) - 1
    # joint histogram of (true id, pred id) gives every intersection at once
    joint = np.zeros((nt + 1, npd + 1), dtype=np.float64)
    np.add.at(joint, (true.ravel().astype(np.int64), pred.ravel().astype(np.int64)), 1.0)
    inter = joint[1:, 1:]
    area_t = joint.sum(1)[1:]
    area_p = joint.sum(0)[1:]
    union = area_t[:, None] + area_p[None, :] - inter
    with np.errstate(divide='ignore', invalid='ignore'):
        pairwise_iou = np.where(inter > 0, inter / union, 0.0)
    if match_iou >= 0.5:
        pairwise_iou[pairwise_iou <= match_iou] = 0.0
        paired_true, paired_pred = np.nonzero(pairwise_iou)
        paired_iou = pairwise_iou[paired_true, paired_pred]
        paired_true = paired_true + 1
        paired_pred = paired_pred + 1
    else:
        pt, pp = linear_sum_assignment(-pairwise_iou)
        piou = pairwise_iou[pt, pp]
        paired_true = list(pt[piou > match_iou] + 1)
        paired_pred = list(pp[piou > match_iou] + 1)
        paired_iou = piou[piou > match_iou]
    unpaired_true = [i for i in true_ids[1:] if i not in paired_true]
    unpaired_pred = [i for i in pred_ids[1:] if i not in paired_pred]
    tp, fp, fn = len(paired_true), len(unpaired_pred), len(unpaired_true)
    dq = tp / (tp + 0.5 * fp + 0.5 * fn)
    sq = paired_iou.sum() / (tp + 1.0e-6)
    return [dq, sq, dq * sq], [paired_true, paired_pred, unpaired_true, unpaired_pred]


def pannuke_stats(true, pred, types, num_classes=5, tissue_types=PANNUKE_TISSUES):
    """The PanNuke split statistics of compute_stats.py:66-170. `true`, `pred`: (N, H, W, >=num_classes) per-class
    instance maps (the 'pannuke' export format), `types`: (N,) tissue names.

    Returns dict(class_pq=[...], tissue_mpq={...}, tissue_bpq={...}, mPQ=..., bPQ=...)."""
    import warnings
    mpq_all, bpq_all = [], []
    for i in range(true.shape[0]):
        pred_bin = remap_label(binarize(pred[i, :, :, :num_classes]))
        true_bin = binarize(true[i, :, :, :num_classes])
        pq_bin = np.nan if len(np.unique(true_bin)) == 1 else get_fast_pq_map(true_bin, pred_bin)[0][2]
        pq = []
        for c in range(num_classes):
            p = remap_label(pred[i, :, :, c].astype('int32'))
            t = remap_label(true[i, :, :, c].astype('int32'))
            pq.append(np.nan if len(np.unique(t)) == 1 else get_fast_pq_map(t, p)[0][2])
        mpq_all.append(pq)
        bpq_all.append([pq_bin])
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', category=RuntimeWarning)   # nanmean of all-NaN slices is NaN, as in the reference
        mpq_img = [np.nanmean(p) for p in mpq_all]
        bpq_img = [np.nanmean(p) for p in bpq_all]
        class_pq = [np.nanmean([p[c] for p in mpq_all]) for c in range(num_classes)]
        t_mpq, t_bpq = {}, {}
        for name in tissue_types:
            idx = [i for i, x in enumerate(types) if x == name]
            t_mpq[name] = np.nanmean([mpq_img[i] for i in idx]) if idx else np.nan
            t_bpq[name] = np.nanmean([bpq_img[i] for i in idx]) if idx else np.nan
        res = dict(class_pq=class_pq, tissue_mpq=t_mpq, tissue_bpq=t_bpq,
                   mPQ=np.nanmean(list(t_mpq.values())), bPQ=np.nanmean(list(t_bpq.values())))
    return res
