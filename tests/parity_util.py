"""Strict instance-level comparison of two `(bbox_results, segm_results)` tuples (reference vs HIP engine).

The north star's bar is "IoU >= 0.999 per instance, identical class ids".  The path holds thresholded / greedy decisions
(score > 0.35, NMS IoU > 0.5, the max_per_img cut, mask probability >= 0.5), so two fp32 implementations may legitimately
disagree on an instance or a pixel whose deciding quantity sits on its threshold.  This module requires exact agreement
and accepts a disagreement only with the proof that it is such a case:

  instance present on one side only  ->  its score is within EPS of the score threshold, or it overlaps an instance of
                                         the other side (same class) at IoU within EPS of the NMS threshold, or it
                                         overlaps (IoU > nms) another instance that is itself present on one side only
                                         (its suppressor was the flipped decision), or the count sits at max_per_img and
                                         its score is within EPS of the last kept score
  mask pixel that differs            ->  the pasted probability at that pixel (oracle's paste of the mask probabilities)
                                         is within EPS_MASK of the 0.5 threshold

Everything tolerated is returned in the report so the tests print it; nothing is averaged away.
"""
import numpy as np

EPS = 2e-5        # score / IoU distance from a threshold that explains a flipped decision (measured fp32 differences: ~1e-6)
EPS_MASK = 1e-4   # mask probabilities of the two sides agree to 1e-4 (the stage tolerance of tests/test_hip_full.py): a pixel can flip
                  # only where the pasted probability is that close to 0.5 (measured on MI355X: 1e-6 .. 2e-5)


def box_iou(a, b):
    x1 = np.maximum(a[:, None, 0], b[None, :, 0]); y1 = np.maximum(a[:, None, 1], b[None, :, 1])
    x2 = np.minimum(a[:, None, 2], b[None, :, 2]); y2 = np.minimum(a[:, None, 3], b[None, :, 3])
    inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
    aa = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]); ab = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / np.maximum(aa[:, None] + ab[None, :] - inter, 1e-12)


def flatten(res):
    """(bbox_results, segm_results) -> boxes (n,5), labels (n,), masks list (class-major order, as the reference concatenates)."""
    bb, sg = res
    boxes = np.concatenate(bb, 0) if len(bb) else np.zeros((0, 5), np.float32)
    labels = np.concatenate([np.full(len(b), c) for c, b in enumerate(bb)]) if len(bb) else np.zeros(0, int)
    masks = [m for cl in sg for m in cl]
    return boxes, labels, masks


def compare_strict(ref, got, score_thr=0.35, nms_iou=0.5, max_per_img=500, values=None, values_side='ref'):
    """-> (report, failures).  `values`: optional (n_side, H, W) pasted probabilities aligned with the flattened instances of
    `values_side`; without it any differing mask pixel is a failure.
    report: n_ref, n_got, matched, explained (list of strings), mask_px_flipped (total), masks_below_0999 (count),
            min_mask_iou, max_px_threshold_dist."""
    rb, rl, rm = flatten(ref)
    gb, gl, gm = flatten(got)
    rep = dict(n_ref=len(rb), n_got=len(gb), matched=0, explained=[], mask_px_flipped=0, masks_below_0999=0, min_mask_iou=1.0,
               max_px_threshold_dist=0.0)
    fails = []
    pairs = []
    used = set()
    if len(rb) and len(gb):
        iou = box_iou(rb[:, :4], gb[:, :4])
        same = rl[:, None] == gl[None, :]
        miou = np.where(same, iou, -1.0)
        for i in np.argsort(-rb[:, 4], kind='stable'):
            j = int(np.argmax(miou[i]))
            if miou[i, j] >= 0.999 and j not in used and abs(rb[i, 4] - gb[j, 4]) < 1e-3:
                used.add(j)
                pairs.append((int(i), j))
    rep['matched'] = len(pairs)
    un_r = sorted(set(range(len(rb))) - {i for i, _ in pairs})
    un_g = sorted(set(range(len(gb))) - used)

    def explain(side, k):
        mine, ml, other, ol, un_other = (rb, rl, gb, gl, un_g) if side == 'ref' else (gb, gl, rb, rl, un_r)
        s = float(mine[k, 4])
        if abs(s - score_thr) <= EPS:
            return f'score {s:.7f} on the {score_thr} threshold'
        if max(len(rb), len(gb)) >= max_per_img and len(other) and abs(s - float(other[:, 4].min())) <= EPS:
            return f'score {s:.7f} on the max_per_img cut'
        if len(other):
            io = box_iou(mine[k:k + 1, :4], other[:, :4])[0]
            io = np.where(ol == ml[k], io, -1.0)
            near = np.nonzero(np.abs(io - nms_iou) <= EPS)[0]
            if len(near):
                return f'IoU {float(io[near[0]]):.7f} with a same-class instance, on the NMS threshold {nms_iou}'
            for q in un_other:
                if io[q] > nms_iou:
                    return f'suppressed by an instance (IoU {float(io[q]):.4f}) that itself exists on one side only'
        return None
    for side, lst in (('ref', un_r), ('got', un_g)):
        for k in lst:
            why = explain(side, k)
            arr, lab = (rb, rl) if side == 'ref' else (gb, gl)
            desc = f'{side}-only instance class {int(lab[k])} box {np.round(arr[k, :4], 2).tolist()} score {arr[k, 4]:.6f}'
            if why is None:
                fails.append(desc + ': no threshold explains it')
            else:
                rep['explained'].append(desc + ': ' + why)
    if rm and gm:
        for i, j in pairs:
            a, b = rm[i], gm[j]
            diff = a != b
            nd = int(diff.sum())
            if nd == 0:
                continue
            uni = int(np.logical_or(a, b).sum())
            v = (uni - nd) / uni if uni else 1.0
            rep['min_mask_iou'] = min(rep['min_mask_iou'], v)
            rep['masks_below_0999'] += v < 0.999
            rep['mask_px_flipped'] += nd
            if values is None:
                fails.append(f'instance ref#{i}/got#{j}: {nd} mask pixels differ and no probabilities were given to explain them')
                continue
            val = values[i] if values_side == 'ref' else values[j]
            dist = float(np.abs(val[diff] - 0.5).max())
            rep['max_px_threshold_dist'] = max(rep['max_px_threshold_dist'], dist)
            if dist > EPS_MASK:
                fails.append(f'instance ref#{i}/got#{j}: {nd} mask pixels differ, farthest pasted probability {dist:.2e} from 0.5 (> {EPS_MASK})')
            else:
                rep['explained'].append(f'instance ref#{i}/got#{j} (mask IoU {v:.5f}): {nd} pixel(s) differ, pasted probability within {dist:.1e} of 0.5')
    elif len(rm) != len(gm) and pairs:
        fails.append(f'mask lists differ in length: {len(rm)} vs {len(gm)}')
    return rep, fails


def explain_proposal(row, other, nms_iou=0.7, min_size=10.0, cap=1000):
    """An RPN proposal row (x1,y1,x2,y2,score) of one implementation that has no partner in `other`: returns the threshold
    that explains it (min_bbox_size, the NMS IoU, the max_per_img cut) or None."""
    w, h = row[2] - row[0], row[3] - row[1]
    if abs(w - min_size) <= 1e-4 or abs(h - min_size) <= 1e-4:
        return f'side {min(w, h):.5f} on the min_bbox_size threshold'
    if len(other):
        io = box_iou(row[None, :4], other[:, :4])[0]
        k = int(np.argmin(np.abs(io - nms_iou)))
        if abs(io[k] - nms_iou) <= EPS:
            return f'IoU {io[k]:.7f} with another proposal, on the NMS threshold {nms_iou}'
        if len(other) >= cap and abs(row[4] - other[:, 4].min()) <= EPS:
            return f'score {row[4]:.7f} on the max_per_img cut'
    return None


def oracle_paste_values(O, inter, ori_hw, scale=2.0):
    """Pasted mask probabilities (D_total,H,W) of an oracle run's intermediates (`keep=True`), per tile, in the oracle's
    detection order (NMS order) -> list per tile of (values (n,H,W), labels (n,)) re-ordered class-major like its results."""
    import torch
    out = []
    off = 0
    for d, l in zip(inter['dets'], inter['labels']):
        n = d.shape[0]
        if n:
            _, vals = O.paste_masks(inter['mask_prob'][off:off + n], (d[:, :4] * scale) / scale, ori_hw[0], ori_hw[1], return_values=True)
            order = torch.argsort(l, stable=True).numpy()
            out.append(vals[order])
        else:
            out.append(np.zeros((0,) + tuple(ori_hw), np.float32))
        off += n
    return out


def fmt(rep):
    return (f"ref {rep['n_ref']} hip {rep['n_got']} matched {rep['matched']}; mask pixels flipped {rep['mask_px_flipped']} "
            f"(masks below IoU 0.999: {rep['masks_below_0999']}, min {rep['min_mask_iou']:.5f}, farthest from 0.5: {rep['max_px_threshold_dist']:.1e}); "
            f"explained: {len(rep['explained'])}")
