#!/usr/bin/env python
"""Run the engine over a PanNuke-format fold and score / export it.

    python tools/test_pannuke.py <config> <checkpoint> --images fold/images.npy [--masks fold/masks.npy --types fold/types.npy]
                                 [--out infer/pannuke] [--batch 16] [--format pannuke|conic|consep]

What it replaces in the reference: `tools/test.py ... --eval segm --eval-options save=True format=pannuke`, i.e.
`WSIDataset.evaluate` (nuhtc/datasets/WSI_coco.py:278-545: score filter 0.1, mask-NMS 0.05, `stat_calc`,
`mutlti_stat_calc`, `convert_format`, preds_<format>.npy) followed by tools/analysis_tools/pannuke/compute_stats.py
(bPQ / mPQ per class and tissue, class_stats.csv / tissue_stats.csv).  The COCO-json dataset plumbing is not rebuilt:
the fold is read in PanNuke's own array format -- images (N,256,256,3), masks (N,256,256,6) per-class instance maps with the
background in the last channel, types (N,) tissue names.
"""
import argparse
import csv
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nuhtc_amd import evaluation as E  # noqa: E402
from nuhtc_amd.apis import concat_results, inference_detector, init_detector  # noqa: E402


def gt_instances(mask, num_classes):
    """(H,W,C+1) per-class instance maps -> (masks (n,H,W) bool, labels (n,))."""
    ms, ls = [], []
    for c in range(num_classes):
        for v in np.unique(mask[:, :, c]):
            if v != 0:
                ms.append(mask[:, :, c] == v)
                ls.append(c)
    h, w = mask.shape[:2]
    return (np.stack(ms) if ms else np.zeros((0, h, w), bool)), np.array(ls, dtype=int)


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument('config')
    p.add_argument('checkpoint')
    p.add_argument('--images', required=True)
    p.add_argument('--masks', default=None)
    p.add_argument('--types', default=None)
    p.add_argument('--out', default='infer/pannuke')
    p.add_argument('--device', default='cuda:0')
    p.add_argument('--batch', type=int, default=16)
    p.add_argument('--format', default='pannuke', choices=['pannuke', 'conic', 'consep'])
    p.add_argument('--fg-thr', type=float, default=0.1, help='score filter of WSIDataset.evaluate')
    p.add_argument('--mask-nms-thr', type=float, default=0.05)
    return p.parse_args()


def main():
    a = parse_args()
    images = np.load(a.images)
    if images.dtype != np.uint8:
        images = np.clip(images, 0, 255).astype(np.uint8)
    masks = np.load(a.masks) if a.masks else None
    types = np.load(a.types) if a.types else None
    model = init_detector(a.config, a.checkpoint, device=a.device, max_batch=a.batch)
    nc = int(model.opts['num_classes'])
    os.makedirs(a.out, exist_ok=True)
    N, H, W = images.shape[:3]
    preds, stats, mpq_info = [], {}, []
    cm = np.zeros((nc + 1, nc + 1))
    for i0 in range(0, N, a.batch):
        # PanNuke images are RGB arrays; tools/test.py reads files through LoadImageFromFile (BGR -> to_rgb), so the
        # network sees true RGB: same channel handling as file input here
        batch = [np.ascontiguousarray(images[i][..., ::-1]) for i in range(i0, min(N, i0 + a.batch))]
        results = inference_detector(model, batch)          # ndarray input = "BGR" branch, swapped back to RGB inside
        for k, res in enumerate(results):
            boxes, labels, pm = concat_results(res)
            sel = boxes[:, 4] >= a.fg_thr
            boxes, labels, pm = boxes[sel], labels[sel], (pm[sel] if len(pm) else np.zeros((0, H, W), bool))
            if len(pm):
                pm, keep = E.mask_nms(pm, boxes[:, 4], thr=a.mask_nms_thr)
                labels = labels[keep]
            preds.append(E.convert_format(pm, labels, H, W, nc, a.format))
            if masks is not None:
                tm, tl = gt_instances(masks[i0 + k], nc)
                s = E.stat_calc(tm, pm)
                if s:
                    for key, v in s.items():
                        stats.setdefault(key, []).append(v)
                mpq_info.append(E.multi_stat_calc(tm, pm, tl, labels, nc))
                E.update_confusion_matrix(cm, tm, pm, tl, labels)
    if a.format != 'consep':
        np.save(os.path.join(a.out, f'preds_{a.format}.npy'), np.array(preds))
    summary = {}
    if masks is not None:
        summary.update({k: float(np.mean(v)) for k, v in stats.items() if k not in ('tp', 'fp', 'fn', 'iou')})
        summary.update({k: float(v) for k, v in E.aggregate_mpq(mpq_info).items()})
        np.save(os.path.join(a.out, 'confusion_matrix.npy'), cm)
        if a.format == 'pannuke' and types is not None:
            r = E.pannuke_stats(masks, np.array(preds), list(types), num_classes=nc)
            summary['mPQ'], summary['bPQ'] = float(r['mPQ']), float(r['bPQ'])
            with open(os.path.join(a.out, 'class_stats.csv'), 'w', newline='') as f:
                w = csv.writer(f)
                w.writerow(['', 'Class Name', 'PQ'])
                for j, (n, v) in enumerate(zip(['Neoplastic', 'Inflam', 'Connective', 'Dead', 'Non-Neoplastic'], r['class_pq'])):
                    w.writerow([j, n, v])
            with open(os.path.join(a.out, 'tissue_stats.csv'), 'w', newline='') as f:
                w = csv.writer(f)
                w.writerow(['', 'Tissue name', 'PQ', 'PQ bin'])
                for j, n in enumerate(E.PANNUKE_TISSUES):
                    w.writerow([j, n, r['tissue_mpq'][n], r['tissue_bpq'][n]])
                w.writerow([len(E.PANNUKE_TISSUES), 'mean', r['mPQ'], r['bPQ']])
    with open(os.path.join(a.out, 'summary.json'), 'w') as f:
        json.dump(summary, f, indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == '__main__':
    main()
