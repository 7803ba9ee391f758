"""Generate golden vectors for nuhtc_amd.evaluation by running the REFERENCE's own metric code (build container only).

    python -m oracle.ref_harness.make_eval_golden       # writes tests/golden/eval_*.npz

Imported from /root/reference (read-only):
  * tools/analysis_tools/pannuke/utils.py        (get_fast_pq on label maps, remap_label, binarize) -- pure numpy/scipy
  * nuhtc/utils/stats_utils.py                   (get_fast_aji / aji_plus / pq / dice, get_pairwise_iou on mask lists)
    -- its cv2 / pycocotools imports are satisfied by empty stub modules (none of the functions used here touch them)
The per-split protocol of tools/analysis_tools/pannuke/compute_stats.py:97-170 needs docopt + files on disk, so its loop
is driven here on the same reference functions (get_fast_pq / remap_label / binarize) and np.nanmean, line for line in
behaviour, to produce the expected class / tissue numbers.

Inputs are seeded synthetic instance sets (discs with overlaps; predictions = jittered / dropped / added copies).
Test infrastructure: only tests/ consume the fixtures.
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(__file__), '..', '..', 'tests', 'golden')


def _load(path, name, stubs=()):
    for s in stubs:
        if s not in sys.modules:
            m = types.ModuleType(s)
            m.__path__ = []
            sys.modules[s] = m
    if 'pycocotools' in sys.modules and not hasattr(sys.modules['pycocotools'], 'mask'):
        sys.modules['pycocotools'].mask = types.ModuleType('pycocotools.mask')
        sys.modules['pycocotools.mask'] = sys.modules['pycocotools'].mask
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def disc_instances(rng, n, size, rmin=3, rmax=8):
    yy, xx = np.mgrid[0:size, 0:size]
    out = []
    for _ in range(n):
        cy, cx = rng.uniform(0, size, 2)
        ry, rx = rng.uniform(rmin, rmax, 2)
        m = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
        if m.sum() > 0:
            out.append(m)
    return np.array(out, dtype=np.uint8).reshape(-1, size, size)


def perturb(rng, masks, drop=0.2, add=2, shift=2):
    size = masks.shape[-1]
    out = []
    for m in masks:
        if rng.uniform() < drop:
            continue
        dy, dx = rng.integers(-shift, shift + 1, 2)
        out.append(np.roll(np.roll(m, dy, 0), dx, 1))
    extra = disc_instances(rng, add, size)
    out = np.array(out, dtype=np.uint8).reshape(-1, size, size)
    return np.concatenate([out, extra]) if len(extra) else out


def label_map(masks):
    """instance list -> label map, later instances on top (ids are list positions + 1, so ids can be non-contiguous after overlap)."""
    lm = np.zeros(masks.shape[1:], np.int32)
    for i, m in enumerate(masks):
        lm[m > 0] = i + 1
    return lm


def main():
    pn = _load(f'{REF}/tools/analysis_tools/pannuke/utils.py', 'ref_pannuke_utils')
    su = _load(f'{REF}/nuhtc/utils/stats_utils.py', 'ref_stats_utils', stubs=('cv2', 'pycocotools'))
    os.makedirs(OUT, exist_ok=True)

    # ---- mask-list statistics (stats_utils.py) on several images
    store = {}
    rng = np.random.default_rng(11)
    n_img = 6
    for i in range(n_img):
        t = disc_instances(rng, int(rng.integers(1, 14)), 64)
        p = perturb(rng, t)
        store[f'true{i}'], store[f'pred{i}'] = t, p
        inter, union = su.get_pairwise_iou(list(t), list(p))
        store[f'inter{i}'], store[f'union{i}'] = inter, union
        iou = inter / union if inter.size else inter
        # the dataset driver's pairing (WSI_coco.py:593-597): exact IoU > 0.5
        th = np.where(iou > 0.5, iou, 0.0)
        pt, pp = np.nonzero(th)
        store[f'aji{i}'] = np.array(su.get_fast_aji(list(t), list(p), pairwise_inter=inter, pairwise_union=union))
        store[f'aji_plus{i}'] = np.array(su.get_fast_aji_plus(list(t), list(p), inter, union, pt, pp))
        pq = su.get_fast_pq(list(t), list(p), inter, union, pt, pp, match_iou=0.5)
        store[f'pq{i}'] = np.array(pq[0])
        store[f'pq_counts{i}'] = np.array([len(pq[1][0]), len(pq[1][3]), len(pq[1][2])])
        store[f'dice{i}'] = np.array(float(su.get_fast_dice(list(t), list(p), inter, union, pt, pp)))
        # Munkres pairing variants (paired_* = None)
        store[f'pq_munkres{i}'] = np.array(su.get_fast_pq(list(t), list(p), inter, union, match_iou=0.3)[0])
    store['n_img'] = np.array(n_img)
    np.savez_compressed(os.path.join(OUT, 'eval_masklist.npz'), **store)

    # ---- PanNuke protocol on label maps (pannuke/utils.py + compute_stats.py loop)
    store = {}
    rng = np.random.default_rng(12)
    N, S, C = 8, 64, 5
    true = np.zeros((N, S, S, C + 1), np.int32)
    pred = np.zeros((N, S, S, C + 1), np.int32)
    tissues = ['Breast', 'Colon', 'Breast', 'Lung', 'Colon', 'Skin', 'Lung', 'Breast']
    for i in range(N):
        for c in range(C):
            if rng.uniform() < 0.3:
                continue
            t = disc_instances(rng, int(rng.integers(1, 6)), S)
            p = perturb(rng, t, add=1)
            true[i, :, :, c] = label_map(t)
            pred[i, :, :, c] = label_map(p) * 3          # non-contiguous ids on purpose
    store['true'], store['pred'], store['types'] = true, pred, np.array(tissues)
    mPQ_all, bPQ_all = [], []
    per_img_pairs = []
    for i in range(N):
        pq = []
        pred_bin = pn.remap_label(pn.binarize(pred[i, :, :, :C]))
        true_bin = pn.binarize(true[i, :, :, :C])
        if i == 0:
            store['binarize0'] = true_bin
            store['remap0'] = pred_bin
        pq_bin = np.nan if len(np.unique(true_bin)) == 1 else pn.get_fast_pq(true_bin, pred_bin)[0][2]
        for j in range(C):
            p = pn.remap_label(pred[i, :, :, j].astype('int32'))
            t = pn.remap_label(true[i, :, :, j].astype('int32'))
            if len(np.unique(t)) == 1:
                pq.append(np.nan)
            else:
                r = pn.get_fast_pq(t, p)
                pq.append(r[0][2])
                per_img_pairs.append([i, j, len(r[1][0]), len(r[1][3]), len(r[1][2])])
        mPQ_all.append(pq)
        bPQ_all.append([pq_bin])
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', category=RuntimeWarning)
        mPQ_img = [np.nanmean(p) for p in mPQ_all]
        bPQ_img = [np.nanmean(p) for p in bPQ_all]
        store['class_pq'] = np.array([np.nanmean([p[c] for p in mPQ_all]) for c in range(C)])
        names = ['Breast', 'Colon', 'Lung', 'Skin']
        store['tissue_names'] = np.array(names)
        store['tissue_mpq'] = np.array([np.nanmean([mPQ_img[i] for i, x in enumerate(tissues) if x == n]) for n in names])
        store['tissue_bpq'] = np.array([np.nanmean([bPQ_img[i] for i, x in enumerate(tissues) if x == n]) for n in names])
    store['mpq_all'] = np.array(mPQ_all, dtype=np.float64)
    store['bpq_all'] = np.array(bPQ_all, dtype=np.float64)
    store['pairs'] = np.array(per_img_pairs)
    # a low-threshold (Munkres) call on one pair of maps
    t0 = pn.remap_label(true[0, :, :, :C].max(-1))
    p0 = pn.remap_label(pred[0, :, :, :C].max(-1))
    store['t0'], store['p0'] = t0, p0
    store['pq_munkres'] = np.array(pn.get_fast_pq(t0, p0, match_iou=0.3)[0])
    np.savez_compressed(os.path.join(OUT, 'eval_pannuke.npz'), **store)
    print('wrote eval_masklist.npz, eval_pannuke.npz')


if __name__ == '__main__':
    main()
