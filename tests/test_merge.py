"""Cross-tile merge: the sequential oracle on hand-made cases (CPU) and the HIP kernel against the oracle,
bit for bit, on random detection sets (GPU)."""
import numpy as np
import pytest


def disc(r):
    yy, xx = np.mgrid[-r:r + 1, -r:r + 1]
    return (yy * yy + xx * xx) <= r * r


def random_slide(rng, n, size, dup=0.35, rmin=4, rmax=14, quant=None):
    """n nuclei on a size x size canvas; a share of them is duplicated with a small shift (tile-overlap duplicates)."""
    masks, scores = [], []
    for _ in range(n):
        r = int(rng.integers(rmin, rmax))
        m = disc(r) & (rng.uniform(size=(2 * r + 1, 2 * r + 1)) < 0.93)
        x0, y0 = int(rng.integers(0, size - 2 * r - 1)), int(rng.integers(0, size - 2 * r - 1))
        s = float(rng.uniform(0.35, 1.0))
        if quant:
            s = round(s * quant) / quant          # exact score ties
        masks.append((m, x0, y0)); scores.append(s)
        if rng.uniform() < dup:
            dx, dy = int(rng.integers(-3, 4)), int(rng.integers(-3, 4))
            m2 = disc(r) & (rng.uniform(size=(2 * r + 1, 2 * r + 1)) < 0.93)
            s2 = float(rng.uniform(0.35, 1.0))
            if quant:
                s2 = round(s2 * quant) / quant
            masks.append((m2, max(x0 + dx, 0), max(y0 + dy, 0))); scores.append(s2)
    return dict(score=scores, mask=masks)


def test_oracle_merge_known_cases():
    from oracle.merge import merge_overlap
    m = np.ones((10, 10), bool)
    # A(0.9) overlaps B(0.8); B overlaps C(0.7); A does not touch C: greedy keeps A and C (B is dead, cannot suppress C)
    rec = dict(score=[0.9, 0.8, 0.7], mask=[(m, 0, 0), (m, 6, 0), (m, 12, 0)])
    assert merge_overlap(rec, 0.05).tolist() == [0, 2]
    # order of the list does not matter, only the scores
    rec = dict(score=[0.7, 0.8, 0.9], mask=[(m, 12, 0), (m, 6, 0), (m, 0, 0)])
    assert merge_overlap(rec, 0.05).tolist() == [0, 2]
    # exact tie: the lower index wins
    rec = dict(score=[0.5, 0.5], mask=[(m, 0, 0), (m, 2, 2)])
    assert merge_overlap(rec, 0.05).tolist() == [0]
    # boxes overlap but the masks do not
    a = np.zeros((10, 10), bool); a[:, :4] = True
    b = np.zeros((10, 10), bool); b[:, 6:] = True
    rec = dict(score=[0.9, 0.8], mask=[(a, 0, 0), (b, 0, 0)])
    assert merge_overlap(rec, 0.05).tolist() == [0, 1]
    # IoU exactly at the threshold does not suppress (strict >)
    a = np.ones((1, 20), bool); b = np.ones((1, 20), bool)          # shifted by 18: inter 2, union 38 -> 1/19
    rec = dict(score=[0.9, 0.8], mask=[(a, 0, 0), (b, 18, 0)])
    assert merge_overlap(rec, 2 / 38).tolist() == [0, 1] and merge_overlap(rec, 0.05).tolist() == [0]
    assert merge_overlap(dict(score=[], mask=[]), 0.05).tolist() == []


def test_pack_masks_layout():
    from nuhtc_amd.wsi import pack_masks
    m = np.zeros((2, 35), bool); m[0, 0] = m[0, 33] = m[1, 34] = True
    boxes, areas, bits, off = pack_masks([(m, 5, 7), (np.ones((1, 1), bool), 0, 0)])
    assert boxes.tolist() == [[5, 7, 40, 9], [0, 0, 1, 1]] and areas.tolist() == [3, 1] and off.tolist() == [0, 4]
    assert bits.tolist() == [1, 2, 0, 4, 1]


@pytest.mark.gpu
def test_device_merge_matches_oracle(hip_device):
    from nuhtc_amd import wsi
    from oracle.merge import merge_overlap as oracle_merge
    rng = np.random.default_rng(7)
    for n, size, quant in ((1, 64, None), (40, 200, None), (600, 1200, None), (2500, 2600, 20), (3000, 900, None)):
        rec = random_slide(rng, n, size, quant=quant)
        ref = oracle_merge(rec, 0.05)
        got = wsi.merge_overlap(rec, 0.05)
        assert np.array_equal(ref, got), (n, len(ref), len(got))
        assert 0 < len(got) <= len(rec['score'])
    # other thresholds, including "suppress on any overlap" and "never"
    rec = random_slide(rng, 800, 900)
    for thr in (0.0, 0.3, 0.999):
        assert np.array_equal(oracle_merge(rec, thr), wsi.merge_overlap(rec, thr))
    # empty crops are never kept; empty input
    rec = dict(score=[0.9, 0.8], mask=[(np.zeros((0, 0), bool), 5, 5), (np.ones((3, 3), bool), 5, 5)])
    assert wsi.merge_overlap(rec, 0.05).tolist() == [1] == oracle_merge(rec, 0.05).tolist()
    assert wsi.merge_overlap(dict(score=[], mask=[]), 0.05).tolist() == []


@pytest.mark.gpu
def test_device_merge_slide_scale_properties(hip_device):
    """2e5 detections (a slide's worth): the kept set is independent (no kept pair above the threshold among neighbours)
    and maximal in the greedy sense is checked on a sample; a second call is bit-identical."""
    import time
    from nuhtc_amd import wsi
    rng = np.random.default_rng(11)
    rec = random_slide(rng, 150000, 19264, dup=0.4)
    t = time.time()
    keep = wsi.merge_overlap(rec, 0.05)
    dt = time.time() - t
    assert np.array_equal(keep, wsi.merge_overlap(rec, 0.05))
    n = len(rec['score'])
    assert 0.5 * n < len(keep) < n
    print(f'{n} detections -> {len(keep)} kept in {dt:.2f} s (incl. host packing)')
    kept = np.zeros(n, bool); kept[keep] = True
    boxes = np.array([[m[1], m[2], m[1] + m[0].shape[1], m[2] + m[0].shape[0]] for m in rec['mask']])
    # sample detections and test them against everything their box touches
    sc = np.asarray(rec['score'], np.float32)
    for i in rng.choice(n, 300, replace=False):
        near = np.nonzero((boxes[:, 0] < boxes[i, 2]) & (boxes[:, 2] > boxes[i, 0]) & (boxes[:, 1] < boxes[i, 3]) & (boxes[:, 3] > boxes[i, 1]))[0]
        sup_by_kept = False
        for j in near:
            if j == i:
                continue
            x0, y0 = max(boxes[i, 0], boxes[j, 0]), max(boxes[i, 1], boxes[j, 1])
            x1, y1 = min(boxes[i, 2], boxes[j, 2]), min(boxes[i, 3], boxes[j, 3])
            mi, mj = rec['mask'][i][0], rec['mask'][j][0]
            inter = np.logical_and(mi[y0 - boxes[i, 1]:y1 - boxes[i, 1], x0 - boxes[i, 0]:x1 - boxes[i, 0]],
                                   mj[y0 - boxes[j, 1]:y1 - boxes[j, 1], x0 - boxes[j, 0]:x1 - boxes[j, 0]]).sum()
            iou = inter / (mi.sum() + mj.sum() - inter)
            if iou > 0.05:
                assert not (kept[i] and kept[j])
                if kept[j] and (sc[j] > sc[i] or (sc[j] == sc[i] and j < i)):
                    sup_by_kept = True
        assert kept[i] != sup_by_kept          # dropped exactly when a kept, higher-priority neighbour overlaps it
