"""Cross-tile merge: the sequential oracle on hand-made cases (CPU) and the HIP kernel against the oracle,
bit for bit, on random detection sets (GPU)."""
import os

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


def _special_shapes():
    """Masks whose ring polygon differs from the pixel set in kind: fragments (only the component cv2 lists first counts),
    holes with islands, a pinched blob (buffer(0) splits it, the largest part counts), one-pixel bridges and spurs, lines."""
    out = []
    m = np.zeros((20, 30), bool); m[2:9, 2:9] = True; m[12:18, 15:28] = True; out.append(m)                    # two fragments
    m = np.zeros((24, 24), bool); m[1:23, 1:23] = True; m[5:19, 5:19] = False; m[9:15, 9:15] = True; out.append(m)   # ring + island
    m = np.zeros((16, 16), bool); m[0:6, 0:6] = True; m[6:15, 6:15] = True; out.append(m)                      # pinch on a diagonal step
    m = np.zeros((12, 30), bool); m[1:11, 1:9] = True; m[5, 9:18] = True; m[1:11, 18:29] = True; out.append(m)  # bridge between lobes
    m = np.zeros((9, 20), bool); m[1:8, 1:8] = True; m[4, 8:19] = True; out.append(m)                           # spur
    m = np.zeros((3, 17), bool); m[1, 1:16] = True; out.append(m)                                               # a line: no area
    m = np.ones((14, 37), bool); out.append(m)                                                                   # crop wider than a word
    m = np.zeros((40, 70), bool); m[5:35, 3:67] = True; m[10:30, 10:60] = False; m[15:25, 30:40] = True; m[0:3, 60:70] = True; out.append(m)
    return out


@pytest.mark.gpu
def test_device_merge_matches_oracle(hip_device):
    """Both overlap measures against their sequential oracles, keep sets bit for bit: 'polygon' (the reference's: IoU of the
    ring polygons, oracle/merge_poly.py on rings traced by oracle/contour.py) and 'mask' (pixel sets, oracle/merge.py)."""
    from nuhtc_amd import wsi
    from oracle.merge import merge_overlap as oracle_mask
    from oracle.merge_poly import merge_overlap_masks as oracle_poly
    rng = np.random.default_rng(7)
    n_diff = 0
    for n, size, quant in ((1, 64, None), (40, 200, None), (600, 1200, None), (1500, 2000, 20), (1500, 700, None)):
        rec = random_slide(rng, n, size, quant=quant)
        ref = oracle_mask(rec, 0.05)
        got = wsi.merge_overlap(rec, 0.05, overlap='mask')
        assert np.array_equal(ref, got), ('mask', n, len(ref), len(got))
        refp = oracle_poly(rec['mask'], rec['score'], 0.05)
        gotp = wsi.merge_overlap(rec, 0.05, overlap='polygon')
        assert np.array_equal(refp, gotp), ('polygon', n, len(refp), len(gotp), np.setxor1d(refp, gotp)[:10])
        assert 0 < len(gotp) <= len(rec['score'])
        n_diff += len(np.setxor1d(ref, refp))
    print('keep-set entries on which mask IoU and polygon IoU disagree over these slides:', n_diff)
    # shapes where polygon and pixel set differ in kind, shifted copies of each other at every small offset
    shapes = _special_shapes()
    masks, scores = [], []
    for k, m in enumerate(shapes):
        for j, (dx, dy) in enumerate(((0, 0), (1, 0), (3, 2), (7, 5), (-2, 9), (12, -3))):
            masks.append((m, 300 * k + 50 + dx, 100 + dy)); scores.append(0.9 - 0.01 * j - 0.001 * k)
        masks.append((m.T.copy(), 300 * k + 52, 98)); scores.append(0.5)
    for thr in (0.0, 0.01, 0.05, 0.3):
        refp = oracle_poly(masks, scores, thr)
        gotp = wsi.merge_overlap(dict(score=scores, mask=masks), thr, overlap='polygon')
        assert np.array_equal(refp, gotp), (thr, refp.tolist(), gotp.tolist())
    # other thresholds, including "suppress on any overlap" and "never"
    rec = random_slide(rng, 800, 900)
    for thr in (0.0, 0.3, 0.999):
        assert np.array_equal(oracle_mask(rec, thr), wsi.merge_overlap(rec, thr, overlap='mask'))
        assert np.array_equal(oracle_poly(rec['mask'], rec['score'], thr), wsi.merge_overlap(rec, thr, overlap='polygon'))
    # empty crops are never kept; empty input
    rec = dict(score=[0.9, 0.8], mask=[(np.zeros((0, 0), bool), 5, 5), (np.ones((3, 3), bool), 5, 5)])
    assert wsi.merge_overlap(rec, 0.05, overlap='mask').tolist() == [1] == oracle_mask(rec, 0.05).tolist()
    assert wsi.merge_overlap(dict(score=[], mask=[]), 0.05).tolist() == []


@pytest.mark.gpu
def test_device_merge_dense_clump_spills(hip_device):
    """Hundreds of detections on one spot: every one has far more than 24 higher-scored overlapping neighbours (the inline
    suppressor list), and the 1000-detection case also overruns the first spill pool so the pass is repeated with a larger
    one.  The keep set still equals the sequential oracle's."""
    from nuhtc_amd import wsi
    from oracle.merge import merge_overlap as oracle_mask
    from oracle.merge_poly import merge_overlap_masks as oracle_poly
    rng = np.random.default_rng(5)
    for n, mode in ((300, 'polygon'), (1000, 'mask')):
        masks = [(disc(int(rng.integers(6, 12))), int(rng.integers(100, 125)), int(rng.integers(100, 125))) for _ in range(n)]
        scores = rng.uniform(0.35, 1.0, n).round(3).tolist()
        rec = dict(score=scores, mask=masks)
        for thr in (0.05, 0.6):
            ref = oracle_poly(masks, scores, thr) if mode == 'polygon' else oracle_mask(rec, thr)
            got = wsi.merge_overlap(rec, thr, overlap=mode)
            assert np.array_equal(ref, got), (n, mode, thr, len(ref), len(got))


@pytest.mark.gpu
def test_device_merge_slide_scale_properties(hip_device):
    """2e5 detections (a slide's worth): the kept set is independent (no kept pair above the threshold among neighbours)
    and maximal in the greedy sense is checked on a sample; a second call is bit-identical."""
    import time
    from nuhtc_amd import wsi
    rng = np.random.default_rng(11)
    rec = random_slide(rng, 150000, 19264, dup=0.4)
    t = time.time()
    keep = wsi.merge_overlap(rec, 0.05, overlap='mask')
    dt = time.time() - t
    assert np.array_equal(keep, wsi.merge_overlap(rec, 0.05, overlap='mask'))
    t = time.time()
    keep_p = wsi.merge_overlap(rec, 0.05, overlap='polygon')
    print(f'polygon overlap: {len(keep_p)} kept in {time.time() - t:.2f} s (incl. host packing); differs from the mask-IoU keep set in '
          f'{len(np.setxor1d(keep, keep_p))} of {len(rec["score"])} detections')
    assert np.array_equal(keep_p, wsi.merge_overlap(rec, 0.05, overlap='polygon'))
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


@pytest.mark.gpu
def test_geojson_file_merged_on_the_gpu_equals_the_host_polygon_path(hip_device, tmp_path):
    """tools/nuclei_merge.py on a written GeoJSON: the rings (traced outer borders, as tools/infer_wsi.py writes them) are filled back into mask
    crops (nuhtc_fill_rings) and merged by nuhtc_merge_overlap with the reference's polygon measure -- the SAME kept features, in the same order,
    with the same nuclei_ids as the host's exact polygon code (contours.merge_features: shapely's semantics restated), on a slide of fragmented
    discs with tile-overlap duplicates and exact score ties, plus one-pixel, line, pinched and holed shapes; the CLI picks the GPU by itself,
    `--host` forces the polygon code, and a file with a user-drawn (non-traced) polygon falls back to it."""
    import json
    import subprocess
    import sys
    from nuhtc_amd import contours
    rng = np.random.default_rng(11)
    rec = random_slide(rng, 1500, 1400, quant=50)
    extra = [np.ones((1, 1), bool), np.ones((1, 9), bool), np.eye(7, dtype=bool), np.ones((6, 6), bool)]
    pinch = np.zeros((8, 8), bool); pinch[:4, :4] = True; pinch[4:7, 4:7] = True                     # two squares (4 x 4, 3 x 3) that touch in a point
    ring_m = np.ones((9, 9), bool); ring_m[3:6, 3:6] = False                                          # a hole
    for m in extra + [pinch, ring_m]:
        for k in range(3):
            rec['mask'].append((m, 300 + 2 * k, 300 + k)); rec['score'].append(0.5 + 0.01 * k)
    classes = ('T', 'I', 'C', 'D', 'E')
    feats = []
    for (m, x0, y0), s in zip(rec['mask'], rec['score']):
        if not m.any():
            continue
        c = contours.trace_outer_contour(m)
        feats.append(contours.feature(np.concatenate([c, c[:1]]) + np.array([x0, y0]), int(rng.integers(0, 5)), s, classes))
    host = contours.merge_features(feats, 0.05)
    dev = contours.merge_features_device(feats, 0.05)
    assert dev is not None and len(host) == len(dev) and 600 < len(host) < len(feats)
    assert json.dumps(host) == json.dumps(dev)
    src = tmp_path / 'slide.geojson'
    json.dump(feats, open(src, 'w'))
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools/nuclei_merge.py')
    out = subprocess.run([sys.executable, tool, '--geojson', str(src), '--overlap_threshold', '0.05'], check=True, capture_output=True, text=True).stdout
    assert 'GPU' in out
    a = open(tmp_path / 'slide_merged.geojson').read()
    out = subprocess.run([sys.executable, tool, '--geojson', str(src), '--overlap_threshold', '0.05', '--host', '--output_name', 'h'], check=True,
                         capture_output=True, text=True).stdout
    assert 'host polygons' in out and open(tmp_path / 'h.geojson').read() == a == json.dumps(host)
    drawn = feats[:50] + [contours.feature(np.array([[0, 0], [40, 3], [17, 29], [0, 0]]), 0, 0.9, classes)]      # not a traced ring
    assert contours.merge_features_device(drawn, 0.05) is None
