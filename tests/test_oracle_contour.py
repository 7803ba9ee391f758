"""Pins oracle/contour.py (the Suzuki-Abe restatement with OpenCV's conventions) with hand-worked cases, and checks the
product's host tracer (nuhtc_amd/contours.py) against it on random masks.  CPU only.

Expected vertex lists were worked out by hand from the border-following rules (and agree with what cv2.findContours is
known to return for these shapes: a rectangle comes out top-left, bottom-left, bottom-right, top-right; several blobs are
listed bottom-most first; a hole border is the octagon through the pixels 4-adjacent to the hole, clockwise)."""
import numpy as np
import pytest

from oracle import contour as OC


def _m(rows):
    return np.array([[c == '#' for c in r] for r in rows])


def test_rectangle_start_and_orientation():
    m = np.zeros((5, 6), bool); m[1:4, 1:5] = True
    c, h = OC.find_contours_tree(m)
    assert [x.tolist() for x in c] == [[[1, 1], [1, 3], [4, 3], [4, 1]]]      # TL, BL, BR, TR: down the left side first
    assert h.tolist() == [[-1, -1, -1, -1]]
    assert OC.mask2inst(m).tolist() == [[1, 1], [1, 3], [4, 3], [4, 1], [1, 1]]


def test_degenerate_shapes():
    assert OC.find_contours_tree(np.zeros((4, 4), bool))[0] == []
    one = np.zeros((3, 3), bool); one[1, 1] = True
    assert OC.find_contours_tree(one)[0][0].tolist() == [[1, 1]]
    line = _m(['......', '.####.', '......'])
    assert OC.find_contours_tree(line)[0][0].tolist() == [[1, 1], [4, 1]]     # out and back: the two end points
    diag = np.eye(4, dtype=bool)
    assert OC.find_contours_tree(diag)[0][0].tolist() == [[0, 0], [3, 3]]
    full = np.ones((3, 4), bool)                                              # touches every tile edge (zero frame is implicit)
    assert OC.find_contours_tree(full)[0][0].tolist() == [[0, 0], [0, 2], [3, 2], [3, 0]]


def test_one_pixel_wide_l_shape_cuts_the_inner_corner_on_the_way_back():
    m = _m(['#....',
            '#....',
            '#....',
            '####.',
            '.....'])
    # down the column, along the row, back west to (1,3), then the 8-connected step north-west to (0,2) and up
    assert OC.find_contours_tree(m)[0][0].tolist() == [[0, 0], [0, 3], [3, 3], [1, 3], [0, 2]]


def test_spur_is_walked_out_and_back():
    m = _m(['.......',
            '.###...',
            '.######',
            '.###...',
            '.......'])
    # 3x3 block with a one-pixel-wide spur to the east on its middle row: out along the spur to (6,2) and back
    assert OC.find_contours_tree(m)[0][0].tolist() == [[1, 1], [1, 3], [3, 3], [4, 2], [6, 2], [4, 2], [3, 1]]


def test_hole_and_nested_island_hierarchy():
    m = np.zeros((7, 7), bool); m[1:6, 1:6] = True; m[2:5, 2:5] = False; m[3, 3] = True
    c, h = OC.find_contours_tree(m)
    assert c[0].tolist() == [[1, 1], [1, 5], [5, 5], [5, 1]]                           # outer border of the ring
    assert c[1].tolist() == [[1, 2], [2, 1], [4, 1], [5, 2], [5, 4], [4, 5], [2, 5], [1, 4]]   # hole: clockwise octagon
    assert c[2].tolist() == [[3, 3]]                                                   # island inside the hole
    assert h.tolist() == [[-1, -1, 1, -1], [-1, -1, 2, 0], [-1, -1, -1, 1]]
    # [0][0] is the ring, not the island, although the island's first pixel comes later in the scan
    assert OC.mask2inst(m)[0].tolist() == [1, 1]


def test_several_blobs_are_listed_last_found_first():
    m = np.zeros((9, 9), bool); m[1:3, 1:3] = True; m[4:6, 5:8] = True; m[7, 0:2] = True
    c, h = OC.find_contours_tree(m)
    assert [x[0].tolist() for x in c] == [[0, 7], [5, 4], [1, 1]]
    assert h[:, 0].tolist() == [1, 2, -1] and h[:, 3].tolist() == [-1, -1, -1]
    # two blobs whose first pixels share a row: the right one is found later
    m = np.zeros((5, 9), bool); m[1:4, 1:3] = True; m[1:3, 5:8] = True
    assert OC.mask2inst(m)[0].tolist() == [5, 1]


def test_chain_approx_none_lists_every_border_pixel():
    m = np.zeros((5, 6), bool); m[1:4, 1:5] = True
    c, _ = OC.find_contours_tree(m, simple=False)
    assert len(c[0]) == 10 and c[0][0].tolist() == [1, 1] and c[0][1].tolist() == [1, 2]


def _random_masks(rng, n, size=24):
    from scipy import ndimage as ndi
    out = []
    for k in range(n):
        a = ndi.gaussian_filter(rng.standard_normal((size, size)), rng.uniform(0.8, 2.5))
        m = a > np.quantile(a, rng.uniform(0.5, 0.9))
        if k % 3 == 0:       # punch holes / add islands
            m &= ~(ndi.gaussian_filter(rng.standard_normal((size, size)), 1.0) > 0.25)
        if k % 4 == 0:       # sparse noise: many tiny components, pixels on the tile edge
            m |= rng.random((size, size)) > 0.93
        if k % 5 == 0:       # a ring around part of the picture: whatever lies inside its hole is nested, not top-level
            y0, x0 = rng.integers(0, size // 3, 2)
            y1, x1 = rng.integers(2 * size // 3, size, 2)
            m[y0:y1 + 1, x0:x1 + 1] &= rng.random((y1 - y0 + 1, x1 - x0 + 1)) > 0.8
            m[y0 + 1:y1, x0 + 1:x1] &= True
            m[y0, x0:x1 + 1] = m[y1, x0:x1 + 1] = True
            m[y0:y1 + 1, x0] = m[y0:y1 + 1, x1] = True
            m[y0 + 1, x0 + 1:x1] = m[y1 - 1, x0 + 1:x1] = False
            m[y0 + 1:y1, x0 + 1] = m[y0 + 1:y1, x1 - 1] = False
        out.append(m)
    return out


def test_host_tracer_matches_oracle_on_random_masks():
    from nuhtc_amd import contours as C
    rng = np.random.default_rng(3)
    masks = _random_masks(rng, 300)
    ring = np.zeros((9, 9), bool); ring[1:8, 1:8] = True; ring[2:7, 2:7] = False; ring[4, 4] = True; ring[8, 8] = True
    masks += [ring, np.ones((6, 6), bool), np.eye(7, dtype=bool), np.eye(7, dtype=bool)[::-1]]
    multi = nested = 0
    for m in masks:
        c, h = OC.find_contours_tree(m)
        got = C.trace_outer_contour(m)
        if not c:
            assert len(got) == 0
            continue
        multi += int((h[:, 3] == -1).sum() > 1)
        nested += int(any(h[k, 3] >= 0 and h[h[k, 3], 3] >= 0 for k in range(len(c))))
        assert np.array_equal(got, c[0]), (m.astype(int), got.tolist(), c[0].tolist())
    assert multi > 50 and nested > 5      # the sample really contains fragmented masks and islands inside holes
