"""The tissue-segmentation primitives of the product (nuhtc_amd/tissue.py: numpy / scipy) against independent scalar restatements of
the OpenCV routines (oracle/cv_ops.py), on hand-worked values and random inputs.  cv2 is absent: parity with the library itself is
unpinned; what is pinned is that two separately written readings of the OpenCV sources agree."""
import numpy as np

from nuhtc_amd import tissue as T
from oracle import cv_ops as O
from oracle import contour as OC


def test_hand_worked_values():
    # saturation: v = 200, min = 120: sdiv[200] = round(1044480 / 200) = 5222; (80 * 5222 + 2048) >> 12 = 102
    assert O.saturation_u8([[[200, 120, 180]]])[0][0] == 102 == T.saturation_u8(np.array([[[200, 120, 180]]], np.uint8))[0, 0]
    assert O.saturation_u8([[[7, 7, 7]]])[0][0] == 0 and O.saturation_u8([[[255, 0, 9]]])[0][0] == 255
    # median of a 3 x 3 window with replicated border at the corner: [9,9,1, 9,9,1, 4,4,2] -> sorted middle = 4
    img = [[9, 1, 5], [4, 2, 7], [3, 8, 6]]
    assert O.median_blur(img, 3)[0][0] == 4
    # close(4): the dilation reaches 1 left / 2 right, the erosion 2 left / 1 right: a 3-pixel gap closes, a 5-pixel one does not
    row = np.zeros((9, 24), np.uint8); row[4, :6] = 255; row[4, 9:14] = 255; row[4, 19:] = 255
    c = np.array(O.morph_close(row.tolist(), 4))
    assert c[4, 6:9].all() and not c[4, 14:19].all()
    # shoelace of a 4 x 3 rectangle and the boundary / inside / outside cases of pointPolygonTest
    sq = [[0, 0], [4, 0], [4, 3], [0, 3]]
    assert O.contour_area(sq) == 12.0
    assert [O.point_polygon_test(sq, p) for p in ((2, 1), (4, 2), (5, 2), (0, 0), (2, 3), (2, 4), (-1, 0))] == [1, 0, -1, 0, 0, -1, -1]


def test_primitives_agree_on_random_inputs():
    rng = np.random.default_rng(0)
    for _ in range(6):
        h, w = rng.integers(5, 24, 2)
        rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        sat = T.saturation_u8(rgb)
        assert np.array_equal(sat, np.array(O.saturation_u8(rgb.tolist()), np.uint8))
        for k in (3, 5, 7):
            assert np.array_equal(T.median_blur(sat, k), np.array(O.median_blur(sat.tolist(), k), np.uint8))
        binary = np.where(rng.uniform(size=(h, w)) < 0.4, 255, 0).astype(np.uint8)
        for k in (2, 3, 4, 5):
            assert np.array_equal(T.morph_close(binary, k), np.array(O.morph_close(binary.tolist(), k), np.uint8)), k
        assert T.otsu_threshold(sat) == O.otsu_threshold(sat.tolist())
    # degenerate histograms: one value, two values
    for img in (np.full((4, 4), 7, np.uint8), np.array([[0, 0, 255, 255]], np.uint8), np.array([[3, 3, 3, 200]], np.uint8)):
        assert T.otsu_threshold(img) == O.otsu_threshold(img.tolist())


def test_polygon_tests_agree():
    rng = np.random.default_rng(1)
    polys = [[[0, 0], [10, 0], [10, 10], [0, 10]], [[0, 0], [8, 0], [8, 3], [3, 3], [3, 8], [0, 8]], [[5, 0], [10, 5], [5, 10], [0, 5]],
             [[0, 0], [6, 0], [6, 6], [4, 6], [4, 2], [2, 2], [2, 6], [0, 6]]]
    for poly in polys:
        pts = rng.integers(-2, 13, (300, 2))
        got = T.points_polygon_test(np.array(poly), pts)
        want = [O.point_polygon_test(poly, p) for p in pts.tolist()]
        assert got.tolist() == want
        assert T.contour_area(poly) == O.contour_area(poly)


def test_borders_with_holes_agree_with_the_suzuki_abe_oracle():
    """find_contours_ccomp (component labelling + Moore trace) against the border-following oracle (oracle/contour.py, the RETR_TREE
    hierarchy): same number of outer borders and holes, and per outer border the same enclosed area and hole areas."""
    rng = np.random.default_rng(2)
    for trial in range(5):
        m = np.zeros((40, 48), np.uint8)
        for _ in range(4):
            y, x = rng.integers(2, 30), rng.integers(2, 36)
            h, w = rng.integers(5, 10), rng.integers(5, 12)
            m[y:y + h, x:x + w] = 255
            if h > 6 and w > 6:
                m[y + 2:y + h - 2, x + 2:x + w - 2] = 0
        got = T.find_contours_ccomp(m)
        conts, hier = OC.find_contours_tree(m > 0)
        outer = [i for i, hh in enumerate(hier) if hh[3] < 0]
        holes = [i for i, hh in enumerate(hier) if hh[3] >= 0 and hier[hh[3]][3] < 0]
        assert len(got) == len(outer) and sum(len(hs) for _, hs in got) == len(holes)
        assert sorted(T.contour_area(c) for c, _ in got) == sorted(O.contour_area(conts[i].tolist()) for i in outer)
        assert sorted(T.contour_area(h) for _, hs in got for h in hs) == sorted(O.contour_area(conts[i].tolist()) for i in holes)
