"""Pins oracle/merge_poly.py (the reference's polygon-IoU merge, tools/nuclei_merge.py:37-133, with exact areas) with
hand-worked cases.  CPU only; the device merge is compared with it in tests/test_merge.py (-m gpu)."""
import numpy as np

from oracle import contour as OC
from oracle import merge_poly as MP


def sq(x, y, s):
    return np.array([[x, y], [x, y + s], [x + s, y + s], [x + s, y], [x, y]], np.int64)


def shoelace(ring):
    r = np.asarray(ring, np.float64)
    return 0.5 * abs(np.sum(r[:-1, 0] * r[1:, 1] - r[1:, 0] * r[:-1, 1]))


def test_polygon_areas_are_exact():
    assert MP.Poly(sq(0, 0, 10)).area8 == 800
    # a traced ring runs through the centres of the border pixels: an 11x11-pixel square gives a 10x10 polygon
    m = np.zeros((15, 15), bool); m[2:13, 2:13] = True
    assert MP.Poly(OC.mask2inst(m)).area8 == 800 and m.sum() == 121
    # octagon around a 3x3 hole's border pixels (diagonal edges): 4x4 square minus four corner triangles of 1/2
    octagon = np.array([[1, 2], [2, 1], [4, 1], [5, 2], [5, 4], [4, 5], [2, 5], [1, 4], [1, 2]])
    assert MP.Poly(octagon).area8 == 8 * 16 - 4 * 4
    # one-pixel-wide L: the ring walks back over itself; only the inner-corner triangle (0,2),(0,3),(1,3) has area
    L = np.array([[0, 0], [0, 3], [3, 3], [1, 3], [0, 2], [0, 0]])
    assert MP.Poly(L).area8 == 4
    # a straight line of pixels: zero area, IoU with anything is 0 (0/0 guarded)
    line = np.array([[1, 1], [4, 1], [1, 1]])
    assert MP.Poly(line).area8 == 0 and MP.polygon_iou(line, sq(0, 0, 5)) == 0.0


def test_invalid_ring_keeps_the_largest_part_only():
    # 3x3-pixel block and 4x4-pixel block joined only by the diagonal step (2,2)-(3,3): the ring pinches there, buffer(0)
    # returns two polygons (2x2 and 3x3), the reference keeps the larger one
    m = np.zeros((8, 8), bool); m[0:3, 0:3] = True; m[3:7, 3:7] = True
    ring = OC.mask2inst(m)
    assert MP.Poly(ring, largest_part=False).area8 == 8 * (4 + 9)
    assert MP.Poly(ring).area8 == 8 * 9
    # block with a one-pixel spur: the spur itself adds nothing, but its first pixel (4,2) is 8-adjacent to the block corners
    # (3,1) and (3,3), so the ring cuts across: 2x2 square + the triangle (3,1),(4,2),(3,3) of area 1
    m = np.zeros((6, 10), bool); m[1:4, 1:4] = True; m[2, 4:9] = True
    assert MP.Poly(OC.mask2inst(m)).area8 == 8 * 5
    # two lobes joined by a one-pixel-wide bridge: parts 2x2 + 1 (left) and 3x2 + 1 (right, with their corner triangles)
    m = np.zeros((6, 12), bool); m[1:4, 1:4] = True; m[2, 4:7] = True; m[1:4, 7:11] = True
    assert MP.Poly(OC.mask2inst(m), largest_part=False).area8 == 8 * 12
    assert MP.Poly(OC.mask2inst(m)).area8 == 8 * 7
    big = MP.Poly(OC.mask2inst(m))
    assert big.inter8(MP.Poly(sq(1, 1, 2))) == 0 and big.inter8(MP.Poly(sq(7, 1, 2))) == 8 * 4   # the left lobe is gone


def test_polygon_iou_known_answers():
    assert MP.polygon_iou(sq(0, 0, 10), sq(0, 0, 10)) == 1.0
    assert MP.polygon_iou(sq(0, 0, 10), sq(20, 20, 10)) == 0.0
    assert MP.polygon_iou(sq(0, 0, 10), sq(10, 0, 10)) == 0.0                 # sharing an edge: zero area
    assert MP.polygon_iou(sq(0, 0, 10), sq(5, 0, 10)) == 50 / 150
    # diamond (diagonal edges) against a square: diamond |x-5|+|y-5| <= 5 (area 50), square [0,5]x[0,5] cuts a triangle of 12.5
    dia = np.array([[5, 0], [0, 5], [5, 10], [10, 5], [5, 0]])
    assert MP.Poly(dia).area8 == 400
    assert MP.Poly(dia).inter8(MP.Poly(sq(0, 0, 5))) == 100
    # two diamonds shifted by one pixel cross at half-integer coordinates: intersection = hexagon, by hand 50 - 2*(5-0.25)...
    dib = dia + np.array([1, 0])
    inter = MP.Poly(dia).inter8(MP.Poly(dib))
    # overlap of |x-5|+|y-5|<=5 and |x-6|+|y-5|<=5: width 9 diamond-like band; area = 50 - 2 * (area of the strip cut off) = 40.5
    assert inter == 8 * 40.5
    # mask IoU and polygon IoU differ: two 11x11 squares offset by 10 px overlap in one pixel column (mask IoU 11/231) but
    # their polygons only touch (polygon IoU 0): exactly the borderline the reference's threshold sees differently
    a = np.zeros((11, 11), bool); a[:] = True
    ra, rb = OC.mask2inst(a), OC.mask2inst(a) + np.array([10, 0])
    assert MP.polygon_iou(ra, rb) == 0.0


def test_area_equals_shoelace_on_traced_rings():
    from scipy import ndimage as ndi
    rng = np.random.default_rng(11)
    n = 0
    for k in range(120):
        a = ndi.gaussian_filter(rng.standard_normal((40, 40)), rng.uniform(1.5, 4.0))
        m = a > np.quantile(a, rng.uniform(0.6, 0.92))
        ring = OC.mask2inst(m)
        if len(ring) < 4:
            continue
        # a border-following ring never crosses itself and all its lobes turn the same way, so the shoelace sum is the area of
        # the non-zero-winding region (spurs cancel): an independent formula for the same number
        assert MP.Poly(ring, largest_part=False).area8 == round(8 * shoelace(ring)), k
        n += 1
    assert n > 100


def test_merge_sequence_semantics():
    m = np.ones((11, 11), bool)
    masks = [(m, 0, 0), (m, 6, 0), (m, 12, 0)]
    # A(0.9) overlaps B(0.8); B overlaps C(0.7); A does not touch C: greedy keeps A and C (B is visited, cannot suppress C)
    assert MP.merge_overlap_masks(masks, [0.9, 0.8, 0.7], 0.05).tolist() == [0, 2]
    assert MP.merge_overlap_masks(masks[::-1], [0.7, 0.8, 0.9], 0.05).tolist() == [0, 2]
    assert MP.merge_overlap_masks([(m, 0, 0), (m, 2, 2)], [0.5, 0.5], 0.05).tolist() == [0]       # tie: lower index first
    # strict '>' at the threshold: 10x10 polygons shifted by 9 -> inter 10, union 190
    two = [(m, 0, 0), (m, 9, 0)]
    assert MP.merge_overlap_masks(two, [0.9, 0.8], 10 / 190).tolist() == [0, 1]
    assert MP.merge_overlap_masks(two, [0.9, 0.8], 0.05).tolist() == [0]
    assert MP.merge_overlap([], [], 0.05).tolist() == []


def test_geojson_merge_of_the_product_matches_oracle():
    """tools/nuclei_merge.py path of the product (nuhtc_amd.contours.merge_features on GeoJSON rings, exact quarter-cell regions)
    against the oracle on a random slide of traced rings, both strategies' shared part ('probability')."""
    from scipy import ndimage as ndi
    from nuhtc_amd import contours as C
    rng = np.random.default_rng(21)
    rings, scores, feats = [], [], []
    names = ['T', 'I', 'C', 'D', 'E']
    for k in range(260):
        a = ndi.gaussian_filter(rng.standard_normal((28, 28)), rng.uniform(1.2, 3.0))
        m = a > np.quantile(a, rng.uniform(0.6, 0.85))
        r = OC.mask2inst(m)
        if len(r) < 4:
            continue
        r = r + rng.integers(0, 160, 2)
        s = float(np.round(rng.uniform(0.35, 1.0), 2))
        rings.append(r); scores.append(s)
        feats.append(C.feature(r, k % 5, s, names))
    for thr in (0.01, 0.05, 0.4):
        ref = MP.merge_overlap(rings, scores, thr)
        got = C.merge_features(feats, thr, 'probability')
        order = np.argsort(-np.asarray(scores, np.float32), kind='stable')      # nuclei_id = rank in score order
        assert sorted(int(order[f['properties']['nuclei_id']]) for f in got) == ref.tolist(), thr
    assert 0 < len(ref) < len(rings)
