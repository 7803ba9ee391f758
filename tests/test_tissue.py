"""CPU tests of the tissue segmentation / tissue-restricted tiling (restated OpenCV steps: parity unpinned, cv2 absent).
Behaviour is pinned on synthetic slides and on hand-worked values of the restated primitives."""
import numpy as np

from nuhtc_amd import tissue as T


def synthetic_slide(H=2048, W=3072):
    rng = np.random.default_rng(0)
    img = np.full((H, W, 3), 235, np.uint8)                       # glass: unsaturated
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    blob1 = ((yy - 750) / 550.0) ** 2 + ((xx - 900) / 650.0) ** 2 <= 1
    hole = ((yy - 750) / 260.0) ** 2 + ((xx - 900) / 330.0) ** 2 <= 1
    inner_hole = ((yy - 750) / 180.0) ** 2 + ((xx - 900) / 240.0) ** 2 <= 1
    blob2 = ((yy - 1450) / 350.0) ** 2 + ((xx - 2300) / 450.0) ** 2 <= 1
    speck = ((yy - 200) / 40.0) ** 2 + ((xx - 2800) / 40.0) ** 2 <= 1   # below the area filter
    tissue = (blob1 & ~hole) | blob2 | speck
    img[tissue] = (200, 120, 180)                                 # H&E-like: saturated
    img += rng.integers(0, 4, img.shape, dtype=np.uint8)
    return img, blob1, hole, blob2, inner_hole


def test_primitives_hand_worked():
    # saturation: v = 200, min = 120 -> (80 * round(255*4096/200) + 2048) >> 12 = 102
    assert T.saturation_u8(np.array([[[200, 120, 180]]], np.uint8))[0, 0] == 102
    assert T.saturation_u8(np.array([[[0, 0, 0]]], np.uint8))[0, 0] == 0
    assert T.saturation_u8(np.array([[[255, 0, 0]]], np.uint8))[0, 0] == 255
    sq = np.array([[0, 0], [4, 0], [4, 3], [0, 3]])
    assert T.contour_area(sq) == 12 and T.bounding_rect(sq) == (0, 0, 5, 4)
    assert T.point_polygon_test(sq, (2, 1)) == 1 and T.point_polygon_test(sq, (4, 2)) == 0 and T.point_polygon_test(sq, (5, 2)) == -1
    assert T.point_polygon_test(sq, (0, 0)) == 0
    # close(4) bridges a 3-pixel gap but not a 5-pixel one
    row = np.zeros((9, 24), np.uint8); row[4, :6] = 255; row[4, 9:14] = 255; row[4, 19:] = 255
    c = T.morph_close(row, 4)
    assert c[4, 6:9].all() and not c[4, 14:19].all()
    # bimodal histogram: Otsu lands between the modes
    im = np.concatenate([np.full(500, 20, np.uint8), np.full(300, 180, np.uint8)]).reshape(20, 40)
    assert 20 <= T.otsu_threshold(im) < 180
    # border following with holes: a 6x6 square with a 2x2 hole
    b = np.zeros((10, 10), np.uint8); b[2:8, 2:8] = 255; b[4:6, 4:6] = 0
    (outer, holes), = T.find_contours_ccomp(b)
    # the hole border visits the 8 foreground pixels 4-adjacent to the 2 x 2 hole and cuts its corners diagonally (Suzuki-Abe from the
    # hole's side; the four pixels diagonal to the hole's corners are not on it): (2 + 1)^2 - 4 * 1/2 = 7, not 9
    assert T.contour_area(outer) == 25 and len(outer) == 20 and len(holes) == 1 and T.contour_area(holes[0]) == 7 and len(holes[0]) == 8


def test_segment_and_tile_synthetic_slide():
    img, blob1, hole, blob2, inner_hole = synthetic_slide()
    coords, conts, holes = T.tissue_tile_coords(img, patch_size=256, step_size=192, scale=8)
    assert len(conts) == 2 and sorted(len(h) for h in holes) == [0, 1]          # the speck is filtered out, one hole found
    areas = sorted(T.contour_area(c) for c in conts)
    assert abs(areas[1] - blob1.sum()) / blob1.sum() < 0.03 and abs(areas[0] - blob2.sum()) / blob2.sum() < 0.03
    assert len(coords) > 40 and coords.dtype == np.int64
    tissue = (blob1 & ~hole) | blob2
    cy, cx = coords[:, 1] + 128, coords[:, 0] + 128
    H, W = tissue.shape
    # every kept tile has tissue at one of the four check points; no tile is centred deep inside the hole or on glass
    hit = np.zeros(len(coords), bool)
    for dy, dx in ((-64, -64), (64, 64), (-64, 64), (64, -64)):
        yy, xx = np.clip(cy + dy, 0, H - 1), np.clip(cx + dx, 0, W - 1)
        hit |= (blob1 | blob2)[yy, xx]
    assert hit.mean() > 0.97
    assert not inner_hole[np.clip(cy, 0, H - 1), np.clip(cx, 0, W - 1)].any()
    # coverage: almost every tissue pixel lies in some tile
    cover = np.zeros_like(tissue)
    for x, y in coords:
        cover[max(y, 0):y + 256, max(x, 0):x + 256] = True
    assert (cover & tissue).sum() / tissue.sum() > 0.97
    # grid structure: origins of one contour are its bounding-box corner + k * step
    x0, y0, _, _ = T.bounding_rect(conts[0])
    n0 = len(T.contour_coords(conts[0], holes[0], (W, H), 256, 192))
    assert ((coords[:n0] - [x0, y0]) % 192 == 0).all()
    # the stricter check keeps fewer tiles, the corner check differs from the centre check
    hard = T.contour_coords(conts[0], holes[0], (W, H), 256, 192, contour_fn='four_pt_hard')
    assert 0 < len(hard) < n0
    tiles = T.read_tiles(img, coords[:3], 256)
    assert tiles.shape == (3, 256, 256, 3) and np.array_equal(tiles[0], img[coords[0, 1]:coords[0, 1] + 256, coords[0, 0]:coords[0, 0] + 256])
    # Otsu instead of the fixed threshold finds the same two regions on this slide
    c2, h2 = T.segment_tissue(img, scale=8, use_otsu=True)
    assert len(c2) == 2


def test_padding_and_edges():
    img = np.full((1600, 1600, 3), 235, np.uint8)
    img[400:1600, 600:1600] = (190, 100, 170)                    # tissue touching the slide corner
    coords, conts, holes = T.tissue_tile_coords(img, 256, 192, scale=8)
    assert len(conts) == 1 and coords[:, 0].max() + 256 > 1600   # use_padding: tiles may run past the edge ...
    t = T.read_tiles(img, coords, 256)
    k = int(np.argmax(coords[:, 0] + coords[:, 1]))
    assert (t[k][-20:, -20:] == 0).all()                         # ... and are zero padded there
    nopad = T.contour_coords(conts[0], holes[0], (1600, 1600), 256, 192, use_padding=False)
    assert len(nopad) and (nopad[:, 0] + 256 <= 1600).all() and (nopad[:, 1] + 256 <= 1600).all()
    empty, c0, _ = T.tissue_tile_coords(np.full((512, 512, 3), 240, np.uint8), 256, 192, scale=8)
    assert len(empty) == 0 and len(c0) == 0


def tissue_slide_with_holes(H=1536, W=2048, seed=0):
    """Glass (grey 235) with two tissue regions textured like the synthetic H&E tiles; the first has a hole large enough to survive the
    a_h filter at a segmentation downsample of 8 (> 16 * 64 level pixels), plus a speck below the a_t filter."""
    from nuhtc_amd import synth
    img = np.full((H, W, 3), 235, np.uint8)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    blob1 = ((yy - 620) / 540.0) ** 2 + ((xx - 700) / 640.0) ** 2 <= 1
    hole = ((yy - 600) / 180.0) ** 2 + ((xx - 660) / 210.0) ** 2 <= 1
    blob2 = ((yy - 1110) / 400.0) ** 2 + ((xx - 1690) / 340.0) ** 2 <= 1
    speck = ((yy - 120) / 60.0) ** 2 + ((xx - 1900) / 60.0) ** 2 <= 1
    tissue = (blob1 & ~hole) | blob2 | speck
    tex = np.concatenate([np.concatenate([synth.nuclei_tile(500 + seed * 100 + 8 * r + c, 256) for c in range(W // 256)], 1) for r in range(H // 256)], 0)
    img[tissue] = tex[tissue]
    return img, blob1, hole, blob2


def test_tile_list_equals_the_scalar_restatement_of_the_reference_sequence():
    """nuhtc_amd.tissue.tissue_tile_coords (numpy / scipy) against oracle/tissue.py (segmentTissue + process_contour restated from the
    reference over the scalar OpenCV restatements and the Suzuki-Abe oracle): the same tissue contours (bounding boxes, areas, holes)
    and, per contour, the same tile origins in the same order."""
    from oracle import cv_ops as C
    from oracle import tissue as OT
    img, blob1, hole, blob2 = tissue_slide_with_holes()
    coords, conts, holes = T.tissue_tile_coords(img, patch_size=64, step_size=64, scale=8)
    ref = OT.segment_tissue(img, 8)
    assert len(conts) == len(ref) == 2 and sorted(len(h) for h in holes) == [0, 1] == sorted(len(h) for _, h in ref)
    # ONE sequence: cv2 lists the outer borders in the reverse of the order the raster scan finds them (oracle/contour.py
    # find_contours_ccomp), so blob2 (first pixel further down) comes first; contours, holes and tiles line up index by index
    assert int(conts[0][:, 1].min()) > int(conts[1][:, 1].min())
    off = 0
    for (c, hs), (rc, rhs) in zip(zip(conts, holes), ref):
        assert T.bounding_rect(c) == OT._bounding_rect(np.asarray(rc))
        assert T.contour_area(c) == C.contour_area(rc.tolist())
        assert [T.contour_area(h) for h in hs] == [C.contour_area(h.tolist()) for h in rhs]
        want = OT.contour_tile_coords(rc, rhs, 64, 64)
        got = coords[off:off + len(want)]
        off += len(want)
        assert len(want) > 50 and got.tolist() == [list(p) for p in want]
    assert off == len(coords)
    assert coords.tolist() == [list(p) for pts in OT.tile_coords(img, 64, 64, 8) for p in pts]
    # no tile is centred inside the hole, and the speck produced none
    assert not hole[np.clip(coords[:, 1] + 32, 0, 1535), np.clip(coords[:, 0] + 32, 0, 2047)].any()
    assert not ((coords[:, 0] > 1800) & (coords[:, 1] < 200)).any()


def test_ccomp_list_order_of_regions_and_holes():
    """RETR_CCOMP list order (every new border linked at the head of its parent's list): product == oracle on a mask with three
    components (one an island inside a hole) and two holes in one component."""
    from oracle import contour as OC
    m = np.zeros((12, 20), np.uint8)
    m[1:4, 1:4] = 1; m[2, 2] = 0                # A: found first, one 1-pixel hole
    m[5:11, 5:18] = 1                            # B
    m[6:10, 6:12] = 0                            # B's hole 1 (found first)
    m[7:9, 8:10] = 1                             # C: an island in hole 1 -- top level in CCOMP
    m[6:8, 14:16] = 0                            # B's hole 2
    conts, hier = OC.find_contours_ccomp(m, simple=False)
    tops = [i for i in range(len(conts)) if hier[i, 3] == -1]
    assert [tuple(conts[i][0]) for i in tops] == [(8, 7), (5, 5), (1, 1)]                      # C, B, A: newest first
    assert [tuple(conts[j][0]) for j in range(len(conts)) if hier[j, 3] == tops[1]] == [(13, 6), (5, 6)]   # hole 2, hole 1
    got = T.find_contours_ccomp(m * 255)
    assert [tuple(c[0]) for c, _ in got] == [(8, 7), (5, 5), (1, 1)]
    assert [[tuple(h[0]) for h in hs] for _, hs in got] == [[], [(13, 6), (5, 6)], [(1, 2)]]
    for (c, hs), i in zip(got, tops):
        assert c.tolist() == conts[i].tolist()
