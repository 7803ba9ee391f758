"""TEST INFRASTRUCTURE ONLY (checker; never imported by the product).

The tissue front end of the slide path as ONE sequence, restated from the reference with the scalar OpenCV restatements of
oracle/cv_ops.py and the border-following oracle of oracle/contour.py (cv2 / openslide absent: PARITY UNPINNED against the libraries):

    segment_tissue      tools/wsi_core/WholeSlideImage.py:105-199  segmentTissue + _filter_contours
    contour_tile_coords tools/wsi_core/WholeSlideImage.py:407-493  process_contour + process_coord_candidate (:495-500),
                        isInHoles (:353-358), tools/wsi_core/util_classes.py:48-111 (basic / center / four_pt / four_pt_hard)
    tile_coords         WholeSlideImage.py:388-405 process_contours (contour after contour), parameters of tools/infer_wsi.py:385-389

Written against the reference, not against nuhtc_amd/tissue.py: per-point Python loops over `oracle.contour.find_contours_ccomp`
(CHAIN_APPROX_SIMPLE instead of _NONE: the same polygons, so the same areas, bounding boxes and point tests), followed by
`_filter_contours` statement by statement.  Round 5: the ORDER of the list is restated too -- RETR_CCOMP lists the outer borders in
the reverse of the order the raster scan finds them, each followed by its holes, newest first (oracle/contour.py) -- so the tile
list of a slide with several tissue regions is compared as ONE sequence."""
import numpy as np

from . import contour as OC
from . import cv_ops as C


def segment_tissue(img, scale, sthresh=8, sthresh_up=255, mthresh=7, close=4, use_otsu=False, a_t=100, a_h=16, max_n_holes=8, ref_patch_size=512,
                   level_scale=None):
    """-> [(outer contour, [hole contours])] in level-0 pixels ((n, 2) int arrays).
    level_scale: `img` IS the segmentation level of a pyramid file (`read_region((0, 0), seg_level, level_dim[seg_level])`, :159) and
    level_scale its (x, y) downsample pair from `_assertLevelDownsamples` (:378-386): thresholds from level_scale[0] (:176), contours back to
    level 0 by `np.array(cont * scale, dtype='int32')` (`scaleContourDim`, :371)."""
    if level_scale is not None:
        lvl = np.asarray(img)
        scale = float(level_scale[0])
    else:
        lvl = np.asarray(img)[::scale, ::scale]                 # the pyramid level `read_region` would return (:161)
    sat = C.saturation_u8(lvl.tolist())                         # img_hsv[:, :, 1] (:162)
    med = C.median_blur(sat, mthresh)                           # :163
    thr = C.otsu_threshold(med) if use_otsu else sthresh        # :166-169 (THRESH_BINARY: v > thr -> maxval)
    binary = [[sthresh_up if v > thr else 0 for v in row] for row in med]
    if close > 0:
        binary = C.morph_close(binary, close)                   # :172-174
    scaled_ref = round(ref_patch_size / scale)                  # :179-182
    a_t, a_h = a_t * scaled_ref, a_h * scaled_ref
    conts, hier = OC.find_contours_ccomp(np.asarray(binary) > 0, simple=True)       # :187
    hier = hier[:, 2:]                                                              # :188 -> [first_child, parent]
    area = lambda c: C.contour_area(c.tolist())
    filtered, all_holes = [], []
    for cont_idx in [i for i in range(len(conts)) if hier[i, 1] == -1]:             # :117-133, in list order
        holes = [j for j in range(len(conts)) if hier[j, 1] == cont_idx]
        a = area(conts[cont_idx]) - sum(area(conts[j]) for j in holes)
        if a == 0:
            continue
        if a_t < a:
            filtered.append(cont_idx)
            all_holes.append(holes)
    out = []
    for cont_idx, hole_ids in zip(filtered, all_holes):                             # :136-152
        hs = sorted([conts[j] for j in hole_ids], key=area, reverse=True)[:max_n_holes]     # stable: equal areas keep list order
        hs = [h for h in hs if area(h) > a_h]
        if level_scale is not None:
            up = lambda c: np.array(np.asarray(c, np.int64) * np.asarray(level_scale, np.float64), dtype='int32').astype(np.int64)
            out.append((up(conts[cont_idx]), [up(h) for h in hs]))
            continue
        out.append((np.asarray(conts[cont_idx], np.int64) * scale, [np.asarray(h, np.int64) * scale for h in hs]))   # scaleContourDim
    return out


def _bounding_rect(c):
    x0, y0, x1, y1 = c[:, 0].min(), c[:, 1].min(), c[:, 0].max(), c[:, 1].max()
    return int(x0), int(y0), int(x1 - x0 + 1), int(y1 - y0 + 1)


def contour_tile_coords(cont, holes, patch_size, step_size, contour_fn='four_pt', center_shift=0.5):
    """process_contour at patch_level 0 with use_padding=True: level-0 (x, y) origins, in the reference's candidate order
    (meshgrid indexing='ij': x outer, y inner)."""
    sx, sy, w, h = _bounding_rect(cont)
    poly = cont.tolist()
    shift = int(patch_size // 2 * center_shift)
    out = []
    for x in range(sx, sx + w, step_size):
        for y in range(sy, sy + h, step_size):
            cx, cy = x + patch_size // 2, y + patch_size // 2
            if contour_fn == 'basic':
                ok = C.point_polygon_test(poly, (x, y)) >= 0
            elif contour_fn == 'center':
                ok = C.point_polygon_test(poly, (cx, cy)) >= 0
            else:
                pts = [(cx - shift, cy - shift), (cx + shift, cy + shift), (cx + shift, cy - shift), (cx - shift, cy + shift)] if shift > 0 else [(cx, cy)]
                tests = [C.point_polygon_test(poly, p) >= 0 for p in pts]
                ok = all(tests) if contour_fn == 'four_pt_hard' else any(tests)
            if not ok:
                continue
            if any(C.point_polygon_test(hh.tolist(), (x + patch_size / 2, y + patch_size / 2)) > 0 for hh in holes):   # isInHoles
                continue
            out.append((x, y))
    return out if len(out) > 1 else []                           # `if len(results) > 1` (:476)


def tile_coords(img, patch_size, step_size, scale, **seg):
    """-> list over tissue contours of that contour's tile origins."""
    return [contour_tile_coords(c, hs, patch_size, step_size) for c, hs in segment_tissue(img, scale, **seg)]
