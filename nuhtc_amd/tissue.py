"""Tissue segmentation and tissue-restricted tile coordinates for the WSI path (host side, numpy/scipy).

Reference: tools/wsi_core/WholeSlideImage.py:105-199 `segmentTissue` (HSV saturation -> median blur -> threshold ->
morphological close -> contours with holes -> area filter), :407-493 `process_contour` (grid over the contour's bounding box,
kept where the contour check passes and the patch centre is not inside a hole), tools/wsi_core/util_classes.py:48-111 (the
`basic` / `center` / `four_pt` / `four_pt_hard` checks), parameters of tools/infer_wsi.py:385-389
(sthresh 8, mthresh 7, close 4, a_t 100, a_h 16, max_n_holes 8, use_padding, four_pt).

The reference does all of this with OpenCV on an OpenSlide pyramid level; neither library is available here, so the OpenCV
steps are restated (8-bit RGB->HSV saturation in OpenCV's fixed point, medianBlur with replicated borders, rectangular
close anchored at size//2, border following for outer contours and holes, shoelace contourArea, boundingRect,
pointPolygonTest) -- PARITY UNPINNED against cv2 itself; checked against the independent scalar restatements of the same OpenCV
routines in oracle/cv_ops.py and the Suzuki-Abe oracle (tests/test_oracle_cv.py), plus behaviour on synthetic slides.  Slides are arrays: the
segmentation level is a strided view of the level-0 image (`scale` = its downsample factor).
"""
import numpy as np
from scipy import ndimage

from .contours import trace_outer_contour

SEG_PARAMS = dict(sthresh=8, sthresh_up=255, mthresh=7, close=4, use_otsu=False)      # tools/infer_wsi.py:385-386
FILTER_PARAMS = dict(a_t=100, a_h=16, max_n_holes=8)                                  # :387


# ----------------------------------------------------------------------------- OpenCV restatements
def saturation_u8(rgb):
    """S channel of cv2.cvtColor(img, COLOR_RGB2HSV) for uint8 input: (v - min) * sdiv_table[v] in 12-bit fixed point,
    sdiv_table[v] = round((255 << 12) / v)."""
    a = np.asarray(rgb)[..., :3].astype(np.int64)
    v = a.max(-1)
    diff = v - a.min(-1)
    sdiv = np.zeros(256, np.int64)
    sdiv[1:] = np.rint((255 << 12) / np.arange(1, 256)).astype(np.int64)
    return ((diff * sdiv[v] + (1 << 11)) >> 12).astype(np.uint8)


def median_blur(img, k):
    """cv2.medianBlur(img, k): k x k median, replicated borders."""
    return ndimage.median_filter(img, size=k, mode='nearest')


def otsu_threshold(img):
    """Threshold value of cv2.threshold(..., THRESH_OTSU) on uint8: maximises the between-class variance."""
    hist = np.bincount(img.reshape(-1), minlength=256).astype(np.float64)
    total = hist.sum()
    mu = (hist * np.arange(256)).sum() / total
    best, best_t, q1, mu1 = 0.0, 0, 0.0, 0.0
    for t in range(256):
        p = hist[t] / total
        mu1 = (q1 * mu1 + t * p) / (q1 + p) if q1 + p > 0 else 0.0
        q1 += p
        q2 = 1.0 - q1
        if min(q1, q2) < 1e-12 or max(q1, q2) > 1.0 - 1e-12:
            continue
        mu2 = (mu - q1 * mu1) / q2
        s = q1 * q2 * (mu1 - mu2) ** 2
        if s > best:
            best, best_t = s, t
    return best_t


def morph_close(binary, k):
    """cv2.morphologyEx(img, MORPH_CLOSE, ones((k, k))): dilate then erode with the anchor at k//2.  The erosion window
    covers offsets -k//2 .. k-k//2-1; OpenCV's dilation reflects kernel and anchor (offsets -(k-k//2-1) .. k//2), which keeps
    an even-sized close from shifting the image -- the same convention as scipy's grey_dilation / grey_erosion pair.  Borders
    never win (0 for the dilation, 255 for the erosion)."""
    d = ndimage.grey_dilation(binary, size=(k, k), mode='constant', cval=0)
    return ndimage.grey_erosion(d, size=(k, k), mode='constant', cval=255)


def contour_area(contour):
    """cv2.contourArea: |shoelace| over the vertex list."""
    c = np.asarray(contour, np.float64).reshape(-1, 2)
    if len(c) < 3:
        return 0.0
    x, y = c[:, 0], c[:, 1]
    return float(abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1))) * 0.5)


def bounding_rect(contour):
    c = np.asarray(contour).reshape(-1, 2)
    x0, y0 = c.min(0)
    x1, y1 = c.max(0)
    return int(x0), int(y0), int(x1 - x0 + 1), int(y1 - y0 + 1)


def points_polygon_test(contour, pts, chunk=512):
    """cv2.pointPolygonTest(contour, pt, False) for many points: +1 inside, 0 on an edge or vertex, -1 outside."""
    c = np.asarray(contour, np.float64).reshape(-1, 2)
    pts = np.asarray(pts, np.float64).reshape(-1, 2)
    out = np.full(len(pts), -1, np.int64)
    if len(c) == 0:
        return out
    x1, y1 = c[:, 0][None, :], c[:, 1][None, :]
    x2, y2 = np.roll(x1, -1, 1), np.roll(y1, -1, 1)
    for i in range(0, len(pts), chunk):
        px, py = pts[i:i + chunk, 0][:, None], pts[i:i + chunk, 1][:, None]
        cross = (x2 - x1) * (py - y1) - (y2 - y1) * (px - x1)
        on = (cross == 0) & (np.minimum(x1, x2) <= px) & (px <= np.maximum(x1, x2)) & (np.minimum(y1, y2) <= py) & (py <= np.maximum(y1, y2))
        straddle = (y1 <= py) != (y2 <= py)
        with np.errstate(divide='ignore', invalid='ignore'):
            xi = x1 + (py - y1) * (x2 - x1) / (y2 - y1)
        inside = (np.count_nonzero(straddle & (px < xi), axis=1) & 1).astype(bool)
        out[i:i + chunk] = np.where(on.any(1), 0, np.where(inside, 1, -1))
    return out


def point_polygon_test(contour, pt):
    return int(points_polygon_test(contour, [pt])[0])


def find_contours_ccomp(binary):
    """cv2.findContours(img, RETR_CCOMP, CHAIN_APPROX_NONE) reduced to what `segmentTissue` uses: for every 8-connected
    foreground component its outer border (all border pixels, (n,2) x,y) and the borders of the holes inside it (4-connected
    background components that do not reach the image edge; a hole's border runs over the foreground pixels around it)."""
    fg = np.asarray(binary) > 0
    lab, n = ndimage.label(fg, structure=np.ones((3, 3), int))
    bg_lab, nb = ndimage.label(~fg)                                   # 4-connectivity
    edge = np.unique(np.concatenate([bg_lab[0], bg_lab[-1], bg_lab[:, 0], bg_lab[:, -1]]))
    hole_ids = [i for i in range(1, nb + 1) if i not in set(edge.tolist())]
    holes_of = {i: [] for i in range(1, n + 1)}
    slices = ndimage.find_objects(bg_lab)
    for h in hole_ids:
        sl = slices[h - 1]
        ys, xs = sl[0], sl[1]
        y0, y1 = max(ys.start - 1, 0), min(ys.stop + 1, fg.shape[0])
        x0, x1 = max(xs.start - 1, 0), min(xs.stop + 1, fg.shape[1])
        hm = bg_lab[y0:y1, x0:x1] == h
        yy, xx = np.nonzero(hm)
        owner = 0
        for dy, dx in ((0, -1), (-1, 0), (0, 1), (1, 0)):              # a 4-neighbour of the hole is foreground of its parent
            py, px = yy[0] + dy + y0, xx[0] + dx + x0
            if 0 <= py < fg.shape[0] and 0 <= px < fg.shape[1] and lab[py, px]:
                owner = lab[py, px]
                break
        if owner == 0:
            continue
        # the hole border runs over the foreground pixels around the hole, followed from the hole's side (Suzuki-Abe: start at the
        # foreground pixel left of the hole's first raster pixel, whose east neighbour is the hole)
        c = _follow_hole_border(fg, int(yy[0]) + y0, int(xx[0]) + x0 - 1)
        holes_of[owner].append(c)
    out = []
    obj = ndimage.find_objects(lab)
    for i in range(1, n + 1):
        sl = obj[i - 1]
        comp = lab[sl] == i
        c = _trace_all(comp) + np.array([sl[1].start, sl[0].start])
        out.append((c, holes_of[i][::-1]))
    # cv2's list order: every new border is linked at the head of its parent's list, so the outer borders come in the REVERSE of
    # the order the raster scan finds them (scipy labels components and background regions in scan order of their first pixel),
    # and so do the holes of one component
    return out[::-1]


_DXY = ((1, 0), (1, -1), (0, -1), (-1, -1), (-1, 0), (-1, 1), (0, 1), (1, 1))     # chain codes counter-clockwise from east (dx, dy), y down


def _follow_hole_border(fg, y0, x0):
    """Suzuki-Abe border following (CHAIN_APPROX_NONE) of the hole border that starts at foreground pixel (y0, x0) whose east
    neighbour belongs to the hole: every foreground pixel 4-adjacent to the hole, in following order, as (n, 2) (x, y).  Unlike
    the outer trace of the dilated hole, it keeps the foreground pixel inside a concave corner of the hole."""
    H, W = fg.shape

    def on(y, x):
        return 0 <= y < H and 0 <= x < W and fg[y, x]
    s = 0                                      # the east neighbour is the hole: search clockwise from it
    k = 0
    while True:
        s = (s - 1) & 7
        k += 1
        if on(y0 + _DXY[s][1], x0 + _DXY[s][0]):
            break
        if k == 8:
            return np.array([[x0, y0]], np.int64)
    y1, x1 = y0 + _DXY[s][1], x0 + _DXY[s][0]
    pts = []
    cy, cx = y0, x0
    while True:
        while True:                            # counter-clockwise from the code after the one pointing back
            s = (s + 1) & 7
            ny, nx = cy + _DXY[s][1], cx + _DXY[s][0]
            if on(ny, nx):
                break
        pts.append((cx, cy))
        if (ny, nx) == (y0, x0) and (cy, cx) == (y1, x1):
            break
        cy, cx = ny, nx
        s = (s + 4) & 7
    return np.array(pts, np.int64)


def _trace_all(mask):
    """Outer border with every border pixel (CHAIN_APPROX_NONE): the Moore trace of contours.py without its compression."""
    m = np.asarray(mask, bool)
    # trace_outer_contour compresses collinear runs; re-expand the runs between consecutive vertices (unit 8-steps)
    v = trace_outer_contour(m)
    if len(v) <= 1:
        return v
    pts = []
    for a, b in zip(v, np.roll(v, -1, 0)):
        d = b - a
        steps = int(max(abs(d[0]), abs(d[1])))
        if steps == 0:
            continue
        s = d // steps
        for t in range(steps):
            pts.append(a + s * t)
    return np.array(pts, np.int64)


# ----------------------------------------------------------------------------- segmentTissue
def segment_tissue(image, scale=64, sthresh=8, sthresh_up=255, mthresh=7, close=4, use_otsu=False, filter_params=None,
                   ref_patch_size=512, exclude_ids=(), keep_ids=(), level_image=None):
    """`WholeSlideImage.segmentTissue` on an RGB array.  `scale`: downsample factor of the segmentation level (the reference
    picks the pyramid level closest to 64x).  Returns (contours_tissue, holes_tissue) in level-0 pixels: lists of (n,2) int32
    contours and, per contour, a list of hole contours.
    `level_image`: the segmentation level of a pyramid slide as the reference reads it (`read_region((0, 0), seg_level, level_dim)`, :159);
    `scale` is then that level's (x, y) downsample pair (`_assertLevelDownsamples`, :378-386): the area thresholds use scale[0] (:176) and
    the contours go back to level 0 by `cont * scale` truncated to int32 (`scaleContourDim`, :371)."""
    fp = dict(FILTER_PARAMS if filter_params is None else filter_params)
    if level_image is not None:
        scale = np.asarray(scale, np.float64).reshape(-1)
        scale = np.array([scale[0], scale[-1]])
        img = np.asarray(level_image)[:, :, :3]
        ref_scale = float(scale[0])
    else:
        scale = int(scale)
        img = np.asarray(image[::scale, ::scale])          # (sliced first: a TIFF slide serves this from its pyramid, nuhtc_amd.tiffslide)
        ref_scale = scale
    sat = saturation_u8(img)
    med = median_blur(sat, mthresh)
    thr = otsu_threshold(med) if use_otsu else sthresh
    binary = np.where(med > thr, np.uint8(min(sthresh_up, 255)), np.uint8(0))
    if close > 0:
        binary = morph_close(binary, close)
    scaled_ref = round(ref_patch_size / ref_scale)
    a_t, a_h = fp['a_t'] * scaled_ref, fp['a_h'] * scaled_ref
    fore, holes = [], []
    for cont, hs in find_contours_ccomp(binary):
        a = contour_area(cont) - sum(contour_area(h) for h in hs)
        if a == 0 or not a > a_t:
            continue
        hs = sorted(hs, key=contour_area, reverse=True)[:fp['max_n_holes']]
        fore.append(cont)
        holes.append([h for h in hs if contour_area(h) > a_h])
    fore = [np.array(c * scale, dtype='int32') for c in fore]
    holes = [[np.array(h * scale, dtype='int32') for h in hs] for hs in holes]
    ids = (set(keep_ids) if len(keep_ids) else set(range(len(fore)))) - set(exclude_ids)
    ids = sorted(ids)
    return [fore[i] for i in ids], [holes[i] for i in ids]


# ----------------------------------------------------------------------------- tile coordinates
def _check_offsets(patch_size, fn, center_shift=0.5):
    """Offsets from a tile origin of the points util_classes.py:53-111 test: 'basic' (corner), 'center', 'four_pt*'."""
    if fn == 'basic':
        return [(0, 0)]
    h = patch_size // 2
    if fn == 'center':
        return [(h, h)]
    s = int(patch_size // 2 * center_shift)
    return [(h - s, h - s), (h + s, h + s), (h + s, h - s), (h - s, h + s)] if s > 0 else [(h, h)]


def in_contour(contour, pts, patch_size, contour_fn='four_pt'):
    """For tile origins `pts` (n,2): 'four_pt' passes when any of the 4 points is inside or on the contour, 'four_pt_hard'
    when all are; 'basic' / 'center' test one point."""
    pts = np.asarray(pts).reshape(-1, 2)
    tests = np.stack([points_polygon_test(contour, pts + np.array(o)) >= 0 for o in _check_offsets(patch_size, contour_fn)], 1)
    return tests.all(1) if contour_fn == 'four_pt_hard' else tests.any(1)


def in_holes(holes, pts, patch_size):
    """WholeSlideImage.isInHoles (:353-358): patch centre strictly inside a hole."""
    pts = np.asarray(pts, np.float64).reshape(-1, 2)
    hit = np.zeros(len(pts), bool)
    for h in holes:
        hit |= points_polygon_test(h, pts + patch_size / 2) > 0
    return hit


def contour_coords(contour, holes, image_wh, patch_size=256, step_size=256, contour_fn='four_pt', use_padding=True,
                   top_left=None, bot_right=None):
    """`process_contour` (:407-493) at patch_level 0: (n,2) int64 level-0 (x, y) tile origins of one tissue contour."""
    if contour_fn not in ('basic', 'center', 'four_pt', 'four_pt_hard'):
        raise NotImplementedError(contour_fn)
    img_w, img_h = image_wh
    start_x, start_y, w, h = bounding_rect(contour) if contour is not None else (0, 0, img_w, img_h)
    if use_padding:
        stop_y, stop_x = start_y + h, start_x + w
    else:
        stop_y, stop_x = min(start_y + h, img_h - patch_size + 1), min(start_x + w, img_w - patch_size + 1)
    if bot_right is not None:
        stop_y, stop_x = min(bot_right[1], stop_y), min(bot_right[0], stop_x)
    if top_left is not None:
        start_y, start_x = max(top_left[1], start_y), max(top_left[0], start_x)
    if (bot_right is not None or top_left is not None) and (stop_x - start_x <= 0 or stop_y - start_y <= 0):
        return np.zeros((0, 2), np.int64)
    xs = np.arange(start_x, stop_x, step=step_size)
    ys = np.arange(start_y, stop_y, step=step_size)
    gx, gy = np.meshgrid(xs, ys, indexing='ij')
    cand = np.stack([gx.reshape(-1), gy.reshape(-1)], 1)
    ok = np.ones(len(cand), bool) if contour is None else in_contour(contour, cand, patch_size, contour_fn)
    ok &= ~in_holes(holes or [], cand, patch_size)
    keep = cand[ok].astype(np.int64)
    # the reference stores a contour's coordinates only when it yields more than one (`if len(results) > 1`, :476)
    return keep if len(keep) > 1 else np.zeros((0, 2), np.int64)


def tissue_tile_coords(image, patch_size=256, step_size=256, scale=64, seg_params=None, filter_params=None,
                       contour_fn='four_pt', use_padding=True):
    """segmentTissue + process_contours: tile origins over all tissue contours, in contour order (the .h5 `coords`)."""
    sp = dict(SEG_PARAMS if seg_params is None else seg_params)
    conts, holes = segment_tissue(image, scale=scale, filter_params=filter_params, **sp)
    H, W = image.shape[:2]
    parts = [contour_coords(c, h, (W, H), patch_size, step_size, contour_fn, use_padding) for c, h in zip(conts, holes)]
    coords = np.concatenate(parts, 0) if parts else np.zeros((0, 2), np.int64)
    return coords, conts, holes


def read_tiles(image, coords, patch_size=256):
    """`Whole_Slide_Bag_FP.__getitem__` (:886-898) on an array slide: RGB tiles at `coords`, zero padded past the edge."""
    H, W = image.shape[:2]
    out = np.zeros((len(coords), patch_size, patch_size, 3), np.uint8)
    for i, (x, y) in enumerate(coords):
        sub = image[max(y, 0):min(y + patch_size, H), max(x, 0):min(x + patch_size, W), :3]
        out[i, :sub.shape[0], :sub.shape[1]] = sub
    return out
