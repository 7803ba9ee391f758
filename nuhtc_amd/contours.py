"""Contours, polygons and GeoJSON for the WSI path (host side, numpy only).

Reference: tools/infer_wsi.py:51-58 `mask2inst` = `cv2.findContours(mask, RETR_TREE, CHAIN_APPROX_SIMPLE)[0][0]`, ring
closed by repeating the first point, shifted by the tile origin (:533-539); QuPath features (:548-585);
tools/nuclei_merge.py:62-174 polygon-IoU merge.  cv2 and shapely are not available: the border follower below follows
OpenCV's conventions (which component comes first in the list, start pixel, orientation, CHAIN_APPROX_SIMPLE vertex rule)
and is checked vertex for vertex against oracle/contour.py, the restatement of the published Suzuki-Abe algorithm."""
import json

import numpy as np

# Freeman chain codes as OpenCV numbers them: 0 = east, counter-clockwise on the screen (x right, y down)
_DX = (1, 1, 0, -1, -1, -1, 0, 1)
_DY = (0, -1, -1, -1, 0, 1, 1, 1)


def contour_start(mask):
    """Start pixel (y, x) of the contour `cv2.findContours(mask, RETR_TREE, CHAIN_APPROX_SIMPLE)` lists first, or None for an
    empty mask.  OpenCV links every border it finds at the head of its parent's child list, so the first listed contour is
    the outer border of the top-level component found last in the raster scan: among the 8-connected components that are not
    enclosed in a hole of another one, the one whose first pixel in raster order comes last.  A component is top-level iff
    the background pixel west of its first pixel belongs to the background region connected to the image frame (4-connected,
    the image taken with a frame of zeros as cv::findContours pads it)."""
    from scipy import ndimage as ndi
    m = np.asarray(mask, bool)
    if not m.any():
        return None
    lab, n = ndi.label(m, structure=np.ones((3, 3), int))
    flat = lab.reshape(-1)
    first = np.full(n + 1, flat.size, np.int64)
    idx = np.nonzero(flat)[0]
    np.minimum.at(first, flat[idx], idx)
    if n == 1:
        return divmod(int(first[1]), m.shape[1])
    pad = np.zeros((m.shape[0] + 2, m.shape[1] + 2), bool)
    pad[1:-1, 1:-1] = m
    bg, _ = ndi.label(~pad)                       # 4-connected background regions; the frame's region is bg[0, 0]
    outer = bg == bg[0, 0]
    best = None
    for k in range(1, n + 1):
        y, x = divmod(int(first[k]), m.shape[1])
        if outer[y + 1, x] and (best is None or first[k] > best[0]):      # padded coords: west neighbour = (y+1, x+1-1)
            best = (first[k], y, x)
    return best[1], best[2]


def trace_outer_contour(mask):
    """`cv2.findContours(mask, RETR_TREE, CHAIN_APPROX_SIMPLE)[0][0]` (tools/infer_wsi.py:51-54) as an (n,2) int array of
    (x, y): the outer border (Suzuki-Abe border following, 8-connectivity, counter-clockwise on the screen: down the left side
    first) of the component `contour_start` selects, starting at its first pixel in raster order, a vertex wherever the chain
    code changes.  Empty array for an empty mask.  The device kernel (csrc/contour.hip) runs the same steps; both are checked
    against the full border-following restatement in oracle/contour.py."""
    m = np.asarray(mask, bool)
    st = contour_start(m)
    if st is None:
        return np.zeros((0, 2), np.int64)
    H, W = m.shape
    y0, x0 = st

    def fg(y, x):
        return 0 <= y < H and 0 <= x < W and m[y, x]
    s = 4                                     # the west neighbour of the start is background: search clockwise from it
    while True:
        s = (s - 1) & 7
        if fg(y0 + _DY[s], x0 + _DX[s]) or s == 4:
            break
    if s == 4:
        return np.array([[x0, y0]], np.int64)  # isolated pixel
    y1, x1 = y0 + _DY[s], x0 + _DX[s]
    pts = []
    cy, cx, prev_s = y0, x0, s ^ 4
    while True:
        while True:                           # counter-clockwise from the code after the one pointing back
            s += 1
            ny, nx = cy + _DY[s & 7], cx + _DX[s & 7]
            if fg(ny, nx):
                break
        s &= 7
        if s != prev_s:
            pts.append((cx, cy))
            prev_s = s
        if (ny, nx) == (y0, x0) and (cy, cx) == (y1, x1):
            break
        cy, cx = ny, nx
        s = (s + 4) & 7
    return np.array(pts, np.int64)


def mask_to_ring(mask, origin=(0, 0)):
    """`mask2inst` + `contour_map`: closed ring (first point repeated) in slide coordinates, shape (n+1, 2)."""
    c = trace_outer_contour(mask)
    if len(c) == 0:
        return c
    c = np.concatenate([c, c[:1]], 0)
    return c + np.asarray(origin, np.int64)[None, :]


def rasterize_polygon(ring, x0, y0, w, h):
    """Even-odd fill at pixel centres (x+0.5, y+0.5) of a w x h window with origin (x0, y0); ring: (n,2) closed or open."""
    r = np.asarray(ring, np.float64)
    if len(r) and np.array_equal(r[0], r[-1]):
        r = r[:-1]
    out = np.zeros((h, w), bool)
    if len(r) < 3:
        return out
    py = y0 + np.arange(h) + 0.5
    px = x0 + np.arange(w) + 0.5
    x1, y1 = r[:, 0], r[:, 1]
    x2, y2 = np.roll(x1, -1), np.roll(y1, -1)
    for a in range(len(r)):
        if y1[a] == y2[a]:
            continue
        ylo, yhi = min(y1[a], y2[a]), max(y1[a], y2[a])
        rows = (py >= ylo) & (py < yhi)
        if not rows.any():
            continue
        xint = x1[a] + (py[rows] - y1[a]) * (x2[a] - x1[a]) / (y2[a] - y1[a])
        out[rows] ^= px[None, :] > xint[:, None]
    return out


class RingRegion:
    """Exact region of a closed integer ring whose edges run along the 8 chain-code directions (every ring infer_wsi.py
    writes), as shapely sees it in tools/nuclei_merge.py:37-59: the points of non-zero winding number (`buffer(0)` of a ring
    that touches itself), of several parts the largest.  Such a region is a union of the quarter triangles that the two
    diagonals cut out of the unit cells between integer points, so it is held as four boolean planes N, E, S, W over the
    cells of the ring's bounding box; areas are multiples of 1/4 and exact."""

    def __init__(self, ring):
        r = np.asarray(ring, np.int64)
        if len(r) and np.array_equal(r[0], r[-1]):
            r = r[:-1]
        self.x0 = self.y0 = 0
        self.q = np.zeros((4, 0, 0), bool)
        if len(r) < 3:
            return
        d = np.roll(r, -1, 0) - r
        if not ((d[:, 0] == 0) | (d[:, 1] == 0) | (np.abs(d[:, 0]) == np.abs(d[:, 1]))).all():
            raise ValueError('ring has an edge that is not horizontal, vertical or diagonal')
        self.x0, self.y0 = int(r[:, 0].min()), int(r[:, 1].min())
        w, h = int(r[:, 0].max()) - self.x0, int(r[:, 1].max()) - self.y0
        if w == 0 or h == 0:
            return
        xa, ya = (r[:, 0] - self.x0).astype(np.float64), (r[:, 1] - self.y0).astype(np.float64)
        xb, yb = np.roll(xa, -1), np.roll(ya, -1)
        cy = np.arange(h, dtype=np.float64)[:, None]
        cx = np.arange(w, dtype=np.float64)[None, :]
        q = np.zeros((4, h, w), bool)
        # winding number at the centroid-side sample point of each quarter (never on an edge or a vertex):
        # N (x+.5, y+.25), E (x+.75, y+.5), S (x+.5, y+.75), W (x+.25, y+.5)
        for k, (ox, oy) in enumerate(((0.5, 0.25), (0.75, 0.5), (0.5, 0.75), (0.25, 0.5))):
            wn = np.zeros((h, w), np.int64)
            py = cy + oy
            for e in range(len(r)):
                if ya[e] == yb[e]:
                    continue
                lo, hi = min(ya[e], yb[e]), max(ya[e], yb[e])
                rows = (py[:, 0] > lo) & (py[:, 0] < hi)
                if not rows.any():
                    continue
                xc = xa[e] + (py[rows] - ya[e]) * (xb[e] - xa[e]) / (yb[e] - ya[e])
                wn[rows] += np.where(cx + ox > xc, 1 if yb[e] > ya[e] else -1, 0)
            q[k] = wn != 0
        self.q = self._largest_part(q)

    @staticmethod
    def _largest_part(q):
        """Parts = sets of quarters connected through shared edges (not through points): label on a 3x3 sub-grid per cell."""
        from scipy import ndimage as ndi
        N, E, S, W = q
        h, w = N.shape
        g = np.zeros((3 * h, 3 * w), bool)
        g[0::3, 1::3], g[1::3, 2::3], g[2::3, 1::3], g[1::3, 0::3] = N, E, S, W
        g[0::3, 2::3], g[2::3, 2::3], g[2::3, 0::3], g[0::3, 0::3] = N & E, E & S, S & W, W & N    # neighbours inside a cell
        lab, k = ndi.label(g)
        if k <= 1:
            return q
        planes = (lab[0::3, 1::3], lab[1::3, 2::3], lab[2::3, 1::3], lab[1::3, 0::3])
        area = np.zeros(k + 1, np.int64)
        for pl, l in zip(q, planes):
            np.add.at(area, l[pl], 1)
        best = int(np.argmax(area[1:])) + 1
        return np.stack([pl & (l == best) for pl, l in zip(q, planes)])

    @property
    def area4(self):
        return int(self.q.sum())

    def inter4(self, other):
        x0, y0 = max(self.x0, other.x0), max(self.y0, other.y0)
        x1 = min(self.x0 + self.q.shape[2], other.x0 + other.q.shape[2])
        y1 = min(self.y0 + self.q.shape[1], other.y0 + other.q.shape[1])
        if x1 <= x0 or y1 <= y0:
            return 0
        a = self.q[:, y0 - self.y0:y1 - self.y0, x0 - self.x0:x1 - self.x0]
        b = other.q[:, y0 - other.y0:y1 - other.y0, x0 - other.x0:x1 - other.x0]
        return int((a & b).sum())


def polygon_iou(ring_a, ring_b, supersample=1):
    """IoU of two ring polygons as tools/nuclei_merge.py:128-129 computes it.  Exact (RingRegion) for traced rings; rings with
    other edge directions or non-integer vertices (GeoJSON from elsewhere) fall back to an even-odd raster at
    `supersample` samples per pixel."""
    a = np.asarray(ring_a, np.float64)
    b = np.asarray(ring_b, np.float64)
    try:
        if not (np.array_equal(a, np.round(a)) and np.array_equal(b, np.round(b))):
            raise ValueError('non-integer vertices')
        ra = ring_a if isinstance(ring_a, RingRegion) else RingRegion(a.astype(np.int64))
        rb = ring_b if isinstance(ring_b, RingRegion) else RingRegion(b.astype(np.int64))
        inter = ra.inter4(rb)
        union = ra.area4 + rb.area4 - inter
        return inter / union if union else 0.0
    except ValueError:
        pass
    x0 = int(np.floor(min(a[:, 0].min(), b[:, 0].min()))) - 1
    y0 = int(np.floor(min(a[:, 1].min(), b[:, 1].min()))) - 1
    x1 = int(np.ceil(max(a[:, 0].max(), b[:, 0].max()))) + 1
    y1 = int(np.ceil(max(a[:, 1].max(), b[:, 1].max()))) + 1
    s = supersample
    ra = rasterize_polygon((a - [x0, y0]) * s, 0, 0, (x1 - x0) * s, (y1 - y0) * s)
    rb = rasterize_polygon((b - [x0, y0]) * s, 0, 0, (x1 - x0) * s, (y1 - y0) * s)
    inter = np.logical_and(ra, rb).sum()
    union = ra.sum() + rb.sum() - inter
    return inter / union if union else 0.0


def feature(ring, label, score, class_names, colors=None):
    """QuPath 0.4.4 annotation feature as written by tools/infer_wsi.py:550-566."""
    colors = colors or [[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 0], [255, 0, 255], [0, 255, 255]]
    return {"type": "Feature", "geometry": {"type": "Polygon", "coordinates": [np.asarray(ring).tolist()]},
            "properties": {"objectType": "annotation", "label": int(label), "score": float(score),
                           "classification": {"name": class_names[int(label)], "color": colors[int(label) % len(colors)]}, "isLocked": False}}


def point_feature(box, label, score, class_names, colors=None):
    colors = colors or [[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 0], [255, 0, 255], [0, 255, 255]]
    return {"type": "Feature", "geometry": {"type": "Point", "coordinates": [float(box[0] + box[2]) / 2, float(box[1] + box[3]) / 2]},
            "properties": {"objectType": "annotation", "label": int(label), "score": float(score),
                           "classification": {"name": class_names[int(label)], "color": colors[int(label) % len(colors)]}, "isLocked": False}}


def _template_pieces(kind, label, class_names, colors=None):
    """(head, mid, tail): the text json.dumps writes around the coordinates and around the score of a feature of class `label` --
    cut out of the dump of one template feature, so key order, separators and the classification block are json's own."""
    if kind == 'polygon':
        s = json.dumps(feature(np.array([[987654321, 987654322]]), label, 0.123456789, class_names, colors))
        head, rest = s.split('[987654321, 987654322]')
    else:
        s = json.dumps(point_feature(np.array([1.25, 2.25, 1.25, 2.25]), label, 0.123456789, class_names, colors))
        head, rest = s.split('1.25, 2.25')
    mid, tail = rest.split('0.123456789')
    return head, mid, tail


def _pieces_c(kind, class_names, colors):
    import ctypes
    nl = len(class_names)
    pieces = [_template_pieces(kind, l, class_names, colors) for l in range(nl)]
    head = pieces[0][0].encode('ascii')
    assert all(p[0].encode('ascii') == head for p in pieces)
    return head, (ctypes.c_char_p * nl)(*[p[1].encode('ascii') for p in pieces]), (ctypes.c_char_p * nl)(*[p[2].encode('ascii') for p in pieces]), nl


def _sized_call(fn, args, tail=()):
    """The library's text writers: one call to size, one to write -> uint8 array."""
    import ctypes
    need = fn(*args, None, 0, *tail)
    if need < 0:
        raise RuntimeError(f'{fn.__name__} failed ({need})')
    out = np.empty(max(need, 1), np.uint8)
    got = fn(*args, ctypes.c_void_p(out.ctypes.data), need, *tail)
    if got < 0:
        raise RuntimeError(f'{fn.__name__} failed ({got})')
    return out[:got]


def ring_features_text(verts, ring_n, labels, scores, class_names, colors=None, threads=0):
    """The text of `json.dumps([feature(ring_i, label_i, score_i, class_names) for i ...])` without its enclosing brackets, written from
    arrays by the library (nuhtc_write_ring_features: memory speed on a few host threads instead of ~40 us of dict building and dumping per
    nucleus, tools/infer_wsi.py:550-566,659-664 of the reference): `verts` int32 (sum ring_n, 2) closed rings back to back, `ring_n` their
    lengths.  Returns (text uint8 array, start int64 (n + 1)): feature i is text[start[i] : start[i + 1] - 2]."""
    import ctypes
    from . import hip
    n = len(ring_n)
    if n == 0:
        return np.zeros(0, np.uint8), np.zeros(1, np.int64)
    lib = hip.load()
    verts = np.ascontiguousarray(verts, np.int32).reshape(-1, 2)
    off = np.concatenate([[0], np.cumsum(np.asarray(ring_n, np.int64))]).astype(np.int64)
    labels = np.ascontiguousarray(labels, np.int32)
    scores = np.ascontiguousarray(scores, np.float64)
    head, mid, tail, nl = _pieces_c('polygon', class_names, colors)
    vp = lambda a: ctypes.c_void_p(a.ctypes.data)
    start = np.zeros(n + 1, np.int64)
    text = _sized_call(lib.nuhtc_write_ring_features, (vp(verts), vp(off), vp(labels), vp(scores), n, head, mid, tail, nl), (vp(start), int(threads)))
    return text, start


def point_features_text(boxes, labels, scores, class_names, colors=None):
    """The text of `json.dumps([point_feature(box_i, label_i, score_i, class_names) ...])` without its brackets (uint8 array)."""
    import ctypes
    from . import hip
    n = len(labels)
    if n == 0:
        return np.zeros(0, np.uint8)
    b = np.asarray(boxes, np.float64)
    xy = np.ascontiguousarray(np.stack([(b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2], 1))       # point_feature's float(b0 + b2) / 2
    labels = np.ascontiguousarray(labels, np.int32)
    scores = np.ascontiguousarray(scores, np.float64)
    head, mid, tail, nl = _pieces_c('point', class_names, colors)
    vp = lambda a: ctypes.c_void_p(a.ctypes.data)
    return _sized_call(hip.load().nuhtc_write_point_features, (vp(xy), vp(labels), vp(scores), n, head, mid, tail, nl))


def join_features_text(text, start, pick, threads=0):
    """The records `pick` of a ring_features_text result, joined by ", " (uint8 array)."""
    import ctypes
    from . import hip
    pick = np.ascontiguousarray(pick, np.int64)
    if len(pick) == 0:
        return np.zeros(0, np.uint8)
    text = np.ascontiguousarray(text, np.uint8)
    start = np.ascontiguousarray(start, np.int64)
    if pick.min() < 0 or pick.max() >= len(start) - 1:
        raise IndexError('join_features_text: record index out of range')
    vp = lambda a: ctypes.c_void_p(a.ctypes.data)
    return _sized_call(hip.load().nuhtc_join_features, (vp(text), vp(start), vp(pick), len(pick)), (int(threads),))


def concat_feature_texts(texts, starts=None):
    """Per-rank texts of ring_features_text / point_features_text (rank order) -> the body of the whole list (uint8 array, ", " between
    the ranks that have records) and, given the ranks' `start` arrays, the start array of the concatenation."""
    live = [np.ascontiguousarray(t, np.uint8) for t in texts if len(t)]
    total = sum(len(t) for t in live) + 2 * max(len(live) - 1, 0)
    body = np.empty(total, np.uint8)
    pos, bases = 0, []
    for t in texts:
        bases.append(pos)
        if len(t):
            body[pos:pos + len(t)] = t
            pos += len(t)
            if pos < total:
                body[pos], body[pos + 1] = 44, 32           # ", "
                pos += 2
    if starts is None:
        return body, None
    st = [np.asarray(s, np.int64)[:-1] + b for s, b, t in zip(starts, bases, texts) if len(t)]
    return body, np.concatenate(st + [np.array([total + 2], np.int64)])


def fill_rings(rings, threads=0):
    """Closed traced rings (integer vertices on pixel centres, edges along the 8 chain directions; the repeated closing vertex is dropped
    here) -> the mask crops nuhtc_merge_overlap takes: (boxes int32 (n,4) x1 / y1 exclusive, areas int32 (n,), bits uint32, off int64 (n,)).
    The pixels inside or on such a ring are the traced component with its holes filled (nuhtc_fill_rings).  None when a ring is not a
    traced ring (an edge off the chain directions, non-integer coordinates, fewer than one vertex)."""
    import ctypes
    from . import hip
    n = len(rings)
    if n == 0:
        return np.zeros((0, 4), np.int32), np.zeros(0, np.int32), np.zeros(1, np.uint32), np.zeros(0, np.int64)
    parts, lens = [], np.zeros(n, np.int64)
    for i, r in enumerate(rings):
        r = np.asarray(r)
        if r.ndim != 2 or r.shape[1] != 2 or len(r) == 0:
            return None
        if r.dtype.kind == 'f':
            ri = np.rint(r)
            if not np.array_equal(ri, r):
                return None
            r = ri
        if len(r) > 1 and (r[0] == r[-1]).all():
            r = r[:-1]
        parts.append(r.astype(np.int32))
        lens[i] = len(r)
    verts = np.ascontiguousarray(np.concatenate(parts, 0))
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    boxes, areas, woff = np.zeros((n, 4), np.int32), np.zeros(n, np.int32), np.zeros(n, np.int64)
    vp = lambda a: ctypes.c_void_p(a.ctypes.data)
    lib = hip.load()
    need = lib.nuhtc_fill_rings(vp(verts), vp(off), n, vp(boxes), vp(areas), vp(woff), None, 0, int(threads))
    if need < 0:
        return None
    bits = np.zeros(max(int(need), 1), np.uint32)
    if lib.nuhtc_fill_rings(vp(verts), vp(off), n, vp(boxes), vp(areas), vp(woff), vp(bits), int(need), int(threads)) < 0:
        return None
    return boxes, areas, bits, woff


def merge_features_device(features, overlap_threshold=0.01, device=0):
    """merge_features(..., 'probability') on the GPU for a file of TRACED rings (what tools/infer_wsi.py writes): the rings are filled back
    into mask crops (fill_rings) and nuhtc_merge_overlap measures the reference's polygon IoU on them exactly (csrc/merge.hip) -- the same
    kept features and nuclei_ids as the host path, ~1 ms per feature faster.  None when the file holds anything else (the caller then
    takes the host path)."""
    from . import wsi
    n = len(features)
    order = sorted(range(n), key=lambda i: -features[i]['properties'].get('score', 0))
    try:
        rings = [np.asarray(features[i]['geometry']['coordinates'][0]) for i in order]
    except (KeyError, IndexError, TypeError, ValueError):
        return None
    if any(r.ndim != 2 or len(r) < 2 for r in rings):
        return None
    crops = fill_rings(rings)
    if crops is None:
        return None
    scores = np.array([features[i]['properties'].get('score', 0) for i in order], np.float64)
    kept = wsi.merge_overlap_packed(crops[0], scores, crops[1], crops[2], crops[3], overlap_threshold, device, 'polygon')
    out = []
    for k in kept.tolist():
        src = features[order[k]]
        f = dict(src)
        f['properties'] = dict(src['properties'], nuclei_id=k)
        out.append(f)
    return out


def merge_features(features, overlap_threshold=0.01, merge_strategy='probability'):
    """tools/nuclei_merge.py:62-174 on a list of GeoJSON features: sort by properties.score (descending, stable), greedy
    suppression of every not-yet-visited feature whose polygon IoU with the query exceeds the threshold; 'probability'
    keeps the query, 'area' keeps the largest suppressed polygon instead.  Returns the kept features in score order with
    `properties.nuclei_id` set (the reference re-indexes after the sort, so ids follow score order)."""
    if merge_strategy not in ('probability', 'area'):
        raise ValueError(f"Invalid merge strategy: {merge_strategy}. Use 'probability' or 'area'.")
    n = len(features)
    order = sorted(range(n), key=lambda i: -features[i]['properties'].get('score', 0))
    rings = [np.asarray(features[i]['geometry']['coordinates'][0], np.float64) for i in order]
    boxes = np.array([[r[:, 0].min(), r[:, 1].min(), r[:, 0].max(), r[:, 1].max()] for r in rings]) if n else np.zeros((0, 4))
    regs = [None] * n

    def region(i):
        """exact region of a traced ring, or the ring itself when it is not one (polygon_iou then rasterises)"""
        if regs[i] is None:
            try:
                if not np.array_equal(rings[i], np.round(rings[i])):
                    raise ValueError
                regs[i] = RingRegion(rings[i].astype(np.int64))
            except ValueError:
                regs[i] = rings[i]
        return regs[i]
    cell = 64.0
    grid = {}
    for i in range(n):
        for cy in range(int(boxes[i, 1] // cell), int(boxes[i, 3] // cell) + 1):
            for cx in range(int(boxes[i, 0] // cell), int(boxes[i, 2] // cell) + 1):
                grid.setdefault((cx, cy), []).append(i)
    visited = np.zeros(n, bool)
    kept = []
    for q in range(n):
        if visited[q]:
            continue
        cands = set()
        for cy in range(int(boxes[q, 1] // cell), int(boxes[q, 3] // cell) + 1):
            for cx in range(int(boxes[q, 0] // cell), int(boxes[q, 2] // cell) + 1):
                cands.update(grid.get((cx, cy), ()))
        sub = []
        for c in sorted(cands):
            if c == q or visited[c]:
                continue
            if boxes[q, 0] > boxes[c, 2] or boxes[c, 0] > boxes[q, 2] or boxes[q, 1] > boxes[c, 3] or boxes[c, 1] > boxes[q, 3]:
                continue
            rq, rc_ = region(q), region(c)
            if isinstance(rq, RingRegion) and isinstance(rc_, RingRegion):
                inter = rq.inter4(rc_)
                union = rq.area4 + rc_.area4 - inter
                iou = inter / union if union else 0.0
            else:
                iou = polygon_iou(rings[q], rings[c], supersample=4)
            if iou > overlap_threshold:
                sub.append(c)
                visited[c] = True
        if sub and merge_strategy == 'area':
            def area_of(c):
                rg = region(c)
                if isinstance(rg, RingRegion):
                    return rg.area4 / 4.0
                r = rings[c]
                return float(rasterize_polygon(r - [np.floor(boxes[c, 0]), np.floor(boxes[c, 1])], 0, 0,
                                               int(boxes[c, 2] - boxes[c, 0]) + 2, int(boxes[c, 3] - boxes[c, 1]) + 2).sum())
            kept.append(max(sub, key=area_of))
        else:
            kept.append(q)
        visited[q] = True
    out = []
    for k in sorted(set(kept)):
        src = features[order[k]]
        f = dict(src)                                    # the caller's features stay as they are: copy the two levels that change
        f['properties'] = dict(src['properties'], nuclei_id=k)
        out.append(f)
    return out
