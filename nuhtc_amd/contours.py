"""Contours, polygons and GeoJSON for the WSI path (host side, numpy only).

Reference: tools/infer_wsi.py:51-58 `mask2inst` = `cv2.findContours(mask, RETR_TREE, CHAIN_APPROX_SIMPLE)[0][0]`, ring
closed by repeating the first point, shifted by the tile origin (:533-539); QuPath features (:548-585);
tools/nuclei_merge.py:62-174 polygon-IoU merge.  cv2 and shapely are not available: the border follower below traces
the 8-connected outer border of the first component in raster order (what OpenCV returns first for a single blob) with
CHAIN_APPROX_SIMPLE-style compression, and polygon overlap is evaluated by even-odd rasterisation at pixel centres.
Parity with the reference is defined on polygon rasters (IoU), not on vertex lists."""
import json

import numpy as np

# 8-neighbourhood in clockwise order starting east (x right, y down)
_DX = (1, 1, 0, -1, -1, -1, 0, 1)
_DY = (0, 1, 1, 1, 0, -1, -1, -1)


def trace_outer_contour(mask):
    """Outer border (Moore tracing, 8-connectivity) of the component that contains the first foreground pixel in raster
    order.  Returns an (n,2) int array of (x,y) vertices with collinear runs compressed; empty if the mask is empty."""
    m = np.asarray(mask, bool)
    H, W = m.shape
    ys, xs = np.nonzero(m)
    if len(ys) == 0:
        return np.zeros((0, 2), np.int64)
    y0, x0 = int(ys[0]), int(xs[0])     # first pixel in raster order: its west and north neighbours are background

    def fg(y, x):
        return 0 <= y < H and 0 <= x < W and m[y, x]
    pts = [(x0, y0)]
    cy, cx = y0, x0
    d = 4                                # we "arrived" from the west: start the scan just after it
    if not any(fg(y0 + _DY[k], x0 + _DX[k]) for k in range(8)):
        return np.array(pts, np.int64)   # isolated pixel
    start_d = None
    for _ in range(4 * H * W + 8):
        found = False
        for t in range(1, 9):
            k = (d + t) % 8
            ny, nx = cy + _DY[k], cx + _DX[k]
            if fg(ny, nx):
                found = True
                break
        if not found:
            break
        if (cy, cx) == (y0, x0):
            if start_d is None:
                start_d = k
            elif k == start_d:
                break                    # Jacob's stopping criterion: back at the start leaving in the same direction
        cy, cx = ny, nx
        pts.append((cx, cy))
        d = (k + 4) % 8                  # direction pointing back to where we came from
    if len(pts) > 1 and pts[-1] == pts[0]:
        pts.pop()
    # CHAIN_APPROX_SIMPLE: keep only vertices where the step direction changes
    p = np.array(pts, np.int64)
    if len(p) <= 2:
        return p
    step = np.roll(p, -1, 0) - p
    prev = p - np.roll(p, 1, 0)
    keep = np.any(step != prev, axis=1)
    return p[keep] if keep.any() else p[:1]


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


def polygon_iou(ring_a, ring_b, supersample=1):
    a = np.asarray(ring_a, np.float64)
    b = np.asarray(ring_b, np.float64)
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
    areas = [None] * n
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
            if polygon_iou(rings[q], rings[c]) > overlap_threshold:
                sub.append(c)
                visited[c] = True
        if sub and merge_strategy == 'area':
            for c in sub:
                if areas[c] is None:
                    r = rings[c]
                    areas[c] = rasterize_polygon(r - [np.floor(boxes[c, 0]), np.floor(boxes[c, 1])], 0, 0,
                                                 int(boxes[c, 2] - boxes[c, 0]) + 2, int(boxes[c, 3] - boxes[c, 1]) + 2).sum()
            kept.append(max(sub, key=lambda c: areas[c]))
        else:
            kept.append(q)
        visited[q] = True
    out = []
    for k in sorted(set(kept)):
        f = json.loads(json.dumps(features[order[k]]))
        f['properties']['nuclei_id'] = k
        out.append(f)
    return out
