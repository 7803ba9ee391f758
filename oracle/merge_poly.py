"""ORACLE — test infrastructure, NOT product code.

The cross-tile merge of the reference with its own overlap measure: tools/nuclei_merge.py:62-174 `merge_overlap`, strategy
'probability', on the polygons of the GeoJSON rings tools/infer_wsi.py writes (`mask2inst`, :51-54: the first contour
cv2.findContours lists, through the centres of the border pixels).

    process_batch (:37-59)   poly = Polygon(ring); if not poly.is_valid: poly.buffer(0), of a MultiPolygon the largest part
    merge_overlap (:107-133) features sorted by score (descending); a not-yet-visited polygon, in that order, is kept and
                             marks as visited every other not-yet-visited polygon of its STRtree query (bounding boxes
                             intersect) with  inter.area / (a.area + b.area - inter.area) > overlap_threshold

shapely / GEOS are third-party code absent from /root/reference and from this image: PARITY UNPINNED; restated from their
documented behaviour and pinned by the hand-worked cases in tests/test_merge_poly.py.  Areas are exact: ring vertices are
integers and every edge of a traced ring runs along one of the 8 chain-code directions, so two edges can only meet at
coordinates that are multiples of 1/2.  Cutting the plane into horizontal slabs of height 1/2 leaves, inside a slab,
straight non-crossing boundaries: a polygon is a list of trapezoids per slab and every area is an integer number of 1/8
pixel^2.  No raster, no supersampling, no floating point until the final ratio (a double division, as in the reference).

  region of a ring   = points of non-zero winding number.  A traced ring never crosses itself but may touch itself
                       (one-pixel spurs walked out and back, pinches): such a ring is invalid for GEOS, and buffer(0)
                       returns exactly that region as one polygon per connected part (parts touching in a point are
                       separate polygons); the reference keeps the largest part only.  For a valid ring it is the interior.
  ties               = pandas sort_values (quicksort) does not define the order of equal scores; here: lower index first.
"""
import numpy as np


class Poly:
    """Exact region of a closed integer ring with 8-direction edges, as trapezoids per half-slab.
    slabs: dict Y2 -> list of (l0, l1, r0, r1): doubled x of the left / right boundary at the top (Y2) and bottom (Y2+1) of
    the slab, in doubled coordinates (unit 1/2 pixel).  area8 = area in 1/8 pixel^2."""

    def __init__(self, ring, largest_part=True):
        r = np.asarray(ring, np.int64)
        if len(r) and np.array_equal(r[0], r[-1]):
            r = r[:-1]
        self.slabs = {}
        self.area8 = 0
        if len(r) < 3:
            return
        d = np.roll(r, -1, 0) - r
        ok = (d[:, 0] == 0) | (d[:, 1] == 0) | (np.abs(d[:, 0]) == np.abs(d[:, 1]))
        if not ok.all():
            raise ValueError('ring has an edge that is not horizontal, vertical or diagonal: not a traced contour')
        edges = {}
        for (xa, ya), (xb, yb) in zip(2 * r, 2 * np.roll(r, -1, 0)):
            if ya == yb:
                continue
            sgn = 1 if yb > ya else -1
            slope = (xb - xa) // (yb - ya)          # -1, 0 or 1
            for Y in range(min(ya, yb), max(ya, yb)):
                edges.setdefault(int(Y), []).append((int(xa + (Y - ya) * slope), int(xa + (Y + 1 - ya) * slope), sgn))
        for Y, es in edges.items():
            es.sort(key=lambda e: (e[0] + e[1], e[2]))
            traps, w, left = [], 0, None
            for x0, x1, sgn in es:
                if w == 0:
                    left = (x0, x1)
                w += sgn
                if w == 0:
                    if x0 + x1 > left[0] + left[1]:
                        if traps and traps[-1][2] == left[0] and traps[-1][3] == left[1]:     # abutting stretches: one region
                            traps[-1] = (traps[-1][0], traps[-1][1], x0, x1)
                        else:
                            traps.append((left[0], left[1], x0, x1))
            if traps:
                self.slabs[Y] = traps
        if largest_part:
            self._keep_largest_part()
        self.area8 = sum((t[2] - t[0]) + (t[3] - t[1]) for ts in self.slabs.values() for t in ts)

    def _keep_largest_part(self):
        ids, parent = {}, []

        def find(a):
            while parent[a] != a:
                parent[a] = parent[parent[a]]
                a = parent[a]
            return a
        for Y in sorted(self.slabs):
            for k in range(len(self.slabs[Y])):
                ids[(Y, k)] = len(parent)
                parent.append(len(parent))
        for Y in sorted(self.slabs):
            if Y + 1 not in self.slabs:
                continue
            for ka, a in enumerate(self.slabs[Y]):
                for kb, b in enumerate(self.slabs[Y + 1]):
                    if min(a[3], b[2]) - max(a[1], b[0]) > 0:          # shared boundary of positive length
                        ra, rb = find(ids[(Y, ka)]), find(ids[(Y + 1, kb)])
                        if ra != rb:
                            parent[rb] = ra
        area = {}
        for (Y, k), i in ids.items():
            t = self.slabs[Y][k]
            area[find(i)] = area.get(find(i), 0) + (t[2] - t[0]) + (t[3] - t[1])
        if len(area) <= 1:
            return
        best = max(sorted(area), key=lambda r_: area[r_])
        self.slabs = {Y: [t for k, t in enumerate(ts) if find(ids[(Y, k)]) == best] for Y, ts in self.slabs.items()}
        self.slabs = {Y: ts for Y, ts in self.slabs.items() if ts}

    def inter8(self, other):
        tot = 0
        for Y, ta in self.slabs.items():
            tb = other.slabs.get(Y)
            if not tb:
                continue
            for a in ta:
                for b in tb:
                    l0, l1 = max(a[0], b[0]), max(a[1], b[1])
                    r0, r1 = min(a[2], b[2]), min(a[3], b[3])
                    if r0 + r1 > l0 + l1:
                        tot += (r0 - l0) + (r1 - l1)
        return tot


def polygon_iou(ring_a, ring_b):
    a, b = Poly(ring_a), Poly(ring_b)
    inter = a.inter8(b)
    uni = a.area8 + b.area8 - inter
    return inter / uni if uni > 0 else 0.0


def merge_overlap(rings, scores, overlap_threshold=0.01):
    """rings: closed (n+1,2) integer rings in slide coordinates; returns the kept indices (ascending).
    Sequential greedy pass of tools/nuclei_merge.py:107-133 ('probability')."""
    n = len(scores)
    if n == 0:
        return np.zeros(0, np.int64)
    order = np.argsort(-np.asarray(scores, np.float32), kind='stable')
    polys = [Poly(r) for r in rings]
    boxes = np.array([[np.min(r[:, 0]), np.min(r[:, 1]), np.max(r[:, 0]), np.max(r[:, 1])] if len(r) else [0, 0, -1, -1] for r in map(np.asarray, rings)], np.int64)
    cell, grid = 64, {}
    for i in range(n):
        if boxes[i, 2] < boxes[i, 0]:
            continue
        for cy in range(boxes[i, 1] // cell, boxes[i, 3] // cell + 1):
            for cx in range(boxes[i, 0] // cell, boxes[i, 2] // cell + 1):
                grid.setdefault((cx, cy), []).append(i)
    visited = np.zeros(n, bool)
    keep = []
    for q in order:
        if visited[q]:
            continue
        visited[q] = True
        keep.append(int(q))
        if boxes[q, 2] < boxes[q, 0]:
            continue
        cands = set()
        for cy in range(boxes[q, 1] // cell, boxes[q, 3] // cell + 1):
            for cx in range(boxes[q, 0] // cell, boxes[q, 2] // cell + 1):
                cands.update(grid.get((cx, cy), ()))
        for c in sorted(cands):
            if visited[c]:
                continue
            if boxes[q, 0] > boxes[c, 2] or boxes[c, 0] > boxes[q, 2] or boxes[q, 1] > boxes[c, 3] or boxes[c, 1] > boxes[q, 3]:
                continue                                   # STRtree query: bounding boxes intersect
            inter = polys[q].inter8(polys[c])
            uni = polys[q].area8 + polys[c].area8 - inter
            if uni > 0 and inter / uni > overlap_threshold:
                visited[c] = True
    return np.array(sorted(keep), np.int64)


def merge_overlap_masks(masks, scores, overlap_threshold=0.01):
    """Convenience for tests: masks [(bool crop, x0, y0)] -> rings by oracle/contour.py (`mask2inst`) -> merge_overlap."""
    from . import contour as OC
    rings = []
    for m, x0, y0 in masks:
        r = OC.mask2inst(m)
        rings.append(r + np.array([x0, y0], np.int64) if len(r) else r)
    return merge_overlap(rings, scores, overlap_threshold)
