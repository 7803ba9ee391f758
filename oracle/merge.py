"""CPU restatement of the cross-tile merge (test infrastructure: only tests/ import this).

Reference: tools/nuclei_merge.py:62-174 `merge_overlap(..., merge_strategy='probability')`: features sorted by score
(descending), each still-alive one suppresses every later one it overlaps with IoU > threshold.  The reference intersects
shapely polygons (GEOS, absent here); this build defines the overlap on the instance masks the polygons are traced from
(same objects, pixel-area instead of polygon-area IoU) -- parity with the reference's polygon IoU is therefore unpinned
(shapely absent), the sequential greedy semantics are restated exactly and the HIP kernel is checked against THIS function
bit for bit."""
import numpy as np


def merge_overlap(rec, overlap_threshold=0.05):
    """rec: dict(score=[n], mask=[(bool crop, x0, y0)]*n).  Returns the kept indices (ascending)."""
    n = len(rec['score'])
    if n == 0:
        return np.zeros(0, np.int64)
    order = np.argsort(-np.asarray(rec['score'], np.float32), kind='stable')
    boxes = np.array([[m[1], m[2], m[1] + m[0].shape[1], m[2] + m[0].shape[0]] for m in rec['mask']], np.int64)
    areas = [int(m[0].sum()) for m in rec['mask']]
    cell = 64
    grid = {}
    for i in range(n):
        if boxes[i, 2] <= boxes[i, 0] or boxes[i, 3] <= boxes[i, 1]:
            continue
        for cy in range(boxes[i, 1] // cell, (boxes[i, 3] - 1) // cell + 1):
            for cx in range(boxes[i, 0] // cell, (boxes[i, 2] - 1) // cell + 1):
                grid.setdefault((cx, cy), []).append(i)
    visited = np.zeros(n, bool)
    keep = []
    for q in order:
        if visited[q] or areas[q] == 0 and rec['mask'][q][0].size == 0:
            continue
        if rec['mask'][q][0].size == 0:
            continue
        visited[q] = True
        keep.append(q)
        mq, qx, qy = rec['mask'][q]
        cands = set()
        for cy in range(boxes[q, 1] // cell, (boxes[q, 3] - 1) // cell + 1):
            for cx in range(boxes[q, 0] // cell, (boxes[q, 2] - 1) // cell + 1):
                cands.update(grid.get((cx, cy), ()))
        for c in cands:
            if visited[c]:
                continue
            x0, y0 = max(boxes[q, 0], boxes[c, 0]), max(boxes[q, 1], boxes[c, 1])
            x1, y1 = min(boxes[q, 2], boxes[c, 2]), min(boxes[q, 3], boxes[c, 3])
            if x1 <= x0 or y1 <= y0:
                continue
            mc, cx0, cy0 = rec['mask'][c]
            inter = int(np.logical_and(mq[y0 - qy:y1 - qy, x0 - qx:x1 - qx], mc[y0 - cy0:y1 - cy0, x0 - cx0:x1 - cx0]).sum())
            union = areas[q] + areas[c] - inter
            if union > 0 and inter / union > overlap_threshold:
                visited[c] = True
    return np.array(sorted(keep), np.int64)
