"""COCO run-length masks: the small part of pycocotools.mask the reference's COCO output modes call
(tools/infer_wsi.py:600-627, tools/infer_patch.py:84-101,271-290: encode, toBbox, iou, the compressed `counts` string).

pycocotools (requirements.txt: pycocotools==2.0.7) is a third-party dependency that is absent from /root/reference and
from this image, so this file restates the published algorithm of cocoapi common/maskApi.c (rleEncode, rleDecode,
rleArea, rleToBbox, rleToString, rleFrString) -- PARITY UNPINNED against the real library; the tests pin it through
round trips, hand-worked vectors and an independent scalar restatement of the same C routines (oracle/rle.py,
tests/test_oracle_outputs.py).

An RLE is {'size': [h, w], 'counts': str}: run lengths over the mask in column-major (Fortran) order, starting with a
run of zeros (possibly empty)."""
import numpy as np


def _runs(mask):
    flat = np.asarray(mask).astype(bool).T.reshape(-1)          # column-major
    if flat.size == 0:
        return []
    change = np.flatnonzero(flat[1:] != flat[:-1]) + 1
    edges = np.concatenate([[0], change, [flat.size]])
    counts = np.diff(edges).tolist()
    if flat[0]:
        counts = [0] + counts
    return counts


def counts_to_string(counts):
    """maskApi.c rleToString: 5 data bits per character (+48), continuation bit 0x20, runs after the third are stored as
    differences from the run two places back."""
    out = []
    for i, x in enumerate(counts):
        x = int(x)
        if i > 2:
            x -= int(counts[i - 2])
        more = True
        while more:
            c = x & 0x1f
            x >>= 5
            more = (x != -1) if (c & 0x10) else (x != 0)
            if more:
                c |= 0x20
            out.append(chr(c + 48))
    return ''.join(out)


def string_to_counts(s):
    """maskApi.c rleFrString."""
    counts, p = [], 0
    b = s.encode('ascii') if isinstance(s, str) else bytes(s)
    while p < len(b):
        x, k, more = 0, 0, True
        while more:
            c = b[p] - 48
            x |= (c & 0x1f) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return counts


def encode(mask):
    """(H, W) 0/1 mask -> {'size': [H, W], 'counts': str}."""
    h, w = np.asarray(mask).shape
    return {'size': [int(h), int(w)], 'counts': counts_to_string(_runs(mask))}


def decode(rle):
    h, w = rle['size']
    counts = string_to_counts(rle['counts']) if not isinstance(rle['counts'], (list, tuple)) else list(rle['counts'])
    flat = np.zeros(h * w, np.uint8)
    pos, v = 0, 0
    for c in counts:
        if v:
            flat[pos:pos + c] = 1
        pos += c
        v ^= 1
    return flat.reshape(w, h).T.copy()


def area(rle):
    counts = string_to_counts(rle['counts'])
    return int(sum(counts[1::2]))


def to_bbox(rle):
    """maskApi.c rleToBbox: [x, y, w, h] of the set pixels (zeros for an empty mask)."""
    h, w = rle['size']
    counts = string_to_counts(rle['counts'])
    m = (len(counts) // 2) * 2
    if m == 0:
        return [0.0, 0.0, 0.0, 0.0]
    xs, ys, xe, ye, cc, xp = w, h, 0, 0, 0, 0
    for j in range(m):
        cc += counts[j]
        t = cc - j % 2
        y = t % h
        x = (t - y) // h
        if j % 2 == 0:
            xp = x
        elif xp < x:
            ys, ye = 0, h - 1
        xs, xe, ys, ye = min(xs, x), max(xe, x), min(ys, y), max(ye, y)
    return [float(xs), float(ys), float(xe - xs + 1), float(ye - ys + 1)]


def iou(dt, gt):
    """Pairwise IoU matrix (len(dt), len(gt)) of two RLE lists (iscrowd = 0)."""
    if len(dt) == 0 or len(gt) == 0:
        return np.zeros((len(dt), len(gt)))
    d = np.stack([decode(r).reshape(-1) for r in dt]).astype(np.float64)
    g = np.stack([decode(r).reshape(-1) for r in gt]).astype(np.float64)
    inter = d @ g.T
    union = d.sum(1)[:, None] + g.sum(1)[None, :] - inter
    return np.where(union > 0, inter / np.maximum(union, 1), 0.0)
