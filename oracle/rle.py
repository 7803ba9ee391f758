"""TEST INFRASTRUCTURE ONLY (checker; never imported by the product).

Independent restatement of the part of cocoapi `common/maskApi.c` (pycocotools 2.0.7, requirements.txt; the package is absent from
/root/reference and from this image: PARITY UNPINNED against the real library) that the reference's COCO outputs call --
tools/infer_wsi.py:611-616 (`maskUtils.encode`, `toBbox`) and tools/infer_patch.py:84-101 (`encode`, `iou`).  Written as the C
routines are: scalar loops over the column-major pixel stream and over runs, no numpy tricks, so that it shares no code path with
nuhtc_amd/cocomask.py (which it checks in tests/test_oracle_outputs.py) beyond the published format."""


def rle_encode(mask):
    """maskApi.c rleEncode: counts of alternating runs over the column-major pixels, the first run counts zeros."""
    h, w = len(mask), len(mask[0]) if len(mask) else 0
    cnts, c, p = [], 0, 0
    for x in range(w):
        for y in range(h):
            v = 1 if mask[y][x] else 0
            if v != p:
                cnts.append(c)
                c = 0
                p = v
            c += 1
    cnts.append(c)
    return h, w, cnts


def rle_to_string(cnts):
    """maskApi.c rleToString (LEB128-like: 5 data bits + continuation bit per char, offset 48; from the 4th run on the value is
    the difference to the run two places back)."""
    s = []
    for i in range(len(cnts)):
        x = cnts[i]
        if i > 2:
            x -= cnts[i - 2]
        more = 1
        while more:
            c = x & 0x1f
            x >>= 5
            more = (x != -1) if (c & 0x10) else (x != 0)
            if more:
                c |= 0x20
            c += 48
            s.append(chr(c))
    return ''.join(s)


def rle_fr_string(s):
    """maskApi.c rleFrString."""
    cnts, p, m = [], 0, 0
    while p < len(s):
        x, k, more = 0, 0, 1
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1f) << (5 * k)
            more = c & 0x20
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if m > 2:
            x += cnts[m - 2]
        cnts.append(x)
        m += 1
    return cnts


def rle_area(cnts):
    """maskApi.c rleArea: the odd runs."""
    return sum(cnts[j] for j in range(1, len(cnts), 2))


def rle_to_bbox(h, w, cnts):
    """maskApi.c rleToBbox -> [x, y, w, h] as doubles."""
    m = (len(cnts) // 2) * 2
    if m == 0:
        return [0.0, 0.0, 0.0, 0.0]
    xs, ys, xe, ye, cc, xp = w, h, 0, 0, 0, 0
    for j in range(m):
        cc += cnts[j]
        t = cc - j % 2
        y = t % h
        x = (t - y) // h
        if j % 2 == 0:
            xp = x
        elif xp < x:
            ys = 0
            ye = h - 1
        xs = min(xs, x); xe = max(xe, x); ys = min(ys, y); ye = max(ye, y)
    return [float(xs), float(ys), float(xe - xs + 1), float(ye - ys + 1)]


def rle_iou(a, b):
    """maskApi.c rleIou for one pair of run lists of equal size (iscrowd = 0): walks both run sequences in step."""
    ka, kb = len(a), len(b)
    if ka == 0 or kb == 0:
        return 0.0
    ca, cb = a[0], b[0]
    va = vb = 0
    ia = ib = 1
    inter = union = 0
    ct = 1
    while ct > 0:
        c = min(ca, cb)
        if va or vb:
            union += c
            if va and vb:
                inter += c
        ca -= c
        if not ca and ia < ka:
            ca = a[ia]; ia += 1; va = 1 - va
        ct = ca
        cb -= c
        if not cb and ib < kb:
            cb = b[ib]; ib += 1; vb = 1 - vb
        ct += cb
    return inter / union if union else 0.0


def encode(mask):
    """pycocotools.mask.encode of one (H, W) mask -> {'size': [h, w], 'counts': str}."""
    h, w, cnts = rle_encode(mask)
    return {'size': [h, w], 'counts': rle_to_string(cnts)}
