"""ORACLE — test infrastructure, NOT product code.

`cv2.findContours(mask, cv2.RETR_TREE, cv2.CHAIN_APPROX_SIMPLE)` restated: the reference's `mask2inst`
(tools/infer_wsi.py:51-54) takes `[0][0]` of it — the first contour of the returned list — closes the ring by repeating
its first point (:53) and adds the tile origin (:56-58, :533-539).

OpenCV (opencv-python 4.11.0.86, README.md:90) is third-party code absent from /root/reference and from this image:
PARITY UNPINNED.  What is restated here is the published algorithm — S. Suzuki, K. Abe, "Topological structural analysis
of digitized binary images by border following", CVGIP 30 (1985), Algorithm 1, which `imgproc/src/contours.cpp` implements
— with OpenCV's conventions:

  * the image is taken with a 1-pixel frame of zeros (cv::findContours pads since 3.2), so blobs may touch the tile edge;
  * raster scan, rows top to bottom; an outer border starts at a 1-pixel whose left neighbour is 0, a hole border at a
    pixel >= 1 whose right neighbour is 0 (labels NBD / -NBD as in the paper);
  * neighbour codes are Freeman chain codes, 0 = east, counted counter-clockwise on the screen (1 = north-east, 2 = north,
    ... 7 = south-east; `icvCodeDeltas`); step (3.1) searches clockwise (decreasing code) from the start's zero neighbour
    (west for an outer border, east for a hole), step (3.3) counter-clockwise (increasing code) from the code after the one
    pointing back to the previous pixel; the walk ends when it steps from (i1,j1) back into the start pixel (3.5);
    a filled axis-aligned rectangle therefore comes out as top-left, bottom-left, bottom-right, top-right;
  * CHAIN_APPROX_SIMPLE (`icvFetchContour`, method 1): the current point is written when the code of the step leaving it
    differs from the code of the step that entered it; for the start pixel the "entering" code is initialised to the code
    of (i1,j1) seen from the start, xor 4; an isolated pixel is one point;
  * hierarchy per the paper's table (parent of an outer border = the hole it lies in, or the frame);
    `cvInsertNodeIntoTree` links every new border at the HEAD of its parent's child list, and the returned list is the
    pre-order walk of that tree from the frame's first child, so index 0 is the top-level outer border that was found LAST
    in the scan (for several top-level blobs: the one whose first pixel in raster order comes last).

Pinned by hand-worked cases in tests/test_oracle_contour.py (rectangle, L-shape, diagonal, one-pixel spur, hole, two blobs,
nested island).
"""
import numpy as np

# Freeman codes: 0 = E, counter-clockwise on the screen (x right, y down)
DX = (1, 1, 0, -1, -1, -1, 0, 1)
DY = (0, -1, -1, -1, 0, 1, 1, 1)


class Border:
    __slots__ = ('nbd', 'is_hole', 'parent', 'children', 'points', 'start')

    def __init__(self, nbd, is_hole, parent):
        self.nbd, self.is_hole, self.parent, self.children, self.points, self.start = nbd, is_hole, parent, [], [], None


def _follow(f, i, j, nbd, is_hole, simple=True):
    """Steps (3.1)-(3.5) from start pixel (row i, col j) of the padded label image f; returns the point list (x, y) in
    padded coordinates."""
    s_end = s = 0 if is_hole else 4
    while True:                                   # (3.1) clockwise from the zero neighbour
        s = (s - 1) & 7
        if f[i + DY[s], j + DX[s]] != 0 or s == s_end:
            break
    if s == s_end:                                # isolated pixel
        f[i, j] = -nbd
        return [(j, i)]
    i1, j1 = i + DY[s], j + DX[s]
    pts = []
    i3, j3 = i, j
    prev_s = s ^ 4
    while True:
        s_end = s
        while True:                               # (3.3) counter-clockwise, starting after the code back to the previous pixel
            s += 1
            i4, j4 = i3 + DY[s & 7], j3 + DX[s & 7]
            if f[i4, j4] != 0:
                break
        s &= 7
        # (3.4): the east neighbour (code 0 = 8) was examined as a 0-pixel iff the search wrapped past it before stopping
        # (contours.cpp: `(unsigned)(s - 1) < (unsigned)s_end`)
        if s != 0 and s - 1 < s_end:
            f[i3, j3] = -nbd
        elif f[i3, j3] == 1:
            f[i3, j3] = nbd
        if not simple or s != prev_s:
            pts.append((j3, i3))
            prev_s = s
        if (i4, j4) == (i, j) and (i3, j3) == (i1, j1):   # (3.5)
            break
        i3, j3 = i4, j4
        s = (s + 4) & 7
    return pts


def _scan(mask, simple=True):
    """The raster scan of Algorithm 1 with the full (RETR_TREE) parent rule -> the frame border; every border hangs in its
    parent's `children` (newest first) and carries `start`, the pixel (x, y) at which the scan discovered it."""
    m = np.asarray(mask) != 0
    H, W = m.shape
    f = np.zeros((H + 2, W + 2), np.int64)
    f[1:-1, 1:-1] = m
    frame = Border(1, True, None)
    borders = {1: frame}
    nbd = 1
    for i in range(1, H + 1):
        lnbd = 1
        for j in range(1, W + 1):
            v = f[i, j]
            if v == 0:
                continue
            is_hole = None
            if v == 1 and f[i, j - 1] == 0:
                is_hole = False
            elif v >= 1 and f[i, j + 1] == 0:
                is_hole = True
                if v > 1:
                    lnbd = v
            if is_hole is not None:
                nbd += 1
                prev = borders[lnbd]
                # paper, Table 1: same type -> sibling (parent of the previous border), different type -> child of it
                parent = prev.parent if prev.is_hole == is_hole else prev
                if parent is None:
                    parent = frame
                b = Border(nbd, is_hole, parent)
                b.points = _follow(f, i, j, nbd, is_hole, simple)
                b.start = (j - 1, i - 1)
                parent.children.insert(0, b)          # cvInsertNodeIntoTree: new node becomes the first child
                borders[nbd] = b
            if f[i, j] != 1:
                lnbd = abs(int(f[i, j]))
    return frame


def _flatten(frame):
    """Pre-order walk from the frame's first child -> (contours, hierarchy [next, previous, first_child, parent])."""
    order = []

    def walk(b):
        order.append(b)
        for c in b.children:
            walk(c)
    for c in frame.children:
        walk(c)
    idx = {id(b): k for k, b in enumerate(order)}
    contours = [np.array(b.points, np.int64) - 1 for b in order]   # remove the frame offset
    hier = np.full((len(order), 4), -1, np.int64)
    for k, b in enumerate(order):
        sib = b.parent.children
        p = sib.index(b)
        if p + 1 < len(sib):
            hier[k, 0] = idx[id(sib[p + 1])]
        if p > 0:
            hier[k, 1] = idx[id(sib[p - 1])]
        if b.children:
            hier[k, 2] = idx[id(b.children[0])]
        if b.parent is not frame:
            hier[k, 3] = idx[id(b.parent)]
    return contours, hier


def find_contours_tree(mask, simple=True):
    """-> (contours, hierarchy) like cv2.findContours(mask, RETR_TREE, CHAIN_APPROX_SIMPLE | NONE): contours is a list of
    (n,2) int arrays (x, y); hierarchy (n,4) = [next, previous, first_child, parent] indices (-1 = none)."""
    return _flatten(_scan(mask, simple))


def find_contours_ccomp(mask, simple=False):
    """cv2.findContours(mask, RETR_CCOMP, CHAIN_APPROX_NONE | SIMPLE) (`segmentTissue`, tools/wsi_core/WholeSlideImage.py:187):
    the same scan and the same borders, in the two-level hierarchy of RETR_CCOMP -- every outer border hangs on the frame
    (an island inside a hole too), every hole border on the outer border of the component it is a hole of (`contours.cpp`:
    in CCOMP mode `par_info` of a non-hole is the frame; a hole keeps the parent the paper's table gives it, which is an outer
    border).  Each new border is still linked at the HEAD of its parent's list, so the pre-order walk lists the outer borders
    in the REVERSE of the order the scan found them (the component whose first raster pixel comes last is index 0), each
    followed by its holes, newest first."""
    frame = _scan(mask, simple)
    allb = []

    def collect(b):
        for c in b.children:
            allb.append(c)
            collect(c)
    collect(frame)
    found = sorted(allb, key=lambda b: b.nbd)        # NBD counts the borders in the order the scan found them
    top = Border(1, True, None)
    for b in found:
        b.children = []
    for b in found:
        par = top if not b.is_hole else b.parent   # a hole's table parent is an outer border (never the frame: the frame pads the image)
        b.parent = par
        par.children.insert(0, b)
    return _flatten(top)


def mask2inst(mask):
    """tools/infer_wsi.py:51-54: first contour of the list, ring closed by repeating its first point -> (n+1, 2) int (x, y).
    (The reference indexes [0] unconditionally: an empty mask raises there; here an empty (0,2) array is returned.)"""
    contours, _ = find_contours_tree(mask, simple=True)
    if not contours:
        return np.zeros((0, 2), np.int64)
    c = contours[0]
    return np.concatenate([c, c[:1]], 0)
