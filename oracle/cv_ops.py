"""TEST INFRASTRUCTURE ONLY (checker; never imported by the product).

Independent restatements of the OpenCV 4.x primitives behind the reference's tissue segmentation (`segmentTissue`,
tools/wsi_core/WholeSlideImage.py:105-199, and the contour checks of tools/wsi_core/util_classes.py:48-111): scalar loops that
follow the published OpenCV sources (modules/imgproc/src/color_hsv.dispatch.cpp `RGB2HSV_b`, median_blur, morph.dispatch.cpp
`morphOp`, thresh.cpp `getThreshVal_Otsu_8u`, shapedescr.cpp `contourArea`, geometry.cpp `pointPolygonTest`).  cv2 is absent from
this image (PARITY UNPINNED against the library itself); these exist so that nuhtc_amd/tissue.py -- numpy / scipy, vectorised -- is
checked by something that is not itself (tests/test_oracle_cv.py), on hand-worked values and on random inputs."""
import math

FLT_EPSILON = 1.1920929e-07


def saturation_u8(rgb):
    """S of COLOR_RGB2HSV for 8-bit input (RGB2HSV_b, hsv_shift = 12): s = (diff * sdiv_table[v] + (1 << 11)) >> 12 with
    sdiv_table[v] = saturate_cast<int>((255 << 12) / (1. * v)) (round to nearest), sdiv_table[0] = 0."""
    h, w = len(rgb), len(rgb[0])
    sdiv = [0] + [int(round((255 << 12) / float(i))) for i in range(1, 256)]
    out = [[0] * w for _ in range(h)]
    for y in range(h):
        for x in range(w):
            r, g, b = (int(c) for c in rgb[y][x][:3])
            v = max(r, g, b)
            diff = v - min(r, g, b)
            out[y][x] = (diff * sdiv[v] + (1 << 11)) >> 12
    return out


def median_blur(img, k):
    """cv2.medianBlur (odd k): median of the k x k neighbourhood, BORDER_REPLICATE."""
    h, w = len(img), len(img[0])
    r = k // 2
    out = [[0] * w for _ in range(h)]
    for y in range(h):
        for x in range(w):
            vals = []
            for dy in range(-r, r + 1):
                yy = min(max(y + dy, 0), h - 1)
                for dx in range(-r, r + 1):
                    xx = min(max(x + dx, 0), w - 1)
                    vals.append(img[yy][xx])
            vals.sort()
            out[y][x] = vals[len(vals) // 2]
    return out


def _morph(img, k, dilate):
    """One pass of morphOp with a k x k rectangle, default anchor (k // 2, k // 2), BORDER_CONSTANT with
    morphologyDefaultBorderValue (the border never wins).  erode: min over src(x + i - anchor); dilate: the kernel is flipped and
    the anchor mirrored (anchor' = k - 1 - anchor), i.e. max over src(x + i - anchor')."""
    h, w = len(img), len(img[0])
    a = k // 2
    if dilate:
        a = k - 1 - a
    out = [[0] * w for _ in range(h)]
    for y in range(h):
        for x in range(w):
            best = None
            for i in range(k):
                yy = y + i - a
                if yy < 0 or yy >= h:
                    continue
                for j in range(k):
                    xx = x + j - a
                    if xx < 0 or xx >= w:
                        continue
                    v = img[yy][xx]
                    if best is None or (v > best if dilate else v < best):
                        best = v
            out[y][x] = best if best is not None else (0 if dilate else 255)
    return out


def morph_close(img, k):
    """cv2.morphologyEx(img, MORPH_CLOSE, ones((k, k))): dilate, then erode."""
    return _morph(_morph(img, k, True), k, False)


def otsu_threshold(img):
    """getThreshVal_Otsu_8u."""
    hist = [0] * 256
    n = 0
    for row in img:
        for v in row:
            hist[int(v)] += 1
            n += 1
    scale = 1.0 / n
    mu = sum(i * hist[i] for i in range(256)) * scale
    mu1 = q1 = 0.0
    max_sigma, max_val = 0.0, 0
    for i in range(256):
        p_i = hist[i] * scale
        mu1 *= q1
        q1 += p_i
        q2 = 1.0 - q1
        if min(q1, q2) < FLT_EPSILON or max(q1, q2) > 1.0 - FLT_EPSILON:
            continue
        mu1 = (mu1 + i * p_i) / q1
        mu2 = (mu - q1 * mu1) / q2
        sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2)
        if sigma > max_sigma:
            max_sigma, max_val = sigma, i
    return max_val


def contour_area(contour):
    """cv2.contourArea(contour) (oriented = False): |sum of cross products| / 2."""
    n = len(contour)
    if n == 0:
        return 0.0
    a00 = 0.0
    px, py = float(contour[n - 1][0]), float(contour[n - 1][1])
    for i in range(n):
        x, y = float(contour[i][0]), float(contour[i][1])
        a00 += px * y - x * py
        px, py = x, y
    return abs(a00 * 0.5)


def point_polygon_test(contour, pt):
    """cv2.pointPolygonTest(contour, pt, measureDist=False) for an integer contour and an integer-valued point (the integer branch
    of geometry.cpp): +1 inside, 0 on the boundary, -1 outside."""
    n = len(contour)
    if n == 0:
        return -1
    px, py = int(round(pt[0])), int(round(pt[1]))
    counter = 0
    v0x, v0y = int(contour[n - 1][0]), int(contour[n - 1][1])
    for i in range(n):
        vx, vy = int(contour[i][0]), int(contour[i][1])
        ox, oy = v0x, v0y                     # v0 of this edge
        v0x, v0y = vx, vy
        if (oy <= py and vy <= py) or (oy > py and vy > py) or (ox < px and vx < px):
            if py == vy and (px == vx or (py == oy and ((ox <= px <= vx) or (vx <= px <= ox)))):
                return 0
            continue
        dist = (py - oy) * (vx - ox) - (px - ox) * (vy - oy)
        if dist == 0:
            return 0
        if vy < oy:
            dist = -dist
        counter += dist > 0
    return -1 if counter % 2 == 0 else 1
