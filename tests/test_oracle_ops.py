"""Known-answer tests for the restated third-party ops (mmcv RoIAlign/NMS are absent from the reference tree,
"parity unpinned" — SURVEY §8c) and the two reference tests that do pin numbers on the path."""
import numpy as np
import torch

from oracle import model as O
from oracle import ops, ops_np


def test_delta2bbox_reference_known_answer():
    # numbers from thirdparty/mmdetection/tests/test_utils/test_coder.py:27-75 (== doctest delta_xywh_bbox_coder.py:211-222)
    rois = torch.tensor([[0., 0., 1., 1.], [0., 0., 1., 1.], [0., 0., 1., 1.], [5., 5., 5., 5.]])
    deltas = torch.tensor([[0., 0., 0., 0.], [1., 1., 1., 1.], [0., 0., 2., -1.], [0.7, -1.9, -0.5, 0.3]])
    exp = torch.tensor([[0.0000, 0.0000, 1.0000, 1.0000], [0.1409, 0.1409, 2.8591, 2.8591],
                        [0.0000, 0.3161, 4.1945, 0.6839], [5.0000, 5.0000, 5.0000, 5.0000]])
    out = O.delta2bbox(rois, deltas, (1., 1., 1., 1.), (32, 32))
    assert torch.allclose(out, exp, atol=1e-4)


def test_anchor_known_answer():
    # thirdparty/mmdetection/tests/test_utils/test_anchor.py:287-305: AnchorGenerator([10],[1.],[1.],[10]) on a 2x2 map
    # -> our generator is specialised to scales=[4], ratios (.5,1,2); check the ratio-1 anchor of stride 10 scaled by 4
    a = O.anchors_for_level(2, 2, 10).view(4, 3, 4)[:, 1]   # ratio 1.0 is the middle base anchor
    exp = torch.tensor([[-20., -20., 20., 20.], [-10., -20., 30., 20.], [-20., -10., 20., 30.], [-10., -10., 30., 30.]])
    assert torch.equal(a, exp)
    # ratio semantics: h/w = ratio, area = (stride*scale)^2
    b = O.anchors_for_level(1, 1, 8)
    w, h = b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]
    assert torch.allclose(h / w, torch.tensor([0.5, 1.0, 2.0]), atol=1e-6)
    assert torch.allclose(w * h, torch.full((3,), 1024.0), atol=1e-3)


def test_roi_align_known_answers():
    # linear ramp f(y,x) = 10*y + x is reproduced exactly by bilinear interpolation away from the border
    H, W = 16, 16
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    feat = (10 * yy + xx)[None, None]
    roi = np.array([[0, 4.0, 4.0, 12.0, 8.0]], np.float32)   # scale 1 -> x in [3.5, 11.5], y in [3.5, 7.5]
    for impl in (ops, ops_np):
        out = impl.roi_align(feat, roi, 2, 1.0, 2)
        # bin centres: x = 3.5 + (pw+.5)*4 -> 5.5, 9.5 ; y = 3.5 + (ph+.5)*2 -> 4.5, 6.5 (mean of symmetric samples)
        exp = np.array([[10 * 4.5 + 5.5, 10 * 4.5 + 9.5], [10 * 6.5 + 5.5, 10 * 6.5 + 9.5]], np.float32)
        np.testing.assert_allclose(out[0, 0], exp, atol=1e-4)
        # adaptive grid: roi_w/P = 4 -> 4 samples per bin per axis; same mean for a linear ramp
        np.testing.assert_allclose(impl.roi_align(feat, roi, 2, 1.0, 0)[0, 0], exp, atol=1e-4)
        # samples further than 1 px outside contribute 0; a far-outside roi gives all zeros
        far = np.array([[0, -40.0, -40.0, -30.0, -30.0]], np.float32)
        assert (impl.roi_align(feat, far, 2, 1.0, 2) == 0).all()
        # zero-size roi (aligned=True has no min-size clamp): adaptive grid 0 -> count=max(0,1), empty sum -> 0
        z = np.array([[0, 5.0, 5.0, 5.0, 5.0]], np.float32)
        assert (impl.roi_align(feat, z, 2, 1.0, 0) == 0).all()
        # sampling_ratio 2 on a zero-size roi samples the point itself
        np.testing.assert_allclose(impl.roi_align(feat, z, 2, 1.0, 2)[0, 0], np.full((2, 2), 10 * 4.5 + 4.5), atol=1e-4)


def test_roi_align_c_matches_numpy_random():
    rng = np.random.default_rng(3)
    feat = rng.standard_normal((2, 5, 19, 23)).astype(np.float32)
    xy = rng.uniform(-12, 90, (64, 2))
    wh = rng.uniform(0.5, 70, (64, 2))
    rois = np.concatenate([rng.integers(0, 2, (64, 1)), xy, xy + wh], 1).astype(np.float32)
    for P, sr, sc in [(7, 2, 0.25), (14, 0, 0.25), (7, 2, 0.125)]:
        np.testing.assert_array_equal(ops.roi_align(feat, rois, P, sc, sr), ops_np.roi_align(feat, rois, P, sc, sr))


def test_nms_known_answers():
    boxes = np.array([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 5], [0, 0, 10, 10]], np.float32)
    scores = np.array([0.9, 0.8, 0.7, 0.6, 0.9], np.float32)
    # IoU(0,1) = 81/119 = .68 ; IoU(0,3) = .5 exactly ; box 4 duplicates box 0 (tie -> lower index first)
    for impl in (ops, ops_np):
        np.testing.assert_array_equal(impl.nms(boxes, scores, 0.5), [0, 2, 3])     # strict '>' keeps the IoU==0.5 box
        np.testing.assert_array_equal(impl.nms(boxes, scores, 0.49), [0, 2])
        np.testing.assert_array_equal(impl.nms(boxes, scores, 0.7), [0, 1, 2, 3])
        assert len(impl.nms(boxes[:0], scores[:0], 0.5)) == 0
        # batched: identical boxes with different ids never suppress each other
        dets, keep = impl.batched_nms(boxes[[0, 4]], scores[[0, 4]], np.array([0, 1]), 0.5)
        assert len(keep) == 2 and dets.shape == (2, 5)


def test_nms_c_matches_numpy_random():
    rng = np.random.default_rng(5)
    xy = rng.uniform(0, 200, (700, 2))
    wh = rng.uniform(5, 60, (700, 2))
    boxes = np.concatenate([xy, xy + wh], 1).astype(np.float32)
    scores = np.round(rng.uniform(0, 1, 700), 2).astype(np.float32)   # many ties
    for thr in (0.3, 0.5, 0.7):
        np.testing.assert_array_equal(ops.nms(boxes, scores, thr), ops_np.nms(boxes, scores, thr))


def test_resize2x_properties():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (9, 11, 3), dtype=np.uint8)
    out = O.resize2x_u8(img)
    assert out.shape == (18, 22, 3) and out.dtype == np.uint8
    const = np.full((5, 5, 3), 77, np.uint8)
    assert (O.resize2x_u8(const) == 77).all()
    # corners replicate (edge clamp): out[0,0] == in[0,0]
    assert (out[0, 0] == img[0, 0]).all() and (out[-1, -1] == img[-1, -1]).all()
    # within 1 LSB of float bilinear with half-pixel centres
    f = torch.nn.functional.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None].float(), scale_factor=2,
                                        mode='bilinear', align_corners=False)[0].permute(1, 2, 0).numpy()
    assert np.abs(out.astype(np.float32) - f).max() <= 1.0


def _cv_resize_scalar(img, dw, dh):
    """Literal scalar transcription of OpenCV's 8-bit INTER_LINEAR loops (one channel), independent of the vectorised
    oracle: table loop of cv::resize, HResizeLinear, VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>."""
    H, W = img.shape
    sx_, a_ = [], []
    scale_x, scale_y = 1.0 / (dw / W), 1.0 / (dh / H)
    for dx in range(dw):
        fx = np.float32((dx + 0.5) * scale_x - 0.5)
        sx = int(np.floor(fx))
        fx = np.float32(fx - np.float32(sx))
        if sx < 0:
            fx, sx = np.float32(0), 0
        if sx >= W - 1:
            fx, sx = np.float32(0), W - 1
        sx_.append(sx)
        a_.append((int(np.rint((np.float32(1) - fx) * np.float32(2048))), int(np.rint(fx * np.float32(2048)))))
    out = np.zeros((dh, dw), np.uint8)
    for dy in range(dh):
        fy = np.float32((dy + 0.5) * scale_y - 0.5)
        sy = int(np.floor(fy))
        fy = np.float32(fy - np.float32(sy))
        b0, b1 = int(np.rint((np.float32(1) - fy) * np.float32(2048))), int(np.rint(fy * np.float32(2048)))
        r0, r1 = min(max(sy, 0), H - 1), min(max(sy + 1, 0), H - 1)
        for dx in range(dw):
            sx = sx_[dx]
            sx1 = min(sx + 1, W - 1)
            S0 = int(img[r0, sx]) * a_[dx][0] + int(img[r0, sx1]) * a_[dx][1]
            S1 = int(img[r1, sx]) * a_[dx][0] + int(img[r1, sx1]) * a_[dx][1]
            out[dy, dx] = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2
    return out


def test_resize_cv2_fixed_point_known_answers():
    # 1) the two vertical products are truncated separately: rows (1,2) over (1,0), output pixel (2,2) takes
    #    far row (1,2) weights (1,3)/4 -> (1+6)>>2 = 1, near row (1,0) weights (3,9)/4 -> 3>>2 = 0, (1+0+2)>>2 = 0
    #    (summing the four taps first would give ((10>>2)+2)>>2 = 1)
    img = np.array([[1, 2, 0], [1, 0, 0], [0, 0, 0]], np.uint8)
    out = O.resize2x_u8(img)
    assert out[2, 2] == 0
    # 2) hand-worked interior pixel: a,b = 10,200 (far row), c,d = 90,35 (near row), even/even output
    img = np.array([[10, 200], [90, 35]], np.uint8)
    out = O.resize2x_u8(img)
    #    out[2,2]: far row = row 0, far col = col 0:  ((10+600)>>2) + ((270+315)>>2) + 2 >> 2 = (152+146+2)>>2 = 75
    assert out[2, 2] == 75
    #    out[1,1] (odd/odd): far row = row 1 (weights swap): rows (0: near, 1: far), cols (0 near, 1 far)
    #    near row 0: 3*(3*10+200)=690>>2=172 ; far row 1: (3*90+35)=305>>2=76 ; (172+76+2)>>2 = 62
    assert out[1, 1] == 62
    # 3) left/right border columns take one source column at weight 2048 (fx reset to 0) ...
    #    (H>>4 = 128*v there, so the far row contributes exactly v and the near row exactly 3v)
    assert out[2, 0] == (10 + 3 * 90 + 2) >> 2 == 70
    assert out[2, 3] == (200 + 3 * 35 + 2) >> 2 == 76
    #    ... while the top/bottom border rows keep weights (1,3)/4 on the same clamped row: (v>>2) + (3v>>2)
    assert out[0, 0] == 10 and out[3, 3] == 35
    assert out[0, 2] == ((((10 + 600) >> 2) + ((3 * (10 + 600)) >> 2) + 2) >> 2)   # 152 + 457 + 2 >> 2 = 152
    # 4) vectorised oracle == literal scalar loops, x2 / x4 / non-integer factors, odd sizes
    rng = np.random.default_rng(5)
    for (h, w, dh, dw) in [(7, 9, 14, 18), (5, 6, 20, 24), (9, 7, 13, 10), (8, 8, 3, 5), (6, 6, 6, 6)]:
        a = rng.integers(0, 256, (h, w), dtype=np.uint8)
        np.testing.assert_array_equal(O.cv2_resize_linear_u8(a, dw, dh), _cv_resize_scalar(a, dw, dh))
    a3 = rng.integers(0, 256, (6, 5, 3), dtype=np.uint8)
    for c in range(3):
        np.testing.assert_array_equal(O.cv2_resize_linear_u8(a3, 20, 24)[:, :, c], _cv_resize_scalar(a3[:, :, c], 20, 24))


def test_cc_proposals_matches_scipy_pipeline():
    # the "watershed" step reduces to hole-filled 4-connected components: check against the literal
    # scipy sequence of the reference (fill -> EDT -> label(dist>0.25)) on a random blob mask
    from scipy import ndimage as ndi
    rng = np.random.default_rng(1)
    pred = torch.from_numpy(ndi.gaussian_filter(rng.standard_normal((1, 1, 32, 32)), 2.0).astype(np.float32)) * 20
    boxes = O.cc_proposals(pred, (128, 128))[0].numpy()
    m = O.semantic_binary_mask(pred, (128, 128))[0].numpy()
    filled = ndi.binary_fill_holes(m)
    dist = ndi.distance_transform_edt(filled)
    markers, n = ndi.label(dist > 0.25)
    exp = []
    for i in range(1, n + 1):
        ys, xs = np.nonzero(markers == i)
        if 10 < len(ys) < 128 * 128 / 4:
            exp.append([xs.min(), ys.min(), xs.max() + 1, ys.max() + 1, 1.0])
    np.testing.assert_array_equal(boxes, np.array(exp, np.float32).reshape(-1, 5))
    assert len(exp) > 0
