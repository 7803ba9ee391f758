"""ORACLE (test infrastructure). ctypes front-end of oracle/ops_c.c (plain-C restatement of mmcv RoIAlign / NMS);
same API as oracle/ops_np.py, which stays as an independent cross-check."""
import ctypes
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, 'libnuhtc_oracle.so')
_lib = None
f32 = np.float32


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_DIR, 'ops_c.c')):
        subprocess.check_call(['make', '-C', _DIR, '-B', 'libnuhtc_oracle.so'], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        fp = ctypes.POINTER(ctypes.c_float)
        _lib.roi_align_forward.argtypes = [fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_float, ctypes.c_int, fp]
        _lib.roi_align_forward.restype = None
        _lib.nms_f32.argtypes = [fp, fp, ctypes.c_int, ctypes.c_float, ctypes.POINTER(ctypes.c_int64)]
        _lib.nms_f32.restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def roi_align(feat, rois, out_size, spatial_scale, sampling_ratio):
    feat = np.ascontiguousarray(feat, dtype=f32)
    rois = np.ascontiguousarray(rois, dtype=f32)
    N, C, H, W = feat.shape
    R = rois.shape[0]
    out = np.zeros((R, C, out_size, out_size), f32)
    if R:
        lib().roi_align_forward(_p(feat), N, C, H, W, _p(rois), R, int(out_size), float(spatial_scale),
                                int(sampling_ratio), _p(out))
    return out


def nms(boxes, scores, iou_thr):
    boxes = np.ascontiguousarray(boxes, dtype=f32)
    scores = np.ascontiguousarray(scores, dtype=f32)
    n = boxes.shape[0]
    keep = np.zeros(max(n, 1), np.int64)
    k = lib().nms_f32(_p(boxes), _p(scores), n, float(iou_thr), keep.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))) if n else 0
    return keep[:k].copy()


def batched_nms(boxes, scores, idxs, iou_thr):
    """mmcv batched_nms, class_agnostic=False, N < split_thr: per-id coordinate offset in float32."""
    boxes = np.asarray(boxes, dtype=f32)
    scores = np.asarray(scores, dtype=f32)
    if boxes.shape[0] == 0:
        return np.zeros((0, 5), f32), np.zeros((0,), np.int64)
    off = np.asarray(idxs).astype(f32) * (boxes.max() + f32(1))
    keep = nms(boxes + off[:, None], scores, iou_thr)
    return np.concatenate([boxes[keep], scores[keep, None]], 1), keep
