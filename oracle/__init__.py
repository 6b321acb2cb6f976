"""CPU oracle for the ISP hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the
product package (adaptiveisp_amd) never does. See isp_oracle.c for the reference citations.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

OPS = dict(ZERO=-1, EXPOSURE=0, GAMMA=1, CCM=2, SHARPEN=3, NLM=4, TONE=5, CONTRAST=6, SATPLUS=7, WNB=8, WB=9,
           USM=10, SHARPEN_V2=11, COLOR=12)


def build(force=False):
    so = os.path.join(_HERE, "libisp_oracle.so")
    src = os.path.join(_HERE, "isp_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libisp_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libisp_oracle.so")
        if not os.path.exists(so):
            build()
        L = ctypes.CDLL(so)
        f32p, i32p = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32)
        L.oracle_forward.argtypes = [f32p, f32p, i32p, f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_uint]
        L.oracle_forward.restype = ctypes.c_int
        L.oracle_pool64.argtypes = [f32p, f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.oracle_pool64.restype = ctypes.c_int
        L.oracle_select_and_update.argtypes = [f32p, f32p, f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_int, ctypes.c_float, i32p, f32p]
        L.oracle_select_and_update.restype = ctypes.c_int
        L.oracle_demosaic.argtypes = [ctypes.POINTER(ctypes.c_uint16), f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_float, ctypes.c_float]
        L.oracle_demosaic.restype = ctypes.c_int
        L.oracle_nms.argtypes = [f32p, ctypes.c_int, ctypes.c_float, ctypes.c_int, i32p]
        L.oracle_nms.restype = ctypes.c_int
        L.oracle_nlm_general.argtypes = [f32p, f32p, f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.oracle_nlm_general.restype = ctypes.c_int
        L.oracle_num_params.argtypes = [ctypes.c_int]
        L.oracle_num_params.restype = ctypes.c_int
        _LIB = L
    return _LIB


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def forward(img, filter_id, params, clip=True):
    """img [B,3,H,W], filter_id [B] (op codes), params [B,n] -> out [B,3,H,W] (numpy fp32)."""
    img = _f32(img)
    B, C, H, W = img.shape
    assert C == 3
    ids = np.ascontiguousarray(np.broadcast_to(np.asarray(filter_id, dtype=np.int32), (B,)))
    params = _f32(params).reshape(B, -1)
    out = np.empty_like(img)
    rc = lib().oracle_forward(_fp(img), _fp(out), ids.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), _fp(params),
                              params.shape[1], B, H, W, 1 if clip else 0)
    if rc != 0:
        raise ValueError(f"oracle_forward failed: {rc}")
    return out


def pool64(img):
    img = _f32(img)
    B, C, H, W = img.shape
    out = np.empty((B, 3, 64, 64), np.float32)
    lib().oracle_pool64(_fp(img), _fp(out), B, H, W)
    return out


def select_and_update(pdf, u, states, train=False, forced=-1, test_steps=5.0):
    pdf, u, states = _f32(pdf), _f32(u).reshape(-1), _f32(states)
    B, F = pdf.shape
    sel = np.empty(B, np.int32)
    ns = np.empty_like(states)
    lib().oracle_select_and_update(_fp(pdf), _fp(u), _fp(states), B, F, 1 if train else 0, int(forced),
                                   float(test_steps), sel.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), _fp(ns))
    return sel, ns


def nlm_general(img, h, search, patch):
    """NonLocalMeansGray(search, patch).forward(img, h) (isp/denoise.py:93-119): img [B,3,H,W], h [B] -> [B,3,H,W]."""
    img = _f32(img)
    B, C, H, W = img.shape
    h = _f32(h).reshape(B)
    out = np.empty_like(img)
    if lib().oracle_nlm_general(_fp(img), _fp(out), _fp(h), B, H, W, int(search), int(patch)) != 0:
        raise ValueError("oracle_nlm_general: odd positive window sizes expected")
    return out


def nms(boxes, iou_thres, max_det=300):
    """boxes [n,4] xyxy sorted by descending score -> kept indices (int64 numpy), greedy IoU NMS."""
    boxes = _f32(boxes).reshape(-1, 4)
    n = boxes.shape[0]
    keep = np.empty(max(max_det, 1), np.int32)
    cnt = lib().oracle_nms(_fp(boxes), n, float(iou_thres), int(max_det), keep.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    if cnt < 0:
        raise MemoryError("oracle_nms")
    return keep[:cnt].astype(np.int64)


def demosaic(raw, pattern=0, black=0.0, white=65535.0):
    """raw uint16 [B,H,W] -> planar fp32 [B,3,H,W] (bilinear, mirrored borders); pattern = 2*ry + rx of the red sample."""
    raw = np.ascontiguousarray(raw, dtype=np.uint16)
    B, H, W = raw.shape
    out = np.empty((B, 3, H, W), np.float32)
    rc = lib().oracle_demosaic(raw.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)), _fp(out), B, H, W, int(pattern),
                               float(black), float(white))
    if rc != 0:
        raise ValueError("oracle_demosaic: H and W must be even, pattern in 0..3")
    return out
