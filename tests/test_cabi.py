"""The C-ABI library loads and exports every symbol include/adaisp.h declares (no compute: no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ada(?:isp|yolo)_\w+)\s*\(", text)))


def test_libadaisp_exports_header_symbols():
    from adaptiveisp_amd import _lib
    L = _lib.load()
    names = _declared("adaisp.h")
    assert set(_lib.EXPORTS) <= set(names)
    for n in names:
        assert hasattr(L, n), f"libadaisp.so does not export {n}"
    assert L.adaisp_abi_version() == _lib.ABI_VERSION
    assert L.adaisp_strerror(0) == b"ok" and b"alias" in L.adaisp_strerror(-3)
    expect = {-1: 0, 0: 1, 1: 1, 2: 9, 3: 1, 4: 1, 5: 8, 6: 1, 7: 1, 8: 1, 9: 3, 10: 2, 11: 1, 12: 24, 13: -1}
    for op, n in expect.items():
        assert L.adaisp_num_params(op) == n


def test_argument_checks_without_gpu():
    from adaptiveisp_amd import _lib
    L = _lib.load()
    # null pointers / bad sizes are rejected before anything touches a device
    assert L.adaisp_process(0, None, None, None, 1, 1, 8, 8, 0, None) == -1
    assert L.adaisp_forward(None, None, None, None, None, 1, 1, 8, 8, 0, None) == -1
    assert L.adaisp_pool64(None, None, 1, 8, 8, None) == -1
    buf = (ctypes.c_float * 16)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert L.adaisp_process(99, p, p, p, 1, 1, 2, 2, 0, None) == -2          # unknown op
    assert L.adaisp_process(2, p, p, p, 1, 1, 2, 2, 0, None) == -1           # CCM needs 9 params, stride 1
    assert L.adaisp_process(3, p, p, p, 1, 1, 2, 2, 0, None) == -3           # stencil in place
    # the host-known-op form of the RL step: same checks as adaisp_process, and never in place (it may be a stencil step)
    q = ctypes.cast((ctypes.c_float * 16)(), ctypes.c_void_p)
    assert L.adaisp_forward_uniform(0, None, None, None, None, 1, 1, 8, 8, 0, None) == -1
    assert L.adaisp_forward_uniform(99, p, q, None, p, 1, 1, 2, 2, 0, None) == -2
    assert L.adaisp_forward_uniform(2, p, q, None, p, 1, 1, 2, 2, 0, None) == -1
    assert L.adaisp_forward_uniform(0, p, p, None, p, 1, 1, 2, 2, 0, None) == -3
    assert L.adaisp_forward_uniform(0, p, q, None, p, 1, 1, 2, 2, 0, None) == -4          # H, W < 3


def test_fused_pooling_geometry_is_host_side_and_total():
    """The cut of the fused-pooling kernels (isp_internal.h: PoolGeom), restated: every pixel column is owned by exactly one
    strip, every strip's read span covers what it owns and fits 64 lanes x 4 px, for the BASELINE sizes and awkward widths."""
    def win_lo(o, n): return (o * n) // 64
    def win_hi(o, n): return ((o + 1) * n + 63) // 64
    for W in (64, 68, 72, 100, 128, 200, 512, 640, 1000, 1280, 1920, 3840, 4096):
        geom = None
        for strips in range(1, 17):
            cps = (64 + strips - 1) // strips
            if (64 + cps - 1) // cps != strips:
                continue
            c0 = lambda s: min(s * cps, 64)                                            # noqa: E731
            lo = lambda s: W if c0(s) >= 64 else win_lo(c0(s), W) & ~3                  # noqa: E731
            end = lambda s: (win_hi(c0(s + 1) - 1, W) + 3) & ~3                         # noqa: E731
            if all(end(s) - lo(s) <= 256 for s in range(strips)):
                geom = (cps, strips, lo, end)
                break
        assert geom is not None, W
        cps, strips, lo, end = geom
        owned = []
        for s in range(strips):
            assert lo(s) % 4 == 0 and end(s) % 4 == 0 and lo(s) < end(s) <= W and end(s) >= lo(s + 1)
            for ox in range(min(s * cps, 64), min((s + 1) * cps, 64)):                 # its pool columns lie inside its span
                assert lo(s) <= win_lo(ox, W) and win_hi(ox, W) <= end(s)
            owned += list(range(lo(s), lo(s + 1)))
        assert owned == list(range(W)), W


def test_libadayolo_exports_header_symbols():
    from adaptiveisp_amd.yolo import _lib
    L = _lib.load()
    names = _declared("adayolo.h")
    assert set(_lib.EXPORTS) <= set(names)
    for n in names:
        assert hasattr(L, n), f"libadayolo.so does not export {n}"
    assert L.adayolo_abi_version() == _lib.ABI_VERSION
    assert not hasattr(L, "adayolo_debug_stamps")            # measurement builds are not in the shipped library


def test_adayolo_argument_checks_without_gpu():
    """Rejected before anything touches a device: null pointers, bad channel counts, kernel numbers that are not in the
    library (only the variants the tuning table may name exist: include/adayolo.h)."""
    from adaptiveisp_amd.yolo import _lib
    L = _lib.load()
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    conv = L.adayolo_conv_fwd_variant
    assert conv(None, 8, p, p, None, 0, p, 8, 1, 4, 4, 8, 8, 3, 1, 1, 0, None) == -1
    assert conv(p, 8, p, p, None, 0, p, 8, 1, 4, 4, 7, 8, 3, 1, 1, 0, None) == -2            # Cin % 8
    assert conv(p, 8, p, p, None, 0, p, 8, 1, 4, 4, 8, 8, 5, 1, 1, 0, None) == -2            # ksize 5
    for bogus in (1, 3, 13, 33, 41, 57, 66, 99):
        assert conv(p, 8, p, p, None, 0, p, 8, 1, 4, 4, 8, 8, 3, 1, 1, bogus, None) == -1, bogus
    assert L.adayolo_letterbox_pack(None, p, 8, 1, 4, 4, 4, 0, 0.5, None) == -1
    assert L.adayolo_letterbox_pack(p, p, 6, 1, 4, 4, 4, 0, 0.5, None) == -2


def test_detloss_argument_checks_without_gpu():
    """adayolo_detloss_fwd / _bwd (include/adayolo.h) refuse malformed argument blocks before any launch, and the
    ctypes mirror of the structs has the header's layout (88 / 424 bytes on LP64)."""
    from adaptiveisp_amd.yolo import _lib
    L = _lib.load()
    assert ctypes.sizeof(_lib.LossLayer) == 88 and ctypes.sizeof(_lib.LossArgs) == 424
    assert L.adayolo_detloss_fwd(None, None) == -1
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p).value
    a = _lib.LossArgs()
    a.nl, a.B, a.na, a.nc, a.no, a.loss, a.ticket = 1, 1, 3, 80, 85, p, p
    lay = a.layer[0]
    lay.raw, lay.cs, lay.ny, lay.nx, lay.tobj, lay.cnt, lay.part, lay.n = p, 256, 2, 2, p, p, p, 0
    lay.cs = 248                                               # narrower than na * no
    assert L.adayolo_detloss_fwd(ctypes.byref(a), None) == -2
    lay.cs = 256
    a.no = 84                                                  # no != nc + 5
    assert L.adayolo_detloss_fwd(ctypes.byref(a), None) == -1
    a.no = 85
    lay.n = 3                                                  # matches announced, no arrays
    assert L.adayolo_detloss_fwd(ctypes.byref(a), None) == -1
    lay.n = 0
    a.nc, a.no = 200, 205                                      # more than two classes per lane
    assert L.adayolo_detloss_fwd(ctypes.byref(a), None) == -2
    a.nc, a.no = 80, 85
    assert L.adayolo_detloss_bwd(ctypes.byref(a), None) == -1  # no upstream gradient
    a.grad_loss = p
    assert L.adayolo_detloss_bwd(ctypes.byref(a), None) == -2  # no gradient map
    lay.grad, lay.grad_cs = p, 252                             # not a multiple of 8
    assert L.adayolo_detloss_bwd(ctypes.byref(a), None) == -2
