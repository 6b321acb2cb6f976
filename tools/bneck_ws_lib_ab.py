#!/usr/bin/env python3
"""The whole-Bottleneck kernel of the shallow stages between builds of libadayolo.so, interleaved in one process (measurement builds:
tools/build_variant.py <name> yolo -D...). usage (GPU box): python tools/bneck_ws_lib_ab.py name=path/to/libadayolo.so [...]"""
import ctypes
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptiveisp_amd.yolo import _lib  # noqa: E402

libs = {"in-tree": _lib.load()}
for a in sys.argv[1:]:
    n, p = a.split("=")
    libs[n] = ctypes.CDLL(os.path.abspath(p))
vp, ci = ctypes.c_void_p, ctypes.c_int
for L in libs.values():
    L.adayolo_bottleneck_ws_fwd.argtypes = [vp, ci, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]
    L.adayolo_bottleneck_ws_fwd.restype = ci
for C, (B, H, W) in ((128, (8, 184, 320)), (64, (8, 368, 640))):
    g = torch.Generator(device="cpu").manual_seed(C)
    x = torch.randn(B, H, W, C, generator=g).to(torch.bfloat16).cuda()
    w1 = (torch.randn(C // 2, C, generator=g) / C ** 0.5).to(torch.bfloat16).cuda()
    b1 = torch.randn(C // 2, generator=g).cuda()
    w2 = (torch.randn(C, 3, 3, C // 2, generator=g) / (9 * C // 2) ** 0.5).to(torch.bfloat16).cuda()
    b2 = torch.randn(C, generator=g).cuda()
    outs = {n: torch.empty_like(x) for n in libs}
    st = _lib.stream_ptr()
    P = lambda t: vp(t.data_ptr())  # noqa: E731
    res = {n: [] for n in libs}
    for rnd in range(9):
        for n, L in libs.items():
            for _ in range(3):
                assert L.adayolo_bottleneck_ws_fwd(P(x), C, P(w1), P(b1), P(w2), P(b2), P(outs[n]), C, B, H, W, C, st) == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                L.adayolo_bottleneck_ws_fwd(P(x), C, P(w1), P(b1), P(w2), P(b2), P(outs[n]), C, B, H, W, C, st)
            e1.record()
            torch.cuda.synchronize()
            res[n].append(e0.elapsed_time(e1) / 10 * 1e3)
    ref = outs["in-tree"]
    print(f"C = {C} @ {B}x{H}x{W}: " + "  ".join(f"{n} {statistics.median(v):.1f} us (min {min(v):.1f}){'' if torch.equal(outs[n], ref) else ' DIFFERS'}" for n, v in res.items()), flush=True)
