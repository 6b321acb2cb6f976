#!/usr/bin/env python3
"""Phase cycles of the whole-Bottleneck kernel of the shallow stages (csrc/yolo_bneck_ws.hip, -DADAYOLO_MEASURE build: workgroup
0's thread 0 sums s_memtime deltas per phase over its tiles). usage (GPU box):
  python tools/build_variant.py measure yolo -DADAYOLO_MEASURE && ADAYOLO_LIB=build/variants/measure/libadayolo.so python tools/bneck_ws_stamps.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptiveisp_amd.yolo import _lib  # noqa: E402

L = _lib.load()
NAMES = ["wait for the patch", "barrier A", "stage B (1x1 on the patch -> h)", "barrier B", "next patch's DMA issue", "stage C (3x3, 72 / 36 MFMAs)",
         "residual request + SiLU + barrier C", "rows + residual add + stores"]
for C, (B, H, W) in ((128, (8, 184, 320)), (64, (8, 368, 640))):
    g = torch.Generator(device="cpu").manual_seed(C)
    x = torch.randn(B, H, W, C, generator=g).to(torch.bfloat16).cuda()
    w1 = (torch.randn(C // 2, C, generator=g) / C ** 0.5).to(torch.bfloat16).cuda()
    b1 = torch.randn(C // 2, generator=g).cuda()
    w2 = (torch.randn(C, 3, 3, C // 2, generator=g) / (9 * C // 2) ** 0.5).to(torch.bfloat16).cuda()
    b2 = torch.randn(C, generator=g).cuda()
    out = torch.empty_like(x)
    P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    for _ in range(3):
        assert L.adayolo_bottleneck_ws_fwd(P(x), C, P(w1), P(b1), P(w2), P(b2), P(out), C, B, H, W, C, _lib.stream_ptr()) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        L.adayolo_bottleneck_ws_fwd(P(x), C, P(w1), P(b1), P(w2), P(b2), P(out), C, B, H, W, C, _lib.stream_ptr())
    e1.record()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    L.adayolo_debug_bws.argtypes = [ctypes.c_void_p]
    assert L.adayolo_debug_bws(buf) == 0
    n = max(1, buf[8])
    tot = sum(buf[i] for i in range(8))
    print(f"C = {C} @ {B}x{H}x{W}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per launch; workgroup 0: {n} tiles, {tot / n:.0f} cycles per tile")
    for i in range(8):
        print(f"   {buf[i] / n:8.0f}  {NAMES[i]}")
