#!/usr/bin/env python3
"""Why is a conv launch slower inside the network than when the same layer is repeated? (clock, or cold
operand lines?) The same launch back to back, 120 times, (a) on ONE set of buffers — input, weights, residual, output stay
in the Infinity Cache / L2 between launches — and (b) rotating over N sets (> 256 MB in total: every launch reads lines that
have left the caches, as in the network where a layer's weights are touched once per forward and its input is the
stream of the previous layer). Clock conditions are identical (both sustained). usage: python3 tools/conv_cold_warm.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptiveisp_amd.yolo import _lib

L = _lib.load()
B = 8
g = torch.Generator(device="cpu").manual_seed(0)
for (H, W, cin, cout, k, s, v, res) in [(92, 160, 128, 256, 3, 1, 50, True), (46, 80, 256, 512, 3, 1, 50, True),
                                        (23, 40, 512, 1024, 3, 1, 60, True), (46, 80, 512, 256, 1, 1, 60, False),
                                        (92, 160, 256, 128, 1, 1, 80, False)]:
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    per_set = (B * H * W * cin + cout * k * k * cin + (2 if res else 1) * B * Ho * Wo * cout) * 2
    nsets = max(2, int(600e6 // per_set) + 1)
    sets = []
    for _ in range(nsets):
        x = torch.randn(B, H, W, cin, generator=g).to(torch.bfloat16).cuda()
        w = (torch.randn(cout, k, k, cin, generator=g) / (k * k * cin) ** 0.5).to(torch.bfloat16).cuda()
        b = torch.randn(cout, generator=g).cuda()
        r = torch.randn(B, Ho, Wo, cout, generator=g).to(torch.bfloat16).cuda() if res else None
        o = torch.zeros(B, Ho, Wo, cout, dtype=torch.bfloat16, device="cuda")
        sets.append((x, w, b, r, o))

    def launch(i):
        x, w, b, r, o = sets[i]
        L.adayolo_conv_fwd_variant(ctypes.c_void_p(x.data_ptr()), cin, ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(b.data_ptr()),
                                   ctypes.c_void_p(r.data_ptr()) if r is not None else None, cout if r is not None else 0,
                                   ctypes.c_void_p(o.data_ptr()), cout, B, H, W, cin, cout, k, s, 1, v, _lib.stream_ptr())

    def timed(rotate, n=120):
        for i in range(10):
            launch(i % nsets if rotate else 0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            launch(i % nsets if rotate else 0)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    warm, cold = timed(False), timed(True)
    fl = 2.0 * B * Ho * Wo * cout * k * k * cin
    print(f"{cin}->{cout} k{k} @{H}x{W} variant {v}{' +res' if res else ''}: one buffer set {warm:6.1f} us ({fl / warm / 1e6:6.0f} TFLOP/s) | "
          f"rotating {nsets} sets ({nsets * per_set / 1e6:.0f} MB) {cold:6.1f} us ({fl / cold / 1e6:6.0f} TFLOP/s) | cold / warm {cold / warm:.3f}", flush=True)
