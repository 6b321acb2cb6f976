#!/usr/bin/env python3
"""Can NLM (fp32 VALU-bound, 48.5 KB LDS, ~100 VGPRs) run INSIDE the detector's conv launches instead of beside them?
The 256x256 ping-pong kernel owns a CU (148 KB LDS, 2 x 246 VGPRs per SIMD): the two streams time-slice CUs and the step
is the sum of CU time. The two-workgroup kernel (pq: 4 waves x 256 VGPRs, 72 KB LDS) leaves half the register file and
88 KB of LDS: NLM workgroups could be co-resident and use the VALU slots its MFMA waves leave. Measured: N launches of one
conv layer on one stream, one NLM launch on another, each alone and both together, per conv variant."""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptiveisp_amd import _lib as isp
from adaptiveisp_amd.yolo import _lib

L = _lib.load()
B = 8
g = torch.Generator(device="cpu").manual_seed(0)
img = (torch.rand(B, 3, 720, 1280, generator=g) ** 2.2 * 0.5).cuda()
nout = torch.empty_like(img)
h = torch.full((B, 1), 0.2, device="cuda")
s_conv, s_nlm = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for (H, W, cin, cout, k, s, n) in [(46, 80, 256, 512, 3, 1, 8), (92, 160, 128, 256, 3, 1, 8), (23, 40, 512, 1024, 3, 1, 8)]:
    x = torch.randn(B, H, W, cin, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(cout, k, k, cin, generator=g) / (k * k * cin) ** 0.5).to(torch.bfloat16).cuda()
    b = torch.randn(cout, generator=g).cuda()
    out = torch.zeros(B, H, W, cout, dtype=torch.bfloat16, device="cuda")
    for v in (50, 60, 80):
        args = (ctypes.c_void_p(x.data_ptr()), cin, ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(b.data_ptr()), None, 0,
                ctypes.c_void_p(out.data_ptr()), cout, B, H, W, cin, cout, k, s, 1, v)

        def convs():
            with torch.cuda.stream(s_conv):
                for _ in range(n):
                    L.adayolo_conv_fwd_variant(*args, ctypes.c_void_p(s_conv.cuda_stream))

        def nlm():
            with torch.cuda.stream(s_nlm):
                isp.process(isp.OP_NLM, img, h, clip=True, out=nout)

        def both():
            convs(); nlm()

        tc, tn, tb = timed(convs), timed(nlm), timed(both)
        print(f"{cin}->{cout} k{k} @{H}x{W} variant {v}: {n} convs {tc:.3f} ms | NLM {tn:.3f} ms | together {tb:.3f} ms "
              f"(sum {tc + tn:.3f}, hidden {tc + tn - tb:+.3f})", flush=True)
