#!/usr/bin/env python3
"""Per-workgroup timeline of the ping-pong conv kernel (variant 57 = stamped build; needs a library built with ADAYOLO_EXTRA_FLAGS=-DADAYOLO_MEASURE python -m adaptiveisp_amd.build --force): medians of the stage durations."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptiveisp_amd.yolo import _lib
L = _lib.load()
L.adayolo_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
B = 8
for (H, W, cin, cout, k, s) in [(92, 160, 128, 256, 3, 1), (46, 80, 256, 512, 3, 1), (23, 40, 512, 1024, 3, 1)]:
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(B, H, W, cin, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(cout, k, k, cin, generator=g) / (k * k * cin) ** 0.5).to(torch.bfloat16).cuda()
    b = torch.randn(cout, generator=g).cuda()
    out = torch.zeros(B, H, W, cout, dtype=torch.bfloat16, device="cuda")
    args = (ctypes.c_void_p(x.data_ptr()), cin, ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(b.data_ptr()), None, 0,
            ctypes.c_void_p(out.data_ptr()), cout, B, H, W, cin, cout, k, s, 1, 57)
    for _ in range(3):
        L.adayolo_conv_fwd_variant(*args, _lib.stream_ptr())
    torch.cuda.synchronize()
    nwg = ((B * H * W + 255) // 256) * (cout // 256)
    n = min(nwg, 4096)
    buf = np.zeros(n * 8, np.uint64)
    assert L.adayolo_debug_stamps(buf.ctypes.data, n * 8) == 0
    t = buf.reshape(n, 8).astype(np.float64)
    t0 = t[:, 0].min()
    names = ["weight rows + kernel args", "row decode + prologue DMA + wait", "k-loop", "tail + epilogue (math, residual, store issue)", "store drain"]
    t = t[:, [0, 1, 2, 3, 6, 7]]
    d = np.diff(t, axis=1)
    print(f"{H}x{W} {cin}->{cout}: {nwg} workgroups, kernel span {(t[:, 5].max() - t0):.0f} ticks; first-round start spread {np.percentile(t[:, 0] - t0, 50):.0f} (median)")
    for i, nm in enumerate(names):
        print(f"   {nm:28s} median {np.median(d[:, i]):8.0f}  p90 {np.percentile(d[:, i], 90):8.0f} ticks")
    print(f"   whole workgroup             median {np.median(t[:, 5] - t[:, 0]):8.0f} ticks; start times (ticks since first): p50 {np.percentile(t[:,0]-t0,50):.0f} p99 {np.percentile(t[:,0]-t0,99):.0f}")
