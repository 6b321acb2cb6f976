#!/usr/bin/env python3
"""Ablation of the conv k-loop on the dominant shapes: full kernel vs compute-only vs DMA-only (variants 5/7/8)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptiveisp_amd.yolo import _lib
L = _lib.load()
B = 8
SHAPES = [tuple(int(v) for v in t.split(',')) for t in sys.argv[2].split(';')] if len(sys.argv) > 2 else None
for (H, W, cin, cout, k, s) in SHAPES or [(92, 160, 128, 256, 3, 1), (46, 80, 256, 512, 3, 1), (92, 160, 256, 128, 1, 1), (23, 40, 512, 1024, 3, 1), (184, 320, 64, 128, 3, 1), (368, 640, 32, 64, 3, 1), (736, 1280, 32, 64, 3, 2), (368, 640, 64, 128, 3, 2)]:
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(B, H, W, cin, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(cout, k, k, cin, generator=g) / (k * k * cin) ** 0.5).to(torch.bfloat16).cuda()
    b = torch.randn(cout, generator=g).cuda()
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    out = torch.zeros(B, Ho, Wo, cout, dtype=torch.bfloat16, device="cuda")
    fl = 2.0 * B * Ho * Wo * cout * k * k * cin
    row, ref = [], None
    for v in [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "2,5,7,8").split(",")]:
        args = (ctypes.c_void_p(x.data_ptr()), cin, ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(b.data_ptr()), None, 0,
                ctypes.c_void_p(out.data_ptr()), cout, B, H, W, cin, cout, k, s, 1, v)
        st = _lib.stream_ptr()
        for _ in range(3):
            L.adayolo_conv_fwd_variant(*args, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            L.adayolo_conv_fwd_variant(*args, st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        if ref is None:
            ref = out.clone()
        d = (out.float() - ref.float()).abs().max().item()
        row.append(f"v{v}: {ms*1e3:6.1f} us {fl/ms/1e9:6.1f} TF d={d:.2g}")
    print(f"{H}x{W} {cin}->{cout} k{k}: " + "   ".join(row))
