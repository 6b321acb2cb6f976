#!/usr/bin/env python3
"""A/B of the default NLM kernel's tile height inside one process (interleaved rounds): 24 rows (3 workgroups per CU)
vs 32 rows (2 per CU). usage: nlm_ab.py [B,H,W]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptiveisp_amd import _lib  # noqa: E402

B, H, W = map(int, (sys.argv[1] if len(sys.argv) > 1 else "8,720,1280").split(","))
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(1234)
x = (torch.rand(B, 3, H, W, generator=g) ** 2.2 * 0.5).to(dev)
h = torch.full((B, 1), 0.2, device=dev)
out = torch.empty_like(x)
res = {"24 rows": [], "32 rows": []}
for rnd in range(5):
    for name, t32 in (("24 rows", False), ("32 rows", True)):
        _lib.process(4, x, h, clip=True, out=out, nlm_tile32=t32)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            _lib.process(4, x, h, clip=True, out=out, nlm_tile32=t32)
        e1.record()
        torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / 5)
a = _lib.process(4, x, h, clip=True)
b = _lib.process(4, x, h, clip=True, nlm_tile32=True)
for name, v in res.items():
    v = sorted(v)
    tf = 4.6e3 * B * H * W / (v[len(v) // 2] * 1e-3) / 1e12
    print(f"{name}: min {v[0]:.4f} median {v[len(v) // 2]:.4f} ms  ({tf:.1f} TFLOP/s in reference arithmetic = {tf / 157.3:.3f} of fp32 VALU peak)")
print("bit-identical outputs:", bool(torch.equal(a, b)))
