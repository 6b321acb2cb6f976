#!/usr/bin/env python3
"""Where is the streaming ceiling on the tensors the stencils run on? (sharpen at 4K reads 0.575 of the
8 TB/s spec — is that the kernel or the memory system?) Same tensors (4 x 3 x 2160 x 3840 fp32, 398 MB each; 3 rotating
input / output pairs = 2.4 GB, nothing stays in the 256 MB Infinity Cache), 24 B/px for every row:
  torch copy_          the vendor's D2D copy kernel
  pointwise identity   k_pointwise (white balance with unit gains): 3 planes in, 3 planes out, no arithmetic to speak of
  sharpen 3x3 / USM 5x5 / fused sharpen + 64x64 pooling
Prints ms, TB/s and the fraction of 8 TB/s; also at config 2's size. usage: python3 tools/copy_ceiling.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptiveisp_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")


def bench(fn, n=12):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for B, H, W in ((4, 2160, 3840), (8, 720, 1280)):
    g = torch.Generator(device="cpu").manual_seed(7)
    x = (torch.rand(B, 3, H, W, generator=g) ** 2.2 * 0.5).to(dev)
    ins = [x, x.clone(), x.clone()]
    outs = [torch.empty_like(x) for _ in range(3)]
    pools = [torch.empty(B, 3, 64, 64, device=dev) for _ in range(3)]
    one3 = torch.ones(B, 3, device=dev)
    f = torch.full((B, 24), 2.5, device=dev)
    usm = torch.tensor([[1.2, 0.8]] * B, device=dev)
    px = B * H * W
    rows = [("torch copy_ (3 planes)", lambda i: outs[i % 3].copy_(ins[i % 3])),
            ("pointwise identity (WB, unit gains)", lambda i: _lib.process(_lib.OP_WB, ins[i % 3], one3, out=outs[i % 3])),
            ("sharpen 3x3", lambda i: _lib.process(_lib.OP_SHARPEN, ins[i % 3], f[:, :1], clip=True, out=outs[i % 3])),
            ("unsharp mask 5x5", lambda i: _lib.process(_lib.OP_USM, ins[i % 3], usm, clip=True, out=outs[i % 3])),
            ("sharpen 3x3 + fused 64x64 pooling", lambda i: _lib.forward(ins[i % 3], None, f, clip=True, out=outs[i % 3],
                                                                      pooled=pools[i % 3], host_op=_lib.OP_SHARPEN)),
            ("exposure + fused 64x64 pooling", lambda i: _lib.forward(ins[i % 3], None, f * 0.1, clip=True, out=outs[i % 3],
                                                                   pooled=pools[i % 3], host_op=_lib.OP_EXPOSURE)),
            ("pool64 alone (12 B/px)", lambda i: _lib.pool64(ins[i % 3]))]
    print(f"# {B} x 3 x {H} x {W} fp32, rotating over 3 buffer pairs ({6 * x.numel() * 4 / 1e6:.0f} MB)")
    for name, fn in rows:
        ms = bench(fn)
        bpp = 12 if "alone" in name else 24
        tbs = bpp * px / (ms * 1e-3) / 1e12
        print(f"{name:40s} {ms * 1e3:8.1f} us  {tbs:5.2f} TB/s  {tbs / 8.0:5.3f} of 8 TB/s")
