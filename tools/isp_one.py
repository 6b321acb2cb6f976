#!/usr/bin/env python3
"""One ISP op at config 2, N launches (for rocprofv3 --pmc passes). usage: isp_one.py OP [iters]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptiveisp_amd import _lib
op = int(sys.argv[1]); iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
g = torch.Generator(device="cpu").manual_seed(1234)
x = (torch.rand(8, 3, 720, 1280, generator=g) ** 2.2 * 0.5).cuda()
out = torch.empty_like(x)
p = torch.rand(8, 24, device="cuda") * 0.8 + 0.6
for _ in range(iters):
    _lib.process(op, x, p, clip=True, out=out)
torch.cuda.synchronize()
