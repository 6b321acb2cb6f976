#!/usr/bin/env python3
"""One conv shape, N launches (for rocprofv3 --pmc passes). usage: conv_one.py H W Cin Cout k s [variant] [iters]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptiveisp_amd.yolo import _lib
H, W, cin, cout, k, s = map(int, sys.argv[1:7])
variant = int(sys.argv[7]) if len(sys.argv) > 7 else 0
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 5
B = 8
L = _lib.load()
g = torch.Generator(device="cpu").manual_seed(0)
x = torch.randn(B, H, W, cin, generator=g).to(torch.bfloat16).cuda()
w = (torch.randn(cout, k, k, cin, generator=g) / (k * k * cin) ** 0.5).to(torch.bfloat16).cuda()
b = torch.randn(cout, generator=g).cuda()
Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
out = torch.zeros(B, Ho, Wo, cout, dtype=torch.bfloat16, device="cuda")
args = (ctypes.c_void_p(x.data_ptr()), cin, ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(b.data_ptr()), None, 0,
        ctypes.c_void_p(out.data_ptr()), cout, B, H, W, cin, cout, k, s, 1, variant)
for _ in range(iters):
    L.adayolo_conv_fwd_variant(*args, _lib.stream_ptr())
torch.cuda.synchronize()
print("done", 2.0 * B * Ho * Wo * cout * k * k * cin / 1e9, "GFLOP per launch")
