#!/usr/bin/env python3
"""Interleaved A/B of adaisp_pool64 between two builds of libadaisp.so in one process. usage: pool_ab.py <other .so> [B,H,W]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptiveisp_amd import _lib
other = sys.argv[1]
B, H, W = map(int, (sys.argv[2] if len(sys.argv) > 2 else "8,720,1280").split(","))
libs = {"in-tree": _lib.load(), "other": ctypes.CDLL(os.path.abspath(other))}
vp, ci = ctypes.c_void_p, ctypes.c_int
for L in libs.values():
    L.adaisp_pool64.argtypes = [vp, vp, ci, ci, ci, vp]
    L.adaisp_pool64.restype = ci
g = torch.Generator(device="cpu").manual_seed(3)
x = torch.rand(B, 3, H, W, generator=g).cuda()
x2 = torch.rand(B, 3, H, W, generator=g).cuda()          # a second image: alternate so that the reads miss the caches
outs = {k: torch.empty(B, 3, 64, 64, device="cuda") for k in libs}
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
res = {k: [] for k in libs}
for rnd in range(8):
    for name, L in libs.items():
        for _ in range(2):
            L.adaisp_pool64(x.data_ptr(), outs[name].data_ptr(), B, H, W, st); L.adaisp_pool64(x2.data_ptr(), outs[name].data_ptr(), B, H, W, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            L.adaisp_pool64(x.data_ptr(), outs[name].data_ptr(), B, H, W, st); L.adaisp_pool64(x2.data_ptr(), outs[name].data_ptr(), B, H, W, st)
        e1.record(); torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / 20 * 1e3)
for name, v in res.items():
    v = sorted(v)
    print(f"{name:8s}: min {v[0]:.2f} median {v[len(v) // 2]:.2f} us  ({B * 3 * H * W * 4 / v[len(v) // 2] / 1e6:.2f} TB/s)")
libs["in-tree"].adaisp_pool64(x.data_ptr(), outs["in-tree"].data_ptr(), B, H, W, st)
libs["other"].adaisp_pool64(x.data_ptr(), outs["other"].data_ptr(), B, H, W, st)
torch.cuda.synchronize()
print("bit-identical outputs:", bool(torch.equal(outs["in-tree"], outs["other"])))
