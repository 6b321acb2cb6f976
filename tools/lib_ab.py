#!/usr/bin/env python3
"""Interleaved A/B of one ISP op between two builds of libadaisp.so inside one process (boxes differ by more than most
kernel changes). usage: lib_ab.py <other libadaisp.so> [op=4 (NLM)] [B,H,W=8,720,1280]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptiveisp_amd import _lib

other = sys.argv[1]
op = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B, H, W = map(int, (sys.argv[3] if len(sys.argv) > 3 else "8,720,1280").split(","))
libs = {"in-tree": _lib.load(), "other": ctypes.CDLL(os.path.abspath(other))}
vp, ci = ctypes.c_void_p, ctypes.c_int
for L in libs.values():
    L.adaisp_process.argtypes = [ci, vp, vp, vp, ci, ci, ci, ci, ctypes.c_uint, vp]
    L.adaisp_process.restype = ci
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(1234)
x = (torch.rand(B, 3, H, W, generator=g) ** 2.2 * 0.5).to(dev)
p = torch.full((B, 16), 0.2, device=dev)
outs = {k: torch.empty_like(x) for k in libs}
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(name, n):
    for _ in range(n):
        rc = libs[name].adaisp_process(op, x.data_ptr(), outs[name].data_ptr(), p.data_ptr(), 16, B, H, W, 1, st)
        assert rc == 0, rc


res = {k: [] for k in libs}
for rnd in range(6):
    for name in libs:
        run(name, 2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(name, 5); e1.record()
        torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / 5)
for name, v in res.items():
    v = sorted(v)
    print(f"{name:8s}: min {v[0]:.4f} median {v[len(v) // 2]:.4f} ms")
print("bit-identical outputs:", bool(torch.equal(outs["in-tree"], outs["other"])))
