#!/usr/bin/env python3
"""Interleaved A/B of the ISP step launches between builds of libadaisp.so in one process (boxes differ by more than most
kernel changes): per op the plain launch (adaisp_process) and the RL step's launch (adaisp_forward_uniform with the next
step's 64x64 pooling), on buffer sets ROTATING over more than the 256 MB Infinity Cache. Outputs and pooled planes of every
build are compared bit for bit with the first one's.
usage: isp_step_ab.py name=path/to/libadaisp.so [name=path ...] [--ops 0,2,5,3] [--shape 8,720,1280] [--pairs 6]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptiveisp_amd import _lib

args = [a for a in sys.argv[1:] if not a.startswith("--")]
opt = {a.split("=")[0][2:]: a.split("=")[1] for a in sys.argv[1:] if a.startswith("--") and "=" in a}
ops = [int(v) for v in opt.get("ops", "0,2,5,3").split(",")]
B, H, W = map(int, opt.get("shape", "8,720,1280").split(","))
pairs = int(opt.get("pairs", "6"))
libs = {"in-tree": _lib.load()}
for a in args:
    n, p = a.split("=")
    libs[n] = ctypes.CDLL(os.path.abspath(p))
vp, ci, cu = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint
for L in libs.values():
    L.adaisp_process.argtypes = [ci, vp, vp, vp, ci, ci, ci, ci, cu, vp]
    L.adaisp_process.restype = ci
    L.adaisp_forward_uniform.argtypes = [ci, vp, vp, vp, vp, ci, ci, ci, ci, cu, vp]
    L.adaisp_forward_uniform.restype = ci
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(1234)
x = (torch.rand(B, 3, H, W, generator=g) ** 2.2 * 0.5).to(dev)
ins = [x] + [x.clone() for _ in range(pairs - 1)]
outs = [torch.empty_like(x) for _ in range(pairs)]
pools = [torch.empty(B, 3, 64, 64, device=dev) for _ in range(pairs)]
p = (torch.rand(B, 24, generator=g) * 0.8 + 0.6).to(dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
NAMES = {0: "E", 1: "G", 2: "CCM", 3: "Shr", 4: "NLM", 5: "T", 6: "Ct", 7: "S+", 8: "BW", 9: "W", 10: "USM", 11: "ShrV2", 12: "C"}
print(f"# {B}x3x{H}x{W}, {pairs} buffer pairs = {pairs * 2 * x.numel() * 4 / 1e6:.0f} MB per cycle; us per launch, median of 7 rounds x 12 launches (min)")


def run(L, op, mode, n):
    for i in range(n):
        k = i % pairs
        if mode == "plain":
            rc = L.adaisp_process(op, ins[k].data_ptr(), outs[k].data_ptr(), p.data_ptr(), 24, B, H, W, 1, st)
        else:
            rc = L.adaisp_forward_uniform(op, ins[k].data_ptr(), outs[k].data_ptr(), pools[k].data_ptr(), p.data_ptr(), 24, B, H, W, 1, st)
        assert rc == 0, rc


for op in ops:
    for mode in ("plain", "step+pool"):
        res = {n: [] for n in libs}
        for rnd in range(7):
            for n, L in libs.items():
                run(L, op, mode, pairs)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(L, op, mode, 12); e1.record()
                torch.cuda.synchronize()
                res[n].append(e0.elapsed_time(e1) / 12 * 1e3)
        ref = None
        same = {}
        for n, L in libs.items():
            outs[0].zero_(); pools[0].zero_()
            run(L, op, mode, 1)
            torch.cuda.synchronize()
            cur = (outs[0].clone(), pools[0].clone())
            if ref is None:
                ref = cur
            same[n] = bool(torch.equal(cur[0], ref[0]) and (mode == "plain" or torch.equal(cur[1], ref[1])))
        px = B * H * W
        print(f"{NAMES.get(op, op):5s} {mode:9s} " + "  ".join(
            f"{n} {sorted(v)[len(v) // 2]:6.1f} ({min(v):5.1f}) {24.0 * px / sorted(v)[len(v) // 2] / 1e6:4.2f} TB/s{'' if same[n] else ' DIFFERS'}"
            for n, v in res.items()))
