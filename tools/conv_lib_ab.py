#!/usr/bin/env python3
"""Interleaved A/B of one conv variant (with its residual, as in a Bottleneck) between two builds of libadayolo.so in one
process. usage: conv_lib_ab.py <other libadayolo.so> <variant> [H,W,Cin,Cout,k,s ...]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptiveisp_amd.yolo import _lib
other, variant = sys.argv[1], int(sys.argv[2])
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[3:]] or [(184, 320, 64, 128, 3, 1), (368, 640, 32, 64, 3, 1)]
libs = {"in-tree": _lib.load(), "other": ctypes.CDLL(os.path.abspath(other))}
vp, ci = ctypes.c_void_p, ctypes.c_int
for L in libs.values():
    L.adayolo_conv_fwd_variant.argtypes = [vp, ci, vp, vp, vp, ci, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, vp]
    L.adayolo_conv_fwd_variant.restype = ci
B = 8
st = _lib.stream_ptr()
for (H, W, cin, cout, k, s) in shapes:
    g = torch.Generator(device="cpu").manual_seed(H + cin)
    x = torch.randn(B, H, W, cin, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(cout, k, k, cin, generator=g) / (k * k * cin) ** 0.5).to(torch.bfloat16).cuda()
    b = torch.randn(cout, generator=g).cuda()
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    res = torch.randn(B, Ho, Wo, cout, generator=g).to(torch.bfloat16).cuda()
    outs = {n: torch.zeros(B, Ho, Wo, cout, dtype=torch.bfloat16, device="cuda") for n in libs}
    def run(n, reps):
        for _ in range(reps):
            rc = libs[n].adayolo_conv_fwd_variant(x.data_ptr(), cin, w.data_ptr(), b.data_ptr(), res.data_ptr(), cout, outs[n].data_ptr(), cout,
                                                  B, H, W, cin, cout, k, s, 1, variant, st)
            assert rc == 0, rc
    t = {n: [] for n in libs}
    for rnd in range(6):
        for n in libs:
            run(n, 2); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(n, 10); e1.record(); torch.cuda.synchronize()
            t[n].append(e0.elapsed_time(e1) / 10 * 1e3)
    same = torch.equal(outs["in-tree"].view(torch.int16), outs["other"].view(torch.int16))
    print(f"{H}x{W} {cin}->{cout} k{k}s{s} v{variant}: " + "  ".join(f"{n} {sorted(v)[len(v) // 2]:.1f} us (min {min(v):.1f})" for n, v in t.items()) + f"  identical {same}")
