#!/usr/bin/env python3
"""Per-op kernel timings of the ISP stack at a BASELINE config (HIP events on the launch stream).
Prints algorithmic GB/s (24 B/px) and the fraction of the 8 TB/s HBM spec."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptiveisp_amd import _lib  # noqa: E402

OPS = {"E": 0, "G": 1, "CCM": 2, "Shr": 3, "NLM": 4, "T": 5, "Ct": 6, "S+": 7, "BW": 8, "W": 9, "USM": 10,
       "ShrV2": 11, "C": 12}
NPAR = {0: 1, 1: 1, 2: 9, 3: 1, 4: 1, 5: 8, 6: 1, 7: 1, 8: 1, 9: 3, 10: 2, 11: 1, 12: 24}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="8,720,1280")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--ops", default=",".join(OPS))
    a = ap.parse_args()
    B, H, W = map(int, a.shape.split(","))
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(1234)
    x = (torch.rand(B, 3, H, W, generator=g) ** 2.2 * 0.5).to(dev)
    out = torch.empty_like(x)
    px = B * H * W
    res = {}
    for name in a.ops.split(","):
        op = OPS[name]
        p = torch.rand(B, NPAR[op], device=dev) * 0.8 + 0.6
        iters = max(2, a.iters // 5) if name == "NLM" else a.iters
        for _ in range(2):
            _lib.process(op, x, p, clip=True, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            _lib.process(op, x, p, clip=True, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        gbs = 24.0 * px / (ms * 1e-3) / 1e9
        res[name] = dict(ms=round(ms, 4), GBps=round(gbs, 1), frac_hbm=round(gbs / 8000.0, 3))
        print(f"{name:6s} {ms:9.4f} ms  {gbs:9.1f} GB/s  {gbs / 80:.1f}% of 8 TB/s", flush=True)
    # pool + mixed-ids forward
    for _ in range(2):
        _lib.pool64(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        _lib.pool64(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    res["pool64"] = dict(ms=round(ms, 4), GBps=round(12.0 * px / (ms * 1e-3) / 1e9, 1))
    print(f"pool64 {ms:9.4f} ms  {12.0 * px / (ms * 1e-3) / 1e9:9.1f} GB/s (12 B/px)")
    ids = torch.tensor([0, 9, 2, 5, 1, 6, 7, 8][:B] * (B // 8 + 1), dtype=torch.int32, device=dev)[:B]
    pp = torch.rand(B, 24, device=dev) * 0.8 + 0.6
    pooled = torch.empty(B, 3, 64, 64, device=dev)
    for _ in range(2):
        _lib.forward(x, ids, pp, clip=True, pooled=pooled, out=out)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(a.iters):
        _lib.forward(x, ids, pp, clip=True, pooled=pooled, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    res["forward_mixed_pointwise+pool"] = dict(ms=round(ms, 4))
    print(f"adaisp_forward (8 pointwise ops mixed, + empty stencil/NLM grids + pool): {ms:.4f} ms")
    print(json.dumps({"shape": [B, H, W], "results": res}))


if __name__ == "__main__":
    main()
