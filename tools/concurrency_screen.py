#!/usr/bin/env python3
"""Bit-reproducibility of every forward ISP op while a second stream keeps the chip full of big-LDS workgroups
(detector forward + NLM): 25 runs each, every run must equal the undisturbed result."""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from adaptiveisp_amd import _lib
a = argparse.Namespace(batch=8, height=720, width=1280, schedule="mixed", retune=False)
step, engine, agent, x0, sched = bench.build_workload(a, torch.device("cuda:0"))
xref = step.isp_chain().clone()
side = torch.cuda.Stream()
rng = np.random.default_rng(0)
npar = {0: 1, 1: 1, 2: 9, 3: 1, 4: 1, 5: 8, 6: 1, 7: 1, 8: 1, 9: 3, 10: 2, 11: 1, 12: 24}
names = ["E", "G", "CCM", "Shr", "NLM", "T", "Ct", "S+", "BW", "W", "USM", "ShrV2", "C"]
xs = {"720x1280": x0, "ragged 333x517": x0[:3, :, :333, :517].contiguous()}
bad_total = 0
for tag, x in xs.items():
    B = x.shape[0]
    for op in range(13):
        p = torch.from_numpy((rng.random((B, npar[op])) * 0.8 + 0.6).astype(np.float32)).cuda()
        if op == 4:
            p = p * 0.3
        fn = lambda: _lib.process(op, x, p, clip=True)
        ref = fn().clone(); torch.cuda.synchronize()
        bad = 0
        for i in range(25 if op != 4 else 8):
            with torch.cuda.stream(side), torch.no_grad():
                engine(xref)
                _lib.process(4, x0[:2], torch.full((2, 1), 0.3, device="cuda:0"), clip=True)
            y = fn(); torch.cuda.synchronize()
            bad += not torch.equal(y, ref)
        bad_total += bad
        print(f"{tag:16s} {names[op]:6s}: {bad} differing runs")
    ref = _lib.pool64(x).clone(); torch.cuda.synchronize()
    bad = 0
    for i in range(25):
        with torch.cuda.stream(side), torch.no_grad():
            engine(xref)
        bad += not torch.equal(_lib.pool64(x), ref); torch.cuda.synchronize()
    bad_total += bad
    print(f"{tag:16s} pool64: {bad} differing runs")
print("TOTAL differing runs:", bad_total)
sys.exit(1 if bad_total else 0)
