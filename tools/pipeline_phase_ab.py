#!/usr/bin/env python3
"""Interleaved A/B of the two-stream pipeline's phase: where the ISP episode is cut against the detector's layers
(bench.build_pipeline(split=s)). usage: pipeline_phase_ab.py [splits=0,1,2,3] [steps=40] [rounds=3]"""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

splits = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,1,2,3").split(",")]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
a = argparse.Namespace(batch=8, height=720, width=1280, schedule=os.environ.get("SCHEDULE", "mixed"), retune=False)
step, engine, agent, x0, sched = bench.build_workload(a, torch.device("cuda:0"))
ref = step().clone()
torch.cuda.synchronize()
pipes = {}
for s in splits:
    prime, run = bench.build_pipeline(step, engine, x0, split=s)
    prime()
    for _ in range(3):
        run()
        torch.cuda.synchronize()
        assert torch.equal(engine.pred, ref), f"split {s}: pipelined result differs from the sequential step"
    pipes[s] = run
res = {s: [] for s in splits}
for r in range(rounds):
    for s in splits:
        run = pipes[s]
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
        torch.cuda.synchronize()
        res[s].append((time.perf_counter() - t0) / steps * 1e3)
for s in splits:
    print(f"split {s} ({[bench.NAMES[k] for k in sched[s:]]} of batch i+1, then {[bench.NAMES[k] for k in sched[:s]]} of i+2): "
          + "  ".join(f"{t:.3f}" for t in res[s]) + f"  ms/step  -> best {8 / min(res[s]) * 1e3:.0f} images/s")
