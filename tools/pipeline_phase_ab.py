#!/usr/bin/env python3
"""Interleaved A/B of the two-stream pipeline's phase: where the ISP episode is cut against the detector's layers and
how many of its half-steps run before the detector is released (bench.build_pipeline(cut=, gate=)).
usage: pipeline_phase_ab.py [cut:gate,... = 0:0,4:0,5:0,5:1] [steps=40] [rounds=3]"""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

splits = [tuple(int(u) for u in v.split(":")) for v in (sys.argv[1] if len(sys.argv) > 1 else "0:0,4:0,5:0,5:1").split(",")]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
a = argparse.Namespace(batch=8, height=720, width=1280, schedule=os.environ.get("SCHEDULE", "mixed"), retune=False)
step, engine, agent, x0, sched = bench.build_workload(a, torch.device("cuda:0"))
ref = step().clone()
torch.cuda.synchronize()
pipes = {}
for s in splits:
    prime, run = bench.build_pipeline(step, engine, x0, cut=s[0], gate=s[1])
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
    print(f"cut {s[0]} gate {s[1]}: "
          + "  ".join(f"{t:.3f}" for t in res[s]) + f"  ms/step  -> best {8 / min(res[s]) * 1e3:.0f} images/s")
