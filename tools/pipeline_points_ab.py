#!/usr/bin/env python3
"""Where the five filter launches of the next batch's ISP episode sit between the detector's layers (bench.build_interleaved
points=) and the priority of the policy stream, interleaved in one process with the two-stream arrangement of rounds 2-3.
usage: pipeline_points_ab.py [steps=40] [rounds=3]"""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a = argparse.Namespace(batch=8, height=720, width=1280, schedule="mixed", retune=False)
step, engine, agent, x0, sched = bench.build_workload(a, torch.device("cuda:0"))
step(); torch.cuda.synchronize()
L = engine.num_launches()
print("detector launches per forward:", L)
cands = {"streams (round 3)": None}
for name, pts, prio in [("default", None, 0), ("default hi-prio policy", None, -1),
                        ("spread 4..64", [4, 19, 34, 49, 64], 0), ("spread 4..64 hi", [4, 19, 34, 49, 64], -1),
                        ("spread 2..58 hi", [2, 16, 30, 44, 58], -1), ("late NLM hi", [2, 10, 38, 52, 66], -1),
                        ("early hi", [1, 8, 16, 30, 44], -1), ("wide hi", [3, 18, 30, 48, 63], -1), ("2..66 hi", [2, 18, 34, 50, 66], -1)]:
    cands[name] = (pts, prio)
runs = {}
for name, spec in cands.items():
    prime, run = bench.build_pipeline(step, engine, x0) if spec is None else bench.build_interleaved(step, engine, x0, points=spec[0], side_priority=spec[1])
    prime(); run(); run(); torch.cuda.synchronize()
    runs[name] = run
res = {k: [] for k in runs}
for r in range(rounds):
    for name, run in runs.items():
        for _ in range(4):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / steps * 1e3)
for name, v in res.items():
    v = sorted(v)
    print(f"{name:28s} {v[len(v) // 2]:.3f} ms per step (min {v[0]:.3f})  {8 / v[len(v) // 2] * 1e3:7.1f} images/s   points {getattr(runs[name], 'points', '-')}")
