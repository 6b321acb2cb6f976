#!/usr/bin/env python3
"""The headline's two-stream pipeline as ONE hipGraph per step with the detector forked onto the second stream inside it, against TWO
one-stream graphs per step launched on the two streams and held in lockstep by events between the launches
(bench.build_pipeline(two_graphs=)). Interleaved in one process; both are checked against the sequential step first.
usage (GPU box): python tools/pipeline_graphs_ab.py [steps=40] [rounds=5]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
a = argparse.Namespace(batch=8, height=720, width=1280, schedule=os.environ.get("SCHEDULE", "mixed"), retune=False)
step, engine, agent, x0, sched = bench.build_workload(a, torch.device("cuda:0"))
ref = step().clone()
torch.cuda.synchronize()
pipes = {}
for two in (False, True):
    prime, run = bench.build_pipeline(step, engine, x0, two_graphs=two)
    prime()
    for _ in range(3):
        run()
        torch.cuda.synchronize()
        assert torch.equal(engine.pred, ref), f"two_graphs={two}: pipelined result differs from the sequential step"
    pipes[two] = run
res = {k: [] for k in pipes}
host = {k: [] for k in pipes}
for r in range(rounds):
    for two, run in pipes.items():
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        res[two].append((time.perf_counter() - t0) / steps * 1e3)
        host[two].append(th / steps * 1e3)
for two in pipes:
    v = sorted(res[two])
    print(f"{'two one-stream graphs + events' if two else 'one graph, fork / join inside '}: " + "  ".join(f"{t:.3f}" for t in res[two])
          + f"  ms/step; median {v[len(v) // 2]:.3f} -> {8 / v[len(v) // 2] * 1e3:.0f} images/s; host {min(host[two]):.3f} ms per step in run()")
