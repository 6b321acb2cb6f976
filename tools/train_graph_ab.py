#!/usr/bin/env python3
"""BASELINE config 4, one rank's share (batch 8 x 512 x 512): the ordinary loop against the iteration as one hipGraph
(train.Trainer graph mode), one process, alternating blocks. Per mode: ms per iteration, the host's time inside step() that is
not waiting for the device (enqueue + bookkeeping), its waits. usage (GPU box): python tools/train_graph_ab.py [--iters 40] [--rounds 3]"""
import argparse
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptiveisp_amd import train as atrain  # noqa: E402
from adaptiveisp_amd.config import cfg  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--early", default="", help="comma list of ADAYOLO_TRAIN_EARLY values: graph-mode trainers only, one per value")
ap.add_argument("--streams", default="", help="comma list of ADAISP_TRAIN_GRAPH_STREAMS values (1, 2): graph-mode trainers, one per value")
ap.add_argument("--headsk", default="", help="comma list of ADAISP_HEADS_KERNEL values (1, 0): graph-mode trainers, one per value (the "
                "switch is read while the iteration is captured)")
ap.add_argument("--ordinary", action="store_true", help="with --early: the ordinary loop instead of graph mode")
a = ap.parse_args()
dev = torch.device("cuda:0")
cache = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "adaptiveisp_amd", "yolo", "tuning", "mi355x.json")
waited = [0.0]


def timed(fn):
    def call(*x, **k):
        t = time.perf_counter()
        r = fn(*x, **k)
        waited[0] += time.perf_counter() - t
        return r
    return call


torch.cuda.Event.synchronize = timed(torch.cuda.Event.synchronize)
atrain._GraphIteration.wait_guard = timed(atrain._GraphIteration.wait_guard)
trainers = {}
modes = [("early=" + v) for v in a.early.split(",")] if a.early else [False, True]
if a.streams:
    modes = [("streams=" + v) for v in a.streams.split(",")]
if a.headsk:
    modes = [("headsk=" + v) for v in a.headsk.split(",")]
for mode in modes:
    if isinstance(mode, str) and mode.startswith("headsk="):
        os.environ["ADAISP_TRAIN_GRAPH"], os.environ["ADAISP_HEADS_KERNEL"] = "1", mode.split("=")[1]
    elif isinstance(mode, str) and mode.startswith("streams="):
        os.environ["ADAISP_TRAIN_GRAPH"], os.environ["ADAISP_TRAIN_GRAPH_STREAMS"] = "1", mode.split("=")[1]
    elif isinstance(mode, str):
        os.environ["ADAISP_TRAIN_GRAPH"], os.environ["ADAYOLO_TRAIN_EARLY"] = "0" if a.ordinary else "1", mode.split("=")[1]
    else:
        os.environ["ADAISP_TRAIN_GRAPH"] = "1" if mode else "0"
    trainers[mode] = atrain.build_trainer(cfg, 0, 1, dev, a.batch, a.size, tune_cache=cache, seed=0)
    assert trainers[mode].graph_mode is (bool(mode) and not a.ordinary)
    trainers[mode].train(6)
    torch.cuda.synchronize()
res = {m: [] for m in trainers}
for rnd in range(a.rounds):
    for mode, tr in trainers.items():
        torch.cuda.synchronize()
        waited[0] = 0.0
        t0 = time.perf_counter()
        for _ in range(a.iters):
            tr.step()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[mode].append((dt / a.iters * 1e3, (t_host - waited[0]) / a.iters * 1e3, waited[0] / a.iters * 1e3))
        tr.materialize()
        nonfinite = [r["iter"] for r in tr.history if not all(x == x and abs(x) != float("inf") for x in (r["agent_loss"], r["value_loss"], r["reward"]))]
        if nonfinite:
            print(f"   !! {mode}: non-finite losses from iteration {nonfinite[0]} on ({len(nonfinite)} so far; dropped batches {sum(r['dropped'] for r in tr.history)})", flush=True)
        print(f"round {rnd} {mode if isinstance(mode, str) else ('graph   ' if mode else 'ordinary')}: {res[mode][-1][0]:.3f} ms / iteration, host busy {res[mode][-1][1]:.3f} ms, "
              f"host waiting {res[mode][-1][2]:.3f} ms", flush=True)
for mode, v in res.items():
    tr = trainers[mode]
    tr.materialize()
    last = tr.history[-1]
    print(f"{mode if isinstance(mode, str) else ('graph   ' if mode else 'ordinary')}: median {statistics.median(x[0] for x in v):.3f} ms / iteration ({a.batch * 1e3 / statistics.median(x[0] for x in v):.0f} images/s), "
          f"host busy {statistics.median(x[1] for x in v):.3f} ms, waiting {statistics.median(x[2] for x in v):.3f} ms; "
          f"last losses agent {last['agent_loss']:.5f} value {last['value_loss']:.5f} reward {last['reward']:.5f}; iterations {tr.iter}")
if os.environ.get("TRAIN_GRAPH_AB_PROFILE") == "1":          # where the host's time inside a graph-mode step() goes
    import cProfile
    import pstats
    tr = trainers[True] if True in trainers else list(trainers.values())[-1]
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(a.iters):
        tr.step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(45)
