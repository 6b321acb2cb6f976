#!/usr/bin/env python3
"""How far do two trainers of one seed drift apart? Four runs of 12 iterations at the test size (ordinary, ordinary, graph, graph):
the reward history of each and the differences ordinary vs ordinary, graph vs graph, ordinary vs graph. One fp32 ulp in a filter
parameter flips bf16 roundings in the detector, so two ORDINARY runs differ by 1e-4 .. 3e-4 from the fourth iteration on; the
graph runs differ from them by the same amount (what tests/test_gpu_train_graph.py's history tolerance rests on).
usage (GPU box): python tools/train_graph_hist.py"""
import os, random, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import test_gpu_train_graph as T
from adaptiveisp_amd.replay import DeviceReplayMemory, SyntheticSource
from adaptiveisp_amd.train import Trainer
from adaptiveisp_amd.util import Dict
DEV = "cuda:0"
B, H, W, N = 4, 64, 96, 12
eng, loss_fn = T._detector(B, H, W)
hists = []
for mode in (False, False, True, True):
    cfg, agent, value = T._fresh(B)
    c = Dict(cfg); c.replay_memory_size = 16
    np.random.seed(0)
    replay = DeviceReplayMemory(c, SyntheticSource((3, H, W), nc=80, seed=2), B, DEV, (3, H, W), rng=random.Random(5))
    tr = Trainer(c, agent, value, eng, loss_fn, replay, batch_size=B, lr=3e-5, epochs=1, graph=mode)
    h = tr.train(iters=N); torch.cuda.synchronize()
    hists.append(h)
    print("mode", mode, " ".join(f"{r['reward']:+.6f}" for r in h), flush=True)
for a, b, tag in ((0, 1, "ord-ord"), (2, 3, "graph-graph"), (0, 2, "ord-graph")):
    print(tag, " ".join(f"{abs(x['reward'] - y['reward']):.1e}" for x, y in zip(hists[a], hists[b])))
