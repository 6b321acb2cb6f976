#!/usr/bin/env python3
"""How far is one RL iteration (config 4's per-rank shape) from being capturable as a hipGraph? Runs three eager iterations, then
tries to capture rl.train_iteration on a fixed feed and reports the first thing that stops it. Measurement /
feasibility aid only. usage (GPU box): python tools/train_graph_probe.py"""
import os
import random
import sys
import traceback

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptiveisp_amd import rl  # noqa: E402
from adaptiveisp_amd.agent import Agent  # noqa: E402
from adaptiveisp_amd.config import cfg  # noqa: E402
from adaptiveisp_amd.replay import DeviceReplayMemory, SyntheticSource  # noqa: E402
from adaptiveisp_amd.train import Trainer  # noqa: E402
from adaptiveisp_amd.value import Value  # noqa: E402
from adaptiveisp_amd.yolo import YoloTrainPairEngine, yolov3  # noqa: E402
from adaptiveisp_amd.yolo.loss import DetectionLoss, default_hyp  # noqa: E402

B, HW, DEV = 8, 512, "cuda:0"
torch.manual_seed(0)
np.random.seed(0)
det = yolov3().to(DEV).train()
for m in det.modules():
    if isinstance(m, torch.nn.BatchNorm2d):
        m.eval()
for p in det.parameters():
    p.requires_grad_(False)
agent = Agent(cfg, shape=(16, 64, 64), device=DEV).to(DEV)
value = Value(cfg, shape=(19, 64, 64)).to(DEV)
loss_fn = DetectionLoss(det.model[-1].anchors, nc=80, hyp=default_hyp(80, HW), device=DEV)
replay = DeviceReplayMemory(cfg, SyntheticSource((3, HW, HW), seed=1, device=DEV), B, DEV, (3, HW, HW), rng=random.Random(1))
detector = YoloTrainPairEngine(det, B, HW, HW, device=DEV)
detector.autotune(cache=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "adaptiveisp_amd", "yolo", "tuning", "mi355x.json"), write=False)
tr = Trainer(cfg, agent, value, detector, loss_fn, replay, batch_size=B)
tr.train(3)
torch.cuda.synchronize()
print("three eager iterations done", flush=True)
feed = replay.get_feed_dict_and_states(B)
labels = [torch.as_tensor(lb) for lb in feed["label"]]
im, z, st = feed["im"].clone(), feed["z"].clone(), feed["state"].clone()
torch.cuda.synchronize()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):                       # (the usual warm-up on a side stream before a capture)
    rl.train_iteration(cfg, agent, value, detector, loss_fn, im, z, st, labels, 0.1, [tr.agent_optimizer, tr.value_optimizer], buckets=tr.buckets)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
print("warm-up iteration on a side stream done", flush=True)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        out = rl.train_iteration(cfg, agent, value, detector, loss_fn, im, z, st, labels, 0.1, [tr.agent_optimizer, tr.value_optimizer], buckets=tr.buckets)
    print("CAPTURED one iteration; replaying 3x", flush=True)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print("replays done; agent_loss", float(out["agent_loss"]), flush=True)
except BaseException as e:  # noqa: BLE001
    print("CAPTURE STOPPED:", type(e).__name__, str(e)[:600], flush=True)
    tb = traceback.extract_tb(e.__traceback__)
    for fr in tb[-8:]:
        print(f"   {os.path.relpath(fr.filename)}:{fr.lineno} in {fr.name}: {fr.line}", flush=True)
