#!/usr/bin/env python3
"""ATen ops per stage of one RL training iteration (TorchDispatchMode counter; forward stages counted where they are
called, the backward as a whole): which stages the host's enqueue work is made of. usage: train_op_count.py"""
import collections, os, random, sys
import numpy as np
import torch
from torch.utils._python_dispatch import TorchDispatchMode
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptiveisp_amd import rl
from adaptiveisp_amd.agent import Agent
from adaptiveisp_amd.config import cfg
from adaptiveisp_amd.replay import DeviceReplayMemory, SyntheticSource
from adaptiveisp_amd.train import Trainer
from adaptiveisp_amd.value import Value
from adaptiveisp_amd.yolo import YoloTrainEngine, YoloTrainPairEngine, yolov3
from adaptiveisp_amd.yolo.loss import DetectionLoss, default_hyp

B, HW, DEV = 8, 512, "cuda:0"
torch.manual_seed(0); np.random.seed(0)
det = yolov3().to(DEV).train()
for p in det.parameters():
    p.requires_grad_(False)
agent = Agent(cfg, shape=(16, 64, 64), device=DEV).to(DEV)
value = Value(cfg, shape=(19, 64, 64)).to(DEV)
loss_fn = DetectionLoss(det.model[-1].anchors, nc=80, hyp=default_hyp(80, HW), device=DEV)
replay = DeviceReplayMemory(cfg, SyntheticSource((3, HW, HW), seed=1, device=DEV), B, DEV, (3, HW, HW), rng=random.Random(1))
detector = (YoloTrainPairEngine if os.environ.get("ADAYOLO_TRAIN_PAIR", "1") == "1" else YoloTrainEngine)(det, B, HW, HW, device=DEV)
detector.autotune(cache=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "adaptiveisp_amd", "yolo", "tuning", "mi355x.json"), write=False)
tr = Trainer(cfg, agent, value, detector, loss_fn, replay, batch_size=B)
tr.train(2)

STAGE = ["other"]
COUNT = collections.defaultdict(collections.Counter)


class Counter(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        COUNT[STAGE[0]][func.overloadpacket.__name__] += 1
        return func(*args, **(kwargs or {}))


def staged(name, fn):
    def call(*a, **k):
        prev, STAGE[0] = STAGE[0], name
        try:
            return fn(*a, **k)
        finally:
            STAGE[0] = prev
    return call


agent.forward = staged("agent forward", agent.forward)
value.forward = staged("value forward (x2)", value.forward)
rl.td_losses = staged("td_losses", rl.td_losses)
_bw = torch.Tensor.backward
torch.Tensor.backward = staged("backward (x2 calls; autograd nodes dispatch here)", _bw)
import adaptiveisp_amd.dist as adist
adist.synced_step = staged("clip + Adam", adist.synced_step)

replay.replace_memory = staged("replay.replace_memory", replay.replace_memory)
replay.get_feed_dict_and_states = staged("replay.get_feed_dict", replay.get_feed_dict_and_states)
with Counter():
    tr.step()
total = 0
for st, c in sorted(COUNT.items(), key=lambda kv: -sum(kv[1].values())):
    n = sum(c.values())
    total += n
    print(f"{n:5d} ops  {st}:  " + ", ".join(f"{k} {v}" for k, v in c.most_common(14)))
print(f"{total:5d} ops in the iteration")
