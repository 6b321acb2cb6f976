#!/usr/bin/env python3
"""Where one RL training iteration spends its time (synchronised timers around the stages of rl.train_iteration)."""
import os, random, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptiveisp_amd import dist as adist
from adaptiveisp_amd.agent import Agent
from adaptiveisp_amd.config import cfg
from adaptiveisp_amd.rl import td_losses
from adaptiveisp_amd.value import Value
from adaptiveisp_amd.yolo import YoloTrainEngine, yolov3
from adaptiveisp_amd.yolo.loss import DetectionLoss, default_hyp, assign_labels_packed

B, HW, DEV = 8, 512, "cuda:0"
torch.manual_seed(0)
det = yolov3().to(DEV).train()
for p in det.parameters():
    p.requires_grad_(False)
agent = Agent(cfg, shape=(16, 64, 64), device=DEV).to(DEV).train()
value = Value(cfg, shape=(19, 64, 64)).to(DEV).train()
loss_fn = DetectionLoss(det.model[-1].anchors, nc=80, hyp=default_hyp(80, HW), device=DEV)
eng = YoloTrainEngine(det, B, HW, HW, device=DEV)
eng.autotune(cache=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "adaptiveisp_amd", "yolo", "tuning", "mi355x.json"))
imgs = torch.rand(B, 3, HW, HW, device=DEV) ** 2.2 * 0.5
z = torch.rand(B, cfg.z_dim, device=DEV)
states = torch.zeros(B, cfg.num_state_dim, device=DEV)
labels = [torch.tensor([[0, b, 0.5, 0.5, 0.3, 0.4], [0, b + 1, 0.3, 0.3, 0.2, 0.2]]) for b in range(B)]
T = {}
def tic():
    torch.cuda.synchronize(); return time.perf_counter()
for it in range(4):
    t = tic(); (retouch, new_states, surrogate, penalty), _, _ = agent((imgs, z, states), 0.1); T["agent fwd"] = tic() - t
    t = tic()
    packed = assign_labels_packed(loss_fn, eng.head_shapes(), labels, DEV)
    T["target assignment (host) + copy"] = tic() - t
    t = tic()
    with torch.no_grad():
        l_in = eng.per_sample_loss(loss_fn, imgs, packed)
    T["detector fwd + fused loss (input)"] = tic() - t
    t = tic(); l_re = eng.per_sample_loss(loss_fn, retouch, packed); T["detector fwd + fused loss (retouch)"] = tic() - t
    t = tic(); ov = value(imgs, states); nv = value(retouch, new_states); T["value x2"] = tic() - t
    t = tic()
    out = td_losses(cfg, l_in, l_re, penalty, surrogate, new_states, ov, nv, torch.mean(retouch, dim=(1, 2, 3)).unsqueeze(-1))
    T["td math"] = tic() - t
    t = tic(); out["value_loss"].backward(); T["value backward"] = tic() - t
    t = tic(); out["agent_loss"].backward(); T["agent backward (loss bwd + detector bwd + ISP param grads + heads)"] = tic() - t
    agent.zero_grad(); value.zero_grad()
for k, v in T.items():
    print(f"{v * 1e3:8.2f} ms  {k}")
print(f"{sum(T.values()) * 1e3:8.2f} ms  total")
