#!/usr/bin/env python3
"""Per-image time of the evaluation loop at BASELINE config 3's shape (batch 1, 512 x 512 letterbox, 5 ISP steps with the
reference's per-step early-exit check, detector, NMS at conf 0.001, matching) on synthetic images and labels — random-init
weights, so the mAP means nothing; the time per stage does. usage: eval_bench.py [images=40] [batch=1]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptiveisp_amd.agent import Agent
from adaptiveisp_amd.config import cfg
from adaptiveisp_amd.val import harness
from adaptiveisp_amd.val.harness import run_eval
from adaptiveisp_amd.yolo import YoloEngine, yolov3

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
DEV = "cuda:0"
torch.manual_seed(0); np.random.seed(0)
agent = Agent(cfg, shape=(6 + len(cfg.filters), 64, 64), device=DEV).to(DEV).eval()
det = yolov3().eval()
eng = YoloEngine(det, B, 512, 512, device=DEV)
eng.autotune(cache=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "adaptiveisp_amd", "yolo", "tuning", "mi355x.json"))
g = torch.Generator().manual_seed(1)


def make_batches(k):
    out = []
    for i in range(k):
        im = torch.rand(B, 3, 512, 512, generator=g) ** 2.2 * 0.5
        t = torch.zeros(3 * B, 6)
        t[:, 0] = torch.arange(3 * B) // 3
        t[:, 1] = torch.randint(0, 80, (3 * B,), generator=g).float()
        t[:, 2:4] = torch.rand(3 * B, 2, generator=g) * 0.6 + 0.2
        t[:, 4:6] = torch.rand(3 * B, 2, generator=g) * 0.3 + 0.05
        out.append((im.pin_memory(), t, [f"img{i}_{b}.png" for b in range(B)], [((512, 512), ((1.0, 1.0), (0.0, 0.0)))] * B))
    return out


# stage timers: wrap the pieces run_eval calls
T = {"isp steps": 0.0, "detector": 0.0, "nms": 0.0}
_agent_fwd = agent.forward
def agent_timed(*a, **k):
    torch.cuda.synchronize(); t = time.perf_counter(); r = _agent_fwd(*a, **k); torch.cuda.synchronize(); T["isp steps"] += time.perf_counter() - t; return r
agent.forward = agent_timed
def det_timed(x):
    torch.cuda.synchronize(); t = time.perf_counter(); r = eng(x); torch.cuda.synchronize(); T["detector"] += time.perf_counter() - t; return r
_nms = harness.non_max_suppression
def nms_timed(*a, **k):
    torch.cuda.synchronize(); t = time.perf_counter(); r = _nms(*a, **k); torch.cuda.synchronize(); T["nms"] += time.perf_counter() - t; return r
harness.non_max_suppression = nms_timed

data = make_batches(n)                                           # synthetic frames are made before the clock starts
run_eval(agent, det_timed, data[:3], cfg)                        # warm-up
for k in T: T[k] = 0.0
torch.cuda.synchronize(); t0 = time.perf_counter()
res = run_eval(agent, det_timed, data, cfg)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{n} batches of {B}: {dt / n * 1e3:.2f} ms per batch ({n * B / dt:.1f} images/s), seen {res['seen']}")
for k, v in T.items():
    print(f"   {v / n * 1e3:7.2f} ms  {k}")
print(f"   {(dt - sum(T.values())) / n * 1e3:7.2f} ms  matching, AP bookkeeping, host glue")
