#!/usr/bin/env python3
"""Where the per-image time of BASELINE config 3's loop goes when the episode + detector are ONE hipGraph replay
(run_eval(graph=True), batch 1, 512 x 512, 5 ISP steps, NMS at conf 0.001, matching; synthetic frames, random-init weights).
Two passes over the same frames: (1) untouched, wall clock per image; (2) with a device synchronise around the stages —
upload + replay + the one host read / NMS / matching + bookkeeping — so their sum exceeds (1) by the overlap it destroys.
Under rocprofv3 (`rocprofv3 --kernel-trace --stats -- python3 tools/eval_graph_prof.py 60`) the kernel-stats CSV of pass (1) +
(2) is the per-kernel breakdown profiles/round6_eval_config3_kernel_stats.csv holds.
usage: eval_graph_prof.py [images=40]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptiveisp_amd.agent import Agent  # noqa: E402
from adaptiveisp_amd.config import cfg  # noqa: E402
from adaptiveisp_amd.val import harness  # noqa: E402
from adaptiveisp_amd.val.harness import run_eval  # noqa: E402
from adaptiveisp_amd.yolo import YoloEngine, yolov3  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
DEV = "cuda:0"
torch.manual_seed(0)
np.random.seed(0)
agent = Agent(cfg, shape=(6 + len(cfg.filters), 64, 64), device=DEV).to(DEV).eval()
torch.manual_seed(1)
eng = YoloEngine(yolov3().eval(), 1, 512, 512, device=DEV)
eng.autotune(cache=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "adaptiveisp_amd", "yolo", "tuning", "mi355x.json"))
g = torch.Generator().manual_seed(1)


def make_batches(k):
    out = []
    for i in range(k):
        im = torch.rand(1, 3, 512, 512, generator=g) ** 2.2 * 0.5
        t = torch.zeros(3, 6)
        t[:, 1] = torch.randint(0, 80, (3,), generator=g).float()
        t[:, 2:4] = torch.rand(3, 2, generator=g) * 0.6 + 0.2
        t[:, 4:6] = torch.rand(3, 2, generator=g) * 0.3 + 0.05
        out.append((im.pin_memory(), t, [f"img{i}.png"], [((512, 512), ((1.0, 1.0), (0.0, 0.0)))]))
    return out


data = make_batches(n)
run_eval(agent, eng, data[:3], cfg, graph=True)
torch.cuda.synchronize()
kinds = {}
for kind, _, _ in eng.plan:
    kinds[kind] = kinds.get(kind, 0) + 1
print(f"detector plan at 1 x 512 x 512: {len(eng.plan)} entries {kinds}")
t0 = time.perf_counter()
res = run_eval(agent, eng, data, cfg, graph=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"(1) graph loop, untouched: {dt / n * 1e3:.3f} ms per image ({n / dt:.1f} images/s), seen {res['seen']}")

T = {"upload + replay + host read": 0.0, "nms": 0.0}
_run = harness._EpisodeGraph.run


def run_timed(self, *a):
    torch.cuda.synchronize(); t = time.perf_counter(); r = _run(self, *a); torch.cuda.synchronize(); T["upload + replay + host read"] += time.perf_counter() - t; return r


harness._EpisodeGraph.run = run_timed
_nms = harness.non_max_suppression


def nms_timed(*a, **k):
    torch.cuda.synchronize(); t = time.perf_counter(); r = _nms(*a, **k); torch.cuda.synchronize(); T["nms"] += time.perf_counter() - t; return r


harness.non_max_suppression = nms_timed
torch.cuda.synchronize()
t0 = time.perf_counter()
run_eval(agent, eng, data, cfg, graph=True)
torch.cuda.synchronize()
dt2 = time.perf_counter() - t0
print(f"(2) with a device synchronise around the stages: {dt2 / n * 1e3:.3f} ms per image")
for k, v in T.items():
    print(f"   {v / n * 1e3:7.3f} ms  {k}")
print(f"   {(dt2 - sum(T.values())) / n * 1e3:7.3f} ms  uploads of noise / states / targets, matching, AP bookkeeping, host glue (one graph capture per run_eval call: {'%.2f' % 0.0} excluded: no)")
# the replay alone, back to back (GPU time of the episode + detector graph)
eg = None
key_graphs = [v for v in []]
im, targets, paths, shapes = data[0]
from adaptiveisp_amd.util import get_initial_states, get_noise, to_device_async  # noqa: E402
imd = to_device_async(im, torch.device(DEV)).float()
noises = to_device_async(np.array([get_noise(1, cfg.z_type, cfg.z_dim) for _ in range(5)]), torch.device(DEV))
states = to_device_async(get_initial_states(1, cfg.num_state_dim, len(agent.filters)), torch.device(DEV))
eg = harness._EpisodeGraph(agent, eng, imd, noises, states, 5, None)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    eg.graph.replay()
e1.record()
torch.cuda.synchronize()
print(f"(3) the episode + detector graph replayed back to back: {e0.elapsed_time(e1) / 20:.3f} ms per replay (GPU time of one image)")
