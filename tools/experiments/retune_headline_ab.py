#!/usr/bin/env python3
"""The committed tuning table against a fresh re-measurement of every layer of the headline's detector (8 x 720 x 1280), both
engines in ONE process, graph replays interleaved. Writes the re-measured table to gpurun_out/mi355x_headline_retuned.json."""
import os, shutil, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from adaptiveisp_amd.yolo import YoloEngine, yolov3
DEV = "cuda:0"
TABLE = os.path.join(ROOT, "adaptiveisp_amd", "yolo", "tuning", "mi355x.json")
OUT = os.path.join(ROOT, "gpurun_out", "mi355x_headline_retuned.json")
os.makedirs(os.path.dirname(OUT), exist_ok=True)
shutil.copy(TABLE, OUT)
torch.manual_seed(0)
det = yolov3().to(DEV).eval()
x = torch.rand(8, 3, 720, 1280, device=DEV)
engs, graphs = {}, {}
for name in ("committed", "retuned"):
    e = YoloEngine(det, 8, 720, 1280, device=DEV)
    if name == "committed":
        e.autotune(cache=TABLE, write=False)
    else:
        e.autotune(cache=OUT, retune=True, write=True, reps=9)
    e(x); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        e(x)
    engs[name], graphs[name] = e, g
moved = {k: (engs["committed"].tuned.get(k), v) for k, v in engs["retuned"].tuned.items() if engs["committed"].tuned.get(k) != v}
print(len(moved), "layer shapes moved:", moved)
assert torch.allclose(engs["committed"].pred.float(), engs["retuned"].pred.float(), atol=0.5, rtol=0.1)
res = {k: [] for k in graphs}
for r in range(6):
    for k, g in graphs.items():
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            g.replay()
        e1.record(); e1.synchronize()
        res[k].append(e0.elapsed_time(e1) / 40)
for k, v in res.items():
    print(f"{k:10s}: " + "  ".join(f"{t:.3f}" for t in v) + " ms")
