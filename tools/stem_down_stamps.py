#!/usr/bin/env python3
"""Per-workgroup phase timeline of the fused stem kernel (needs a library built with
ADAYOLO_EXTRA_FLAGS=-DADAYOLO_MEASURE python -m adaptiveisp_amd.build --force): medians of the phase durations."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptiveisp_amd.yolo import YoloEngine, yolov3, _lib
torch.manual_seed(1)
eng = YoloEngine(yolov3().eval(), 8, 720, 1280)
eng.autotune(cache=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "adaptiveisp_amd", "yolo", "tuning", "mi355x.json"))
x = torch.rand(8, 3, 720, 1280, device="cuda")
for _ in range(3):
    eng(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
# the head launch alone: run the engine's forward but time only a replay of the first launch via its own path
import time
N = 4096
buf = (ctypes.c_ulonglong * (N * 12))()
eng(x); torch.cuda.synchronize()
assert eng.L.adayolo_debug_stem_down(buf, N * 12) == 0
a = np.ctypeslib.as_array(buf).reshape(N, 12).astype(np.int64)
names = ["setup + weight / image loads issued + image patch stored", "barrier", "B: stem conv + SiLU -> patch", "barrier",
         "C: second conv (36 MFMAs per wave)", "barrier", "D: bias + SiLU -> LDS tile", "barrier", "stores", "E: 1x1 + SiLU + stores"]
last = 10 if eng._head_next is not None else 9            # (round 6: with the whole-Bottleneck launch the 1x1 stage E is not run)
names = names[:last]
d = np.diff(a[:, :last + 1], axis=1)
ok = (a[:, last] > a[:, 0])
print(f"{ok.sum()} workgroups sampled; whole workgroup median {np.median(a[ok, last] - a[ok, 0]):.0f} cycles")
for i, n in enumerate(names):
    print(f"  {np.median(d[ok, i]):8.0f}  p90 {np.percentile(d[ok, i], 90):8.0f}   {n}")
span = a[ok, last].max() - a[ok, 0].min()
print(f"first stamp to last stamp: {span} cycles")
