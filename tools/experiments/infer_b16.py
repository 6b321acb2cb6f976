#!/usr/bin/env python3
"""Inference forward at 8 x 720 x 1280 (the headline's detector) against ONE forward over 16 images: per-image cost."""
import os, shutil, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from adaptiveisp_amd.yolo import YoloEngine, yolov3
DEV = "cuda:0"
TMP = "/tmp/mi355x_inf16.json"
shutil.copy(os.path.join(ROOT, "adaptiveisp_amd", "yolo", "tuning", "mi355x.json"), TMP)
torch.manual_seed(0)
det = yolov3().to(DEV).eval()
for B in (8, 16):
    x = torch.rand(B, 3, 720, 1280, device=DEV)
    eng = YoloEngine(det, B, 720, 1280, device=DEV)
    eng.autotune(cache=TMP, write=True)
    g = torch.cuda.CUDAGraph()
    eng(x); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        eng(x)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        g.replay()
    e1.record(); e1.synchronize()
    t = e0.elapsed_time(e1) / 30
    print(f"B={B}: {t:.3f} ms per forward, {t / B:.4f} ms per image")
    del eng, g
