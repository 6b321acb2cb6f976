#!/usr/bin/env python3
"""Is ONE training forward of 16 images cheaper than two of 8 (the input batch and the retouched batch of an RL iteration)?"""
import os, shutil, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from adaptiveisp_amd.yolo import YoloTrainEngine, yolov3
DEV = "cuda:0"
TABLE = os.path.join(ROOT, "adaptiveisp_amd", "yolo", "tuning", "mi355x.json")
TMP = "/tmp/mi355x_b16.json"
shutil.copy(TABLE, TMP)
torch.manual_seed(0)
det = yolov3().to(DEV).train()
for p in det.parameters():
    p.requires_grad_(False)
for B in (8, 16):
    x = torch.rand(B, 3, 512, 512, device=DEV)
    eng = YoloTrainEngine(det, B, 512, 512, device=DEV)
    eng.autotune(cache=TMP, write=True)
    for name, fn in (("forward", lambda: eng._forward_raw(x)), ("backward", lambda: eng._backward_raw())):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            fn()
        e1.record(); e1.synchronize()
        print(f"B={B} {name}: {e0.elapsed_time(e1) / 30:.3f} ms")
    del eng
