#!/usr/bin/env python3
"""Re-measure the kernel choice of every conv of the TRAINING engine at the config-4 per-rank shape (8 x 512 x 512) and write
the merged table to gpurun_out/mi355x_retuned.json (the committed table is left alone); prints the layers whose choice moved
and the detector's forward / backward times with both tables. usage: retune_train.py [B=8] [HW=512]"""
import os, shutil, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adaptiveisp_amd.yolo import YoloTrainEngine, yolov3

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
HW = int(sys.argv[2]) if len(sys.argv) > 2 else 512
DEV = "cuda:0"
TABLE = os.path.join(ROOT, "adaptiveisp_amd", "yolo", "tuning", "mi355x.json")
OUT = os.path.join(ROOT, "gpurun_out", "mi355x_retuned.json")
os.makedirs(os.path.dirname(OUT), exist_ok=True)
shutil.copy(TABLE, OUT)
torch.manual_seed(0)
det = yolov3().to(DEV).train()
for p in det.parameters():
    p.requires_grad_(False)
x = torch.rand(B, 3, HW, HW, device=DEV)


def time_engine(eng, reps=30):
    out = []
    for fn in (lambda: eng._forward_raw(x), lambda: eng._backward_raw()):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        out.append(e0.elapsed_time(e1) / reps)
    return out


eng = YoloTrainEngine(det, B, HW, HW, device=DEV)
old = dict(eng.autotune(cache=TABLE, write=False))
t_old = time_engine(eng)
new = dict(eng.autotune(cache=OUT, retune=True, write=True, reps=9))
t_new = time_engine(eng)
moved = {k: (old.get(k), v) for k, v in new.items() if old.get(k) != v}
print(f"table {len(old)} keys; {len(moved)} moved")
for k, (a, b) in sorted(moved.items()):
    print("  ", k, a, "->", b)
print(f"forward {t_old[0]:.3f} -> {t_new[0]:.3f} ms, backward {t_old[1]:.3f} -> {t_new[1]:.3f} ms "
      f"(keep-fused {eng.keep_fused}, dsilu-fused {getattr(eng, 'dsilu_fused', '?')})")
