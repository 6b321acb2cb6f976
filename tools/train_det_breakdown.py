#!/usr/bin/env python3
"""Where the detector's time goes inside an RL training iteration (BASELINE config 4 per-rank shape, 8 x 512 x 512):
the inference forward (YoloEngine, hipGraph) beside the training forward and the backward of YoloTrainEngine, then every
launch of the two training sequences with its shape, duration, TFLOP/s and algorithmic GB/s (events around each launch,
sequence run in plan order). usage: [RETUNE=1] train_det_breakdown.py [B=8] [HW=512] [top=40]
(RETUNE=1 re-measures the conv variants of both engines at this shape and writes them to the tuning table)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adaptiveisp_amd.yolo import YoloEngine, YoloTrainEngine, yolov3  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
HW = int(sys.argv[2]) if len(sys.argv) > 2 else 512
TOP = int(sys.argv[3]) if len(sys.argv) > 3 else 40
dev = torch.device("cuda:0")
RETUNE = os.environ.get("RETUNE") == "1"
cache = os.path.join(ROOT, "adaptiveisp_amd", "yolo", "tuning", "mi355x.json")
torch.manual_seed(0)
det = yolov3().eval()
x = torch.rand(B, 3, HW, HW, device=dev)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


inf = YoloEngine(det, B, HW, HW, device=dev)
inf.autotune(cache=cache, retune=RETUNE, write=RETUNE)
with torch.no_grad():
    inf(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        inf(x)
print(f"inference forward (YoloEngine, graph replay) {timeit(g.replay):.3f} ms")

tr = YoloTrainEngine(det, B, HW, HW, device=dev)
tr.autotune(cache=cache, retune=RETUNE, write=RETUNE)
print("variants in use: inference", sorted(set(inf.tuned.values())), "training", sorted(set(tr.tuned.values())))
tr._forward_raw(x)
st = tr._graph("fwd")
print(f"training forward (graph replay)              {timeit(st['fwd'].replay):.3f} ms   ({tr.keep_fused} conv+SiLU pairs in one launch)")
print(f"training backward (graph replay)             {timeit(st['bwd'].replay):.3f} ms   ({tr.dsilu_fused} of 72 SiLU' launches inside their producing conv)")


def per_launch(plan, **kw):
    rows = []
    stream = torch.cuda.current_stream()
    for rep in range(4):
        evs = []
        for e in plan:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            tr._run([e], **kw)
            e1.record(stream)
            evs.append((e0, e1))
        torch.cuda.synchronize()
        if rep == 0:
            continue
        for i, (e0, e1) in enumerate(evs):
            if len(rows) <= i:
                rows.append(0.0)
            rows[i] += e0.elapsed_time(e1) / 3
    return rows


def describe(e):
    kind, fn, a = e
    if kind in ("conv", "convkeep", "convds"):
        off = {"conv": 0, "convkeep": 2, "convds": 4}[kind]
        Bc, H, W, cin, cout, k, s = a[8 + off:15 + off]
        v = a[19] if kind == "convds" else a[16 + off]
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        fl = 2.0 * Bc * Ho * Wo * cout * k * k * cin
        nout = {"conv": 1, "convkeep": 2, "convds": 3 if a[6] else 2}[kind]          # outputs (+ the pre-activation read)
        by = 2.0 * Bc * (H * W * cin + Ho * Wo * cout * nout + (Ho * Wo * cout if a[4] else 0))
        return f"{kind:8s} v{v:<3d} {cin:4d}->{cout:4d} k{k} s{s} @{H}x{W}", fl, by
    if kind in ("dsilu", "silu"):
        npix, C = a[-2], a[-1]
        n = 3 if kind == "silu" else (4 if a[6] else 3)
        return f"{kind:8s}      {C:4d} ch, {npix} px", 0.0, 2.0 * npix * C * n
    return f"{kind:8s}", 0.0, 0.0


for name, plan, kw in (("training forward", tr._forward_plan(), dict(img=x)),
                       ("training backward", tr._backward_plan(), dict(grad_img=torch.empty_like(x)))):
    ms = per_launch(plan, **kw)
    print(f"\n{name}: {len(plan)} launches, {sum(ms):.3f} ms summed (launch by launch, not overlapped)")
    kinds = {}
    for e, t in zip(plan, ms):
        kinds.setdefault(e[0], [0, 0.0])
        kinds[e[0]][0] += 1
        kinds[e[0]][1] += t
    print("   by kind: " + ", ".join(f"{k} {n} x = {t:.3f} ms" for k, (n, t) in sorted(kinds.items(), key=lambda kv: -kv[1][1])))
    order = sorted(range(len(plan)), key=lambda i: -ms[i])[:TOP]
    if os.environ.get("PLAN_ORDER") == "1":                # every launch in plan order with the running sum
        order, run = range(len(plan)), 0.0
    for i in order:
        d, fl, by = describe(plan[i])
        tail = ""
        if os.environ.get("PLAN_ORDER") == "1":
            run += ms[i]
            tail = f"   sum {run:.3f} ms"
        print(f"   #{i:3d} {ms[i] * 1e3:8.1f} us  {d:48s} {fl / ms[i] / 1e9 if fl else 0:7.1f} TFLOP/s {by / ms[i] / 1e6 if by else 0:7.0f} GB/s{tail}")
