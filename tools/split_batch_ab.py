#!/usr/bin/env python3
"""Does the detector run faster as TWO half-batch forwards on two streams (the other half's workgroups fill the partial
last round / prologue-epilogue bubbles of every launch: work-conserving across kernel boundaries) than as one batch-8
forward? A/B in one process, hipGraph replay, interleaved."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adaptiveisp_amd.yolo import YoloEngine, yolov3  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(1)
det = yolov3().eval()
B, H, W = 8, 720, 1280
cache = os.path.join(ROOT, "adaptiveisp_amd", "yolo", "tuning", "mi355x.json")
g = torch.Generator(device="cpu").manual_seed(1235)
x = (torch.rand(B, 3, H, W, generator=g) ** 2.2 * 0.5).to(dev)
full = YoloEngine(det, B, H, W, device=dev)
full.autotune(cache=cache)
parts = int(os.environ.get("PARTS", "2"))
halves = [YoloEngine(det, B // parts, H, W, device=dev) for _ in range(parts)]
for h in halves:
    h.autotune(cache=cache, write=True)
streams = [torch.cuda.Stream() for _ in range(parts)]


def run_full():
    full(x)


def run_split():
    cur = torch.cuda.current_stream()
    for i, (h, s) in enumerate(zip(halves, streams)):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            h(x[i * (B // parts):(i + 1) * (B // parts)])
    for s in streams:
        cur.wait_stream(s)


def graphed(fn):
    with torch.no_grad():
        fn(); torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            fn()
    return gr.replay


def timeit(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    a, b = graphed(run_full), graphed(run_split)
    ref = full(x).clone()
    run_split()
    torch.cuda.synchronize()
    got = torch.cat([h.pred for h in halves], 0)
    print("max |split - full| on the decoded prediction:", float((got - ref).abs().max()), "(tuned variants may differ per batch size)")
    for r in range(4):
        print(f"round {r}: one batch-{B} forward {timeit(a):.3f} ms | {parts} x batch-{B // parts} on {parts} streams {timeit(b):.3f} ms", flush=True)
