#!/usr/bin/env python3
"""Interleaved A/B in ONE process of the tuned detector forward at the benchmark's size under two (or more) ENVIRONMENT settings
that YoloEngine reads while it builds its plan (ADAYOLO_K1, ADAYOLO_CHAIN, ADAYOLO_FUSE_1X1, ADAYOLO_BNECK, ...): one engine
per setting, each forward a replayed hipGraph, `rounds` x `reps` replays interleaved; prediction differences are printed; per
plan-entry kind the launches whose count differs are timed one by one (event pairs) so that the moved layers can be read off.
usage (GPU box): python tools/engine_env_ab.py "ADAYOLO_K1=0" "ADAYOLO_K1=1" [--rounds 10] [--per-layer]"""
import argparse
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("settings", nargs="+")
    ap.add_argument("--rounds", type=int, default=10)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--per-layer", action="store_true")
    a = ap.parse_args()
    from _synth import synth_yolo_state_dict, test_image
    from adaptiveisp_amd.yolo import YoloEngine, _lib, yolov3
    dev = torch.device("cuda:0")
    tune = os.path.join(ROOT, "adaptiveisp_amd", "yolo", "tuning", "mi355x.json")
    det = yolov3()
    det.load_state_dict(synth_yolo_state_dict(det, seed=2))
    det = det.eval()
    x = torch.from_numpy(test_image(a.batch, a.height, a.width, seed=3, special=False)).to(dev)
    engines, graphs, preds = {}, {}, {}
    for s in a.settings:
        kv = dict(p.split("=", 1) for p in s.split() if "=" in p)
        old = {k: os.environ.get(k) for k in kv}
        os.environ.update(kv)
        try:
            e = YoloEngine(det, a.batch, a.height, a.width, device=dev)
            e.autotune(cache=tune, write=False)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        engines[s] = e
        preds[s] = e(x).clone()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            e(x)
        graphs[s] = g
        kinds = {}
        for kind, _, _ in e.plan:
            kinds[kind] = kinds.get(kind, 0) + 1
        print(f"[{s}] plan: {len(e.plan)} entries {kinds}", flush=True)
    base = a.settings[0]
    for s in a.settings[1:]:
        d = (preds[s] - preds[base]).abs()
        print(f"[{s}] vs [{base}]: prediction bit-identical {torch.equal(preds[s], preds[base])}, max |d| {d.max().item():.3e} "
              f"(max |pred| {preds[base].abs().max().item():.1f})")
    times = {s: [] for s in a.settings}
    for r in range(a.rounds):
        order = a.settings if r % 2 == 0 else list(reversed(a.settings))
        for s in order:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            graphs[s].replay()
            e0.record()
            for _ in range(a.reps):
                graphs[s].replay()
            e1.record()
            torch.cuda.synchronize()
            times[s].append(e0.elapsed_time(e1) / a.reps)
    for s in a.settings:
        t = times[s]
        print(f"detector forward [{s}]: median {statistics.median(t):.4f} ms  min {min(t):.4f}  max {max(t):.4f}   "
              f"/ [{base}] = {statistics.median(t) / statistics.median(times[base]):.4f}", flush=True)
    if a.per_layer:
        st = _lib.stream_ptr()
        for s, e in engines.items():
            first = 3 if e._head_next is not None else 2
            rows = {}
            for kind, fn, args in e.plan[first:]:
                ts = []
                for _ in range(7):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    fn(*args, st)
                    e1.record()
                    e1.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3)
                if kind == "bneckws":
                    key = f"bneckws {args[9]}x{args[10]} C{args[11]}"
                elif kind == "k1":
                    key = f"k1 {args[7]}x{args[8]} {args[9]}->{args[10]}"
                elif kind in ("conv", "conv2"):
                    key = f"{kind} {args[9]}x{args[10]} {args[11]}->{args[12]} k{args[13]}s{args[14]} v{args[16] if kind == 'conv' else 'fused'}"
                else:
                    key = kind
                r_ = rows.setdefault(key, [0, 0.0])
                r_[0] += 1
                r_[1] += statistics.median(ts)
            print(f"--- [{s}] per launch (alone, median of 7), sum {sum(v[1] for v in rows.values()):.1f} us")
            for key, (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
                print(f"  x{n:2d} {t / n:7.1f} us each  {t:7.1f} us  {key}")


if __name__ == "__main__":
    main()
