#!/usr/bin/env python3
"""Interleaved A/B in ONE process: the tuned detector at 8 x 720 x 1280 with its runs of 256 x 256-kernel layers as persistent
chains (ADAYOLO_CHAIN=1, YoloEngine.fuse_chains) against the same plan with one launch per layer (ADAYOLO_CHAIN=0).
  * whole forward, each as a replayed hipGraph, `rounds` x `reps` replays interleaved: median / min ms;
  * every chain alone against its own layers launched one by one (event pairs, same buffers);
  * the predictions must be bit-identical.
Usage (GPU box): python tools/chain_ab.py [--rounds 12] [--reps 5] [--batch 8]"""
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
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--width", type=int, default=1280)
    a = ap.parse_args()
    from _synth import synth_yolo_state_dict, test_image
    from adaptiveisp_amd.yolo import YoloEngine, _lib, yolov3
    dev = torch.device("cuda:0")
    tune = os.path.join(ROOT, "adaptiveisp_amd", "yolo", "tuning", "mi355x.json")
    det = yolov3()
    det.load_state_dict(synth_yolo_state_dict(det, seed=2))
    det = det.eval()
    x = torch.from_numpy(test_image(a.batch, a.height, a.width, seed=3, special=False)).to(dev)
    engines = {}
    for flag in ("0", "1"):
        os.environ["ADAYOLO_CHAIN"] = flag
        e = YoloEngine(det, a.batch, a.height, a.width, device=dev)
        e.autotune(cache=tune, write=False)
        engines[flag] = e
    plain, chained = engines["0"], engines["1"]
    print(f"plan: {len(plain.plan)} launches without chains, {len(chained.plan)} with; chains: "
          f"{[(c['layers'], round(c['flops'] / 1e9, 1)) for c in chained.chains]} (layers, GFLOP)")
    want = plain(x).clone()
    got = chained(x).clone()
    torch.cuda.synchronize()
    print("bit-identical:", torch.equal(want, got), " chain status:", chained.chain_status())
    graphs = {}
    for k, e in engines.items():
        e(x)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            e(x)
        graphs[k] = g
    times = {"0": [], "1": []}
    for r in range(a.rounds):
        for k in (("0", "1") if r % 2 == 0 else ("1", "0")):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            graphs[k].replay()
            e0.record()
            for _ in range(a.reps):
                graphs[k].replay()
            e1.record()
            torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) / a.reps)
    for k, name in (("0", "one launch per layer"), ("1", "persistent chains  ")):
        t = times[k]
        print(f"detector forward, {name}: median {statistics.median(t):.4f} ms  min {min(t):.4f}  max {max(t):.4f}")
    print(f"chains / launches (median): {statistics.median(times['1']) / statistics.median(times['0']):.4f}")
    # every chain alone against its own layers
    st = _lib.stream_ptr()
    for ci, c in enumerate(chained.chains):
        kind, fn, args = next(p for p in chained.plan if p[0] == "chain" and p[2][2].value == c["ws"].data_ptr())

        def run_chain():
            _lib.check(fn(*args, st), "chain")

        def run_layers():
            for _, f, ar in c["entries"]:
                _lib.check(f(*ar, st), "layer")
        res = {}
        for name, call in (("chain", run_chain), ("layers", run_layers)):
            res[name] = []
        for r in range(a.rounds):
            for name, call in ((("layers", run_layers), ("chain", run_chain)) if r % 2 == 0 else (("chain", run_chain), ("layers", run_layers))):
                call()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    call()
                e1.record()
                torch.cuda.synchronize()
                res[name].append(e0.elapsed_time(e1) / a.reps * 1e3)
        ml, mc = statistics.median(res["layers"]), statistics.median(res["chain"])
        print(f"chain {ci}: {c['layers']} layers, {c['flops'] / 1e9:.1f} GFLOP: separate launches {ml:.1f} us ({c['flops'] / ml / 1e6:.0f} TFLOP/s), "
              f"chain {mc:.1f} us ({c['flops'] / mc / 1e6:.0f} TFLOP/s), ratio {mc / ml:.4f}; status {chained.chain_status()}")


if __name__ == "__main__":
    main()
