#!/usr/bin/env python3
"""The whole-K 1x1 kernel (adayolo_conv1x1_stream_fwd) against every ring kernel that serves the shape, interleaved in one
process, rotating over enough buffer sets that no launch finds its operands in a cache it would not find them in inside the
network (a 1x1 layer's input was written by the launch before it: sets = 2 keeps it Infinity-Cache warm, sets = 24 cold).
usage (GPU box): python tools/k1_bench.py [--sets 2]"""
import argparse
import ctypes
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adaptiveisp_amd.yolo import _lib  # noqa: E402

SHAPES = [(8, 46, 80, 512, 256), (8, 92, 160, 256, 256), (8, 23, 40, 512, 256)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sets", type=int, default=2)
    ap.add_argument("--reps", type=int, default=40)
    a = ap.parse_args()
    L = _lib.load()
    vp = ctypes.c_void_p
    for B, H, W, cin, cout in SHAPES:
        g = torch.Generator(device="cpu").manual_seed(1)
        xs = [torch.randn(B, H, W, cin, generator=g).to(torch.bfloat16).cuda() for _ in range(a.sets)]
        outs = [torch.empty(B, H, W, cout, dtype=torch.bfloat16, device="cuda") for _ in range(a.sets)]
        w = (torch.randn(cout, cin, generator=g) / cin ** 0.5).to(torch.bfloat16).cuda()
        wp = w.reshape(cout // 32, 32, cin // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous()
        b = torch.randn(cout, generator=g).cuda()
        fl = 2.0 * B * H * W * cin * cout

        def ring(v):
            def f(i):
                return L.adayolo_conv_fwd_variant(vp(xs[i].data_ptr()), cin, vp(w.data_ptr()), vp(b.data_ptr()), None, 0,
                                                  vp(outs[i].data_ptr()), cout, B, H, W, cin, cout, 1, 1, 1, v, _lib.stream_ptr())
            return f

        def k1(i):
            return L.adayolo_conv1x1_stream_fwd(vp(xs[i].data_ptr()), cin, vp(wp.data_ptr()), vp(b.data_ptr()), vp(outs[i].data_ptr()),
                                                cout, B, H, W, cin, cout, 1, _lib.stream_ptr())
        cands = {"k1": k1, "v60": ring(60), "v50": ring(50), "v80": ring(80), "v85": ring(85), "v22": ring(22)}
        times = {k: [] for k in cands}
        for k, f in cands.items():
            assert f(0) == 0, k
        torch.cuda.synchronize()
        for rnd in range(5):
            for k, f in cands.items():
                # `reps` launches captured into one graph (what the engine's forward is): back-to-back, no host gaps
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    for j in range(a.reps):
                        f(j % a.sets)
                gr.replay()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                gr.replay()
                e1.record()
                torch.cuda.synchronize()
                times[k].append(e0.elapsed_time(e1) / a.reps * 1e3)
        print(f"{B}x{H}x{W} {cin}->{cout}, sets {a.sets}: " +
              "  ".join(f"{k} {statistics.median(v):.1f} us ({fl / statistics.median(v) / 1e6:.0f} TF)" for k, v in times.items()), flush=True)


if __name__ == "__main__":
    main()
