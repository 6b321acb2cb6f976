#!/usr/bin/env python3
"""Per-launch time of the detector plan (events around each launch), with the tuned conv variants."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptiveisp_amd.yolo import YoloEngine, yolov3
from adaptiveisp_amd.yolo import _lib
torch.manual_seed(1)
eng = YoloEngine(yolov3().eval(), 8, 720, 1280)
eng.autotune(cache=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'adaptiveisp_amd', 'yolo', 'tuning', 'mi355x.json'), retune='--retune' in sys.argv)
x = torch.rand(8, 3, 720, 1280, device="cuda")
eng(x); torch.cuda.synchronize()
st = _lib.stream_ptr()
rows = []
for kind, fn, args in eng.plan:
    if kind == "stem":
        continue
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(*args, st); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    t = sorted(ts)[2] * 1e3
    if kind == "conv2":                                   # fused 3x3 + next 1x1 (adayolo_conv_fused1x1_fwd)
        B, H, W, cin, cout, k, s = args[8:15]
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        fl = 2.0 * B * Ho * Wo * cout * (k * k * cin + args[20])
        rows.append((t, f"conv {H}x{W} {cin}->{cout} k{k}s{s} + 1x1 {cout}->{args[20]} fused: {t:7.1f} us {fl / t / 1e6:7.1f} TF"))
    elif kind == "chain":                                 # a run of layers as one persistent launch (YoloEngine.fuse_chains)
        c = next(c for c in eng.chains if c["ws"].data_ptr() == args[2].value)
        rows.append((t, f"chain of {c['layers']} layers ({c['flops'] / 1e9:.0f} GFLOP): {t:7.1f} us {c['flops'] / t / 1e6:7.1f} TF"))
    elif kind == "bneckws":                               # a whole Bottleneck of the C = 64 / 128 stages in one launch
        B, H, W, C = args[8:12]
        fl = 2.0 * B * H * W * (C * (C // 2) + 9 * (C // 2) * C)
        rows.append((t, f"bottleneck {H}x{W} C{C} whole-block: {t:7.1f} us {fl / t / 1e6:7.1f} TF  {B * H * W * C * 4 / t / 1e3:6.0f} GB/s"))
    elif kind == "k1":                                    # a 1x1 layer on the whole-K kernel (YoloEngine.fuse_k1)
        B, H, W, cin, cout = args[6:11]
        fl = 2.0 * B * H * W * cin * cout
        rows.append((t, f"conv {H}x{W} {cin}->{cout} k1s1 whole-K: {t:7.1f} us {fl / t / 1e6:7.1f} TF  {B * H * W * (cin + cout) * 2 / t / 1e3:6.0f} GB/s"))
    elif kind == "conv":
        B, H, W, cin, cout, k, s, act, v = args[8:17]
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        fl = 2.0 * B * Ho * Wo * cout * k * k * cin
        byt = (B * H * W * cin + B * Ho * Wo * cout) * 2
        rows.append((t, f"conv {H}x{W} {cin}->{cout} k{k}s{s} v{v}: {t:7.1f} us {fl / t / 1e6:7.1f} TF  {byt / t / 1e3:6.0f} GB/s"))
    else:
        rows.append((t, f"{kind}: {t:7.1f} us"))
tot = sum(r[0] for r in rows)
agg = {}
for t, s in rows:
    key = s.split(":")[0]
    a = agg.setdefault(key, [0, 0.0, s]); a[0] += 1; a[1] += t
for key, (n, t, s) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"x{n:2d} total {t:7.1f} us ({100 * t / tot:4.1f}%)  | {s}")
print("sum of launches:", tot, "us")
