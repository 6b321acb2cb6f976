#!/usr/bin/env python3
"""Shader clock while the detector runs (is the in-graph inflation of the conv kernels clock?).

A one-wave probe kernel on a side stream samples (s_memrealtime @100 MHz, s_memtime = shader cycles) every ~4 us
(tools/clock_probe.hip); stamp kernels on the main stream mark each workload window in the same time base. Reported per
window: kernel/graph wall time, the median / min / max clock over the probe samples inside it, and the probe's own
fixed VALU chain in cycles (contention check). Windows:
  idle            nothing but the probe
  pp from idle    ONE launch of the dominant conv (each pp layer shape of the network) after 3 ms of idle chip
  pp back-to-back the same launch 200 times back to back
  detector graph  200 replays of the whole detector forward (hipGraph), first 10 and last 100 reported separately
Usage: python3 tools/clock_trace.py [> profiles/round3_clock.txt]"""
import ctypes
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adaptiveisp_amd.yolo import YoloEngine, yolov3, _lib  # noqa: E402

SO = os.path.join(ROOT, "tools", "libclockprobe.so")
if not os.path.exists(SO):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", os.path.join(ROOT, "tools", "clock_probe.hip"), "-o", SO])
P = ctypes.CDLL(SO)
P.clockprobe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
P.clockprobe_stamp.argtypes = [ctypes.c_void_p, ctypes.c_void_p]

dev = torch.device("cuda:0")
probe_stream = torch.cuda.Stream()
SAMPLES_MAX = 400000


def run_window(name, work, n_samples, sleeps=1):
    """Start the probe (n_samples x ~(1 + sleeps x 3.4) us), stamp, run work() on the current stream, stamp; return stats."""
    log = torch.zeros(3 * n_samples, dtype=torch.int64, device=dev)
    st = torch.zeros(4, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    P.clockprobe_launch(log.data_ptr(), n_samples, sleeps, ctypes.c_void_p(probe_stream.cuda_stream))
    time.sleep(0.003)                                            # chip idle (but for the probe) before the window
    cur = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P.clockprobe_stamp(st.data_ptr(), cur)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    work()
    e1.record()
    P.clockprobe_stamp(st.data_ptr() + 16, cur)
    torch.cuda.synchronize()
    lg = log.cpu().numpy().reshape(-1, 3).astype(np.float64)
    s = st.cpu().numpy().astype(np.float64)
    rt, ct, chain = lg[:, 0], lg[:, 1], lg[:, 2]
    ok = rt > 0
    rt, ct, chain = rt[ok], ct[ok], chain[ok]
    clk = np.diff(ct) / np.diff(rt) * 100.0                       # MHz between consecutive samples
    mid = 0.5 * (rt[1:] + rt[:-1])
    return dict(name=name, ms=e0.elapsed_time(e1), t0=s[0], t1=s[2], mid=mid, clk=clk, chain=chain[1:],
                window_ticks=(s[3] - s[1]), window_rt=(s[2] - s[0]))


def summarize(r, lo=0.0, hi=1.0, label=None):
    a = r["t0"] + lo * (r["t1"] - r["t0"])
    b = r["t0"] + hi * (r["t1"] - r["t0"])
    m = (r["mid"] >= a) & (r["mid"] <= b)
    pre = r["mid"] < r["t0"]
    def f(x):
        return f"{np.median(x):7.0f} [{x.min():5.0f} .. {x.max():5.0f}]" if x.size else "   (no sample)"
    print(f"{label or r['name']:58s} wall {r['ms'] * (hi - lo):9.3f} ms | clock MHz median [min .. max] inside {f(r['clk'][m])} "
          f"(n={int(m.sum())}) | before (idle) {f(r['clk'][pre])} | probe chain cycles inside {np.median(r['chain'][m]) if m.any() else float('nan'):.0f} "
          f"before {np.median(r['chain'][pre]) if pre.any() else float('nan'):.0f}", flush=True)


def main():
    torch.manual_seed(1)
    det = yolov3().eval()
    B, H, W = 8, 720, 1280
    eng = YoloEngine(det, B, H, W, device=dev)
    eng.autotune(cache=os.path.join(ROOT, "adaptiveisp_amd", "yolo", "tuning", "mi355x.json"))
    g = torch.Generator(device="cpu").manual_seed(1235)
    x = (torch.rand(B, 3, H, W, generator=g) ** 2.2 * 0.5).to(dev)
    with torch.no_grad():
        eng(x)
    torch.cuda.synchronize()
    st = _lib.stream_ptr

    print("# shader clock = d(s_memtime)/d(s_memrealtime) x 100 MHz, one-wave probe on a side stream (tools/clock_probe.hip)")
    r = run_window("idle (probe only, 2 ms)", lambda: time.sleep(0.002) or torch.cuda.synchronize(), 2000)
    summarize(r)
    # the window stamps: ticks of s_memtime over the window / its real time = the clock seen by the stamp kernels themselves
    seen = set()
    for kind, fn, args in eng.plan:
        if kind not in ("conv", "conv2"):
            continue
        v = 58 if kind == "conv2" else args[16]
        if v not in (50, 58, 60):
            continue
        key = (kind, tuple(args[8:16]))
        if key in seen:
            continue
        seen.add(key)
        Bc, Hc, Wc, cin, cout, k, s_ = args[8:15]
        shape = f"{'fused pair ' if kind == 'conv2' else ''}variant {v} {cin}->{cout} k{k} s{s_} @{Hc}x{Wc}"
        one = lambda fn=fn, args=args: fn(*args, st())          # noqa: E731
        for _ in range(3):
            one()
        torch.cuda.synchronize()
        r1 = run_window(f"{shape}: ONE launch from an idle chip", one, 1500)
        summarize(r1)
        r2 = run_window(f"{shape}: 200 launches back to back", lambda: [one() for _ in range(200)], 12000)
        summarize(r2, 0.0, 1.0, label=f"{shape}: 200 back to back (per launch {r2['ms'] / 200 * 1e3:.1f} us)")
    # the whole detector as a graph
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph), torch.no_grad():
        eng(x)
    graph.replay()
    torch.cuda.synchronize()
    r1 = run_window("detector graph: ONE replay from an idle chip", graph.replay, 3000)
    summarize(r1)
    n = 200
    r3 = run_window(f"detector graph: {n} replays back to back", lambda: [graph.replay() for _ in range(n)], 250000)
    print(f"# {n} replays: {r3['ms'] / n:.3f} ms per replay")
    summarize(r3, 0.0, 0.05, label="detector graph: replays 1-10 of 200")
    summarize(r3, 0.05, 0.5, label="detector graph: replays 11-100 of 200")
    summarize(r3, 0.5, 1.0, label="detector graph: replays 101-200 of 200")
    c = r3["clk"][(r3["mid"] >= r3["t0"]) & (r3["mid"] <= r3["t1"])]
    print("# detector graph, all 200 replays: clock percentiles MHz  p5 %.0f  p10 %.0f  p25 %.0f  p50 %.0f  p75 %.0f  p90 %.0f" %
          tuple(np.percentile(c, q) for q in (5, 10, 25, 50, 75, 90)))
    # rocm-smi view, if this user can read it
    for cmd in (["rocm-smi", "--showclocks"], ["amd-smi", "metric", "--clock"]):
        try:
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=20)
            print(f"# {' '.join(cmd)} (rc {out.returncode}):")
            for ln in (out.stdout or out.stderr).splitlines()[:25]:
                print("#   " + ln)
            break
        except Exception as e:
            print(f"# {' '.join(cmd)}: {type(e).__name__}: {e}")


if __name__ == "__main__":
    main()
