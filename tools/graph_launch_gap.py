#!/usr/bin/env python3
"""What a hipGraph boundary costs on this runtime: graphs of N small dependent kernels (one stream, or with a second branch) replayed
back to back with the host far ahead; per replay: wall time, N x the kernel's in-graph time, and the rest = the gap a replay adds.
Measurement aid for the captured RL iteration. usage (GPU box): python tools/graph_launch_gap.py"""
import time

import torch

dev = torch.device("cuda:0")
x = torch.zeros(1 << 16, device=dev)
y = torch.zeros(1 << 16, device=dev)
side = torch.cuda.Stream()


def build(n, branch):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        if branch:
            cur = torch.cuda.current_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for _ in range(n // 4):
                    y.add_(1.0)
        for _ in range(n):
            x.add_(1.0)
        if branch:
            torch.cuda.current_stream().wait_stream(side)
    return g


def per_replay(g, reps=60):
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6, t_host / reps * 1e6


for branch in (False, True):
    base = None
    for n in (20, 85, 340, 1360):
        g = build(n, branch)
        us, host = per_replay(g)
        if base is None:
            base = (n, us)
        per_kernel = (us - base[1]) / (n - base[0]) if n != base[0] else float("nan")
        print(f"{'two branches' if branch else 'one stream  '} N = {n:5d}: {us:8.1f} us per replay (host {host:7.1f} us in replay()), "
              f"{us / n:6.2f} us per node; slope against N = {base[0]}: {per_kernel:5.2f} us per node", flush=True)
# the same kernels launched one by one (host ahead): the per-kernel floor without a graph
for n in (340,):
    for _ in range(3):
        for _ in range(n):
            x.add_(1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        for _ in range(n):
            x.add_(1.0)
    torch.cuda.synchronize()
    print(f"eager        N = {n:5d}: {(time.perf_counter() - t0) / 10 * 1e6:8.1f} us per {n} launches")
# ---- GPU-bound replays (the host far ahead): ~20 us kernels. What does a fork / join inside the graph add per replay? --------------
big = torch.zeros(12 << 20, device=dev)


def build_big(n, nside, where):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        for i in range(n):
            if nside and i == where:
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    for _ in range(nside):
                        y.add_(1.0)
            big.add_(1.0)
        if nside:
            cur.wait_stream(side)
    return g


rows = []
for label, nside, where in (("one stream", 0, 0), ("side branch of 40 small nodes forked at node 170", 40, 170), ("forked at node 0", 40, 0),
                            ("forked at node 330", 40, 330)):
    g = build_big(340, nside, where)
    us, host = per_replay(g, reps=30)
    rows.append(us)
    print(f"GPU-bound, 340 nodes of ~{rows[0] / 340:.1f} us, {label}: {us:8.1f} us per replay (host {host:6.1f} us), + {us - rows[0]:6.1f} us against one stream", flush=True)
