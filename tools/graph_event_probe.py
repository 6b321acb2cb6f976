#!/usr/bin/env python3
"""Do timing events recorded INSIDE a captured hipGraph (torch.cuda.Event(enable_timing=True, external=True) -> event-record
nodes) give per-kernel durations on replay? If so bench.py can time the conv kernels inside the pipelined graph itself."""
import torch
x = torch.rand(8, 3, 720, 1280, device="cuda")
y = torch.empty_like(x)
def work():
    torch.mul(x, 1.5, out=y)
for ext in (True, False):
    try:
        evs = [torch.cuda.Event(enable_timing=True, external=ext) if ext else torch.cuda.Event(enable_timing=True) for _ in range(4)]
        work(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            evs[0].record(); work(); evs[1].record(); work(); work(); evs[2].record()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        print(f"external={ext}: one mul {evs[0].elapsed_time(evs[1]) * 1e3:.1f} us, two muls {evs[1].elapsed_time(evs[2]) * 1e3:.1f} us")
    except Exception as e:
        print(f"external={ext}: {type(e).__name__}: {e}")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); work(); e1.record(); torch.cuda.synchronize()
print(f"eager reference: one mul {e0.elapsed_time(e1) * 1e3:.1f} us")
