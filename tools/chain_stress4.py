#!/usr/bin/env python3
"""k_fc1 under a concurrent detector: which elements are wrong, and is the input (act3) stale when it runs?"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from adaptiveisp_amd import _lib
from adaptiveisp_amd.config import cfg
a = argparse.Namespace(batch=8, height=720, width=1280, schedule="mixed", retune=False)
step, engine, agent, x0, sched = bench.build_workload(a, torch.device("cuda:0"))
xref = step.isp_chain().clone()
side = torch.cuda.Stream()
z = torch.rand(a.batch, cfg.z_dim, device="cuda:0")
st = torch.zeros(a.batch, cfg.num_state_dim, device="cuda:0")
fast = agent._fast
pooled = _lib.pool64(x0)
fast.run(pooled, z, st, 1.0, 0); torch.cuda.synchronize()
bufs = fast._buffers(a.batch, x0.device)
feats = bufs["acts"][-1].clone()                       # [2][B][4096]
F = len(agent.filters)
# exact reference of the hidden layer from the (final) features, fp64
W1, B1 = fast.w1.double(), fast.b1.double()            # [F+1][128][4096], [F+1][128]
src = fast.head_src.long()
fe = feats.reshape(2, a.batch, -1).double()
hid_ref = torch.stack([torch.nn.functional.leaky_relu(fe[src[h]] @ W1[h].T + B1[h], 0.2) for h in range(F + 1)], 1)   # [B][F+1][128]
stats = {}
for i in range(40):
    with torch.cuda.stream(side), torch.no_grad():
        engine(xref)
    fast.run(pooled, z, st, 1.0, 0); torch.cuda.synchronize()
    h = bufs["hidden"].double()
    err = (h - hid_ref).abs()
    badmask = err > 1e-4
    if badmask.any():
        idx = badmask.nonzero()
        key = (tuple(sorted(set(idx[:, 0].tolist()))), tuple(sorted(set(idx[:, 1].tolist())))[:6], len(idx))
        stats[key] = stats.get(key, 0) + 1
print("wrong elements (batch rows, heads, count) -> occurrences:")
for k, v in list(stats.items())[:12]:
    print("  ", k, v)
print("runs with a wrong hidden element:", sum(stats.values()), "of 40; max err of a clean run:",
      float((bufs['hidden'].double() - hid_ref).abs().max()))
