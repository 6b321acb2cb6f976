#!/usr/bin/env python3
"""Which policy-step tensor differs when the detector runs on a second stream?"""
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
def policy():
    pooled = _lib.pool64(x0)
    o = fast.run(pooled, z, st, 1.0, 0)
    bufs = fast._buffers(a.batch, x0.device)
    d = {k: v.clone() for k, v in o.items() if isinstance(v, torch.Tensor)}
    d["pooled"] = pooled.clone()
    for i, t in enumerate(bufs["acts"]):
        d[f"act{i}"] = t.clone()
    d["hidden"] = bufs["hidden"].clone()
    return d
ref = policy(); torch.cuda.synchronize()
counts = {}
for i in range(60):
    with torch.cuda.stream(side), torch.no_grad():
        engine(xref)
    cur = policy(); torch.cuda.synchronize()
    for k in ("pooled", "act0", "act1", "act2", "act3", "hidden", "params_all", "packed", "op_ids", "pdf", "new_states", "penalty"):
        if not torch.equal(cur[k], ref[k]):
            dd = (cur[k].float() - ref[k].float()).abs()
            counts.setdefault(k, []).append((int((dd > 0).sum()), float(dd.max())))
            break
print({k: (len(v), v[:3]) for k, v in counts.items()})
