#!/usr/bin/env python3
"""Is the ISP chain bit-reproducible (a) replayed alone from a graph, (b) eagerly beside a busy second stream?"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
a = argparse.Namespace(batch=8, height=720, width=1280, schedule="mixed", retune=False)
step, engine, agent, x0, sched = bench.build_workload(a, torch.device("cuda:0"))
xref = step.isp_chain().clone()
torch.cuda.synchronize()
# (a) graph of the chain alone
out = torch.empty_like(x0)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out.copy_(step.isp_chain())
bad = 0
for i in range(40):
    g.replay(); torch.cuda.synchronize()
    bad += not torch.equal(out, xref)
print("graph, chain alone:", bad, "mismatches of 40")
# (b) eager chain while another stream runs the detector
side = torch.cuda.Stream()
bad = 0
for i in range(40):
    with torch.cuda.stream(side), torch.no_grad():
        engine(xref)
    x = step.isp_chain()
    torch.cuda.synchronize()
    bad += not torch.equal(x, xref)
print("eager chain beside the detector on a second stream:", bad, "mismatches of 40")
# (c) per-step: which RL step first differs? (teacher-forced schedule, eager, beside the detector)
from adaptiveisp_amd.config import cfg
z = torch.rand(a.batch, cfg.z_dim, device="cuda:0")
def chain_steps():
    x, st, outs = x0, torch.zeros(a.batch, cfg.num_state_dim, device="cuda:0"), []
    with torch.no_grad():
        for k in sched:
            (x, st, _, _), dbg, _ = agent((x, z, st), 1.0, selected_filter_id=k)
            outs.append((x.clone(), dbg["filter_debug_info"][k]["filter_parameters"].clone()))
    return outs
ref = chain_steps(); torch.cuda.synchronize()
first = {}
for i in range(40):
    with torch.cuda.stream(side), torch.no_grad():
        engine(xref)
    cur = chain_steps(); torch.cuda.synchronize()
    for s, ((x, p), (xr, pr)) in enumerate(zip(cur, ref)):
        if not torch.equal(p, pr) or not torch.equal(x, xr):
            key = (s, "param" if not torch.equal(p, pr) else "image")
            first[key] = first.get(key, 0) + 1
            break
print("first differing (step, what) over 40 runs:", first)
