#!/usr/bin/env python3
"""Which kernel of the ISP episode is not bit-reproducible beside the detector: the policy trunk (per layer), fc1, or NLM."""
import argparse, ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from adaptiveisp_amd import _lib
from adaptiveisp_amd.config import cfg
a = argparse.Namespace(batch=8, height=720, width=1280, schedule="mixed", retune=False)
step, engine, agent, x0, sched = bench.build_workload(a, torch.device("cuda:0"))
xref = step.isp_chain().clone()
side = torch.cuda.Stream()
fast = agent._fast
L = fast.L
B = a.batch
st0 = torch.rand(B, cfg.num_state_dim, device="cuda:0")
pooled = _lib.pool64(x0)
fast.run(pooled, torch.rand(B, cfg.z_dim, device="cuda:0"), st0, 1.0, 0)
bufs = fast._buffers(B, x0.device)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None

def trunk(upto=4):
    sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    src, size, cin = pooled, 64, 3 + st0.shape[1]
    for li, (w, b) in enumerate(fast.layers[:upto]):
        L.adaisp_policy_conv(P(src), P(st0) if li == 0 else None, st0.shape[1] if li == 0 else 0, P(w), P(b), P(bufs["acts"][li]), 2, B,
                             cin, size, w.shape[1], sp)
        src, size, cin = bufs["acts"][li], size // 2, w.shape[1]
    torch.cuda.synchronize()
    return [t.clone() for t in bufs["acts"][:upto]]

def disturbed(fn, n=30, pre=None):
    ref = fn()
    bad = [0] * (len(ref) if isinstance(ref, list) else 1)
    for i in range(n):
        with torch.cuda.stream(side), torch.no_grad():
            engine(xref)
        if pre is not None:
            pre()
        out = fn()
        if isinstance(ref, list):
            for k, (o, r) in enumerate(zip(out, ref)):
                bad[k] += not torch.equal(o, r)
        else:
            bad[0] += not torch.equal(out, ref)
    return bad

print("trunk layers beside the detector, differing runs per layer:", disturbed(trunk))
print("  ... with NLM launched just before on the same stream:", disturbed(trunk, pre=lambda: _lib.process(4, x0, torch.full((B, 1), 0.1, device="cuda:0"), clip=True)))
h = torch.full((B, 1), 0.1, device="cuda:0")
def nlm():
    y = _lib.process(4, x0, h, clip=True); torch.cuda.synchronize(); return y
print("NLM beside the detector:", disturbed(nlm, 10))
def nlm_v1():
    y = _lib.process(4, x0, h, clip=True, nlm_v1=True); torch.cuda.synchronize(); return y
print("NLM (compiled form) beside the detector:", disturbed(nlm_v1, 10))
def chain():
    y = step.isp_chain(); torch.cuda.synchronize(); return y.clone()
print("whole episode beside the detector:", disturbed(chain, 10))
