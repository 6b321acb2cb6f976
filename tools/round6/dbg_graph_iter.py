import copy, os, sys, types
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import test_gpu_train_graph as T
from _synth import test_image
from adaptiveisp_amd.rl import train_iteration
from adaptiveisp_amd.train import _GraphIteration
DEV = "cuda:0"
B, H, W = 4, 64, 96
eng, loss_fn = T._detector(B, H, W)
cfg, agent, value = T._fresh(B)
opts = [torch.optim.Adam(agent.parameters(), lr=3e-5, fused=True), torch.optim.Adam(value.parameters(), lr=3e-5, fused=True)]
names = [("agent." + n) for n, _ in agent.named_parameters()] + [("value." + n) for n, _ in value.named_parameters()]
def feed(i):
    return dict(im=torch.from_numpy(test_image(B, H, W, seed=30 + i, special=False)).to(DEV),
                z=torch.full((B, cfg.z_dim), 0.2 + 0.2 * i, device=DEV), state=torch.zeros(B, cfg.num_state_dim, device=DEV), label=T._labels(B, i))
for i in range(2):
    f = feed(i)
    train_iteration(cfg, agent, value, eng, loss_fn, f["im"], f["z"], f["state"], f["label"], 0.1, opts)
torch.cuda.synchronize()
snap = copy.deepcopy((agent.state_dict(), value.state_dict(), opts[0].state_dict(), opts[1].state_dict()))
def restore():
    agent.load_state_dict(snap[0]); value.load_state_dict(snap[1])
    opts[0].load_state_dict(copy.deepcopy(snap[2])); opts[1].load_state_dict(copy.deepcopy(snap[3]))
def P():
    return [p.detach().clone() for p in list(agent.parameters()) + list(value.parameters())]
def bufs():
    return {k: v.detach().clone() for m, pre in ((agent, "agent."), (value, "value.")) for k, v in m.named_buffers(prefix=pre)}
sched = [(0.25, 3e-5, 3e-4), (0.75, 1e-5, 1e-4)]
def ordinary(n=2):
    restore(); outs = []; ps = []
    for i, (prog, lra, lrv) in enumerate(sched[:n]):
        f = feed(2 + i)
        opts[0].param_groups[0]["lr"], opts[1].param_groups[0]["lr"] = lra, lrv
        out = train_iteration(cfg, agent, value, eng, loss_fn, f["im"], f["z"], f["state"], f["label"], prog, opts)
        torch.cuda.synchronize()
        outs.append({k: out[k].detach().clone() for k in ("value_loss", "agent_loss", "reward")}); ps.append(P())
    return outs, ps, bufs()
o1, p1, b1 = ordinary()
o2, p2, b2 = ordinary()
p0 = None
restore(); p0 = P()
def cmp(tag, A, Bp):
    d = [(float((a - b).abs().max()), n) for a, b, n in zip(A, Bp, names)]
    d.sort(reverse=True)
    print(tag, "max diff", d[:4], flush=True)
cmp("ordinary vs ordinary after it1:", p1[0], p2[0]); cmp("ordinary vs ordinary after it2:", p1[1], p2[1])
cmp("step size it1 (ordinary vs start):", p1[0], p0)
print("ordinary value_loss", [float(o["value_loss"]) for o in o1], [float(o["value_loss"]) for o in o2])
restore()
tr = types.SimpleNamespace(cfg=cfg, agent=agent, value=value, detector=eng, loss_fn=loss_fn, batch_size=B, max_bri=0.9,
                           use_truncated=True, agent_optimizer=opts[0], value_optimizer=opts[1], buckets=None)
G = _GraphIteration(tr, feed(2), cap=256)
gp = []; go = []
for i, (prog, lra, lrv) in enumerate(sched):
    f = feed(2 + i)
    assert G.tables.fill(f["label"])
    G.set_scalars((1.0 - prog) * cfg.exploration_penalty, lra, lrv)
    G.im.copy_(f["im"]); G.z.copy_(f["z"]); G.state.copy_(f["state"]); G.tables.upload()
    if G.graph is None:
        G.capture()
    G.graph.replay(); G.wait_guard(30.0); torch.cuda.synchronize()
    gp.append(P()); go.append({k: float(G.out[k]) if G.out[k].numel() == 1 else float(G.out[k].mean()) for k in ("value_loss", "agent_loss", "reward")})
gb = bufs()
cmp("graph vs ordinary after it1:", gp[0], p1[0]); cmp("graph vs ordinary after it2:", gp[1], p1[1])
print("graph", go)
bd = sorted(((float((gb[k].float() - b1[k].float()).abs().max()), k) for k in gb), reverse=True)[:5]
print("buffers graph vs ordinary", bd)
bd = sorted(((float((b2[k].float() - b1[k].float()).abs().max()), k) for k in gb), reverse=True)[:5]
print("buffers ordinary vs ordinary", bd)
print("adam steps", sorted({float(s["step"]) for s in opts[1].state.values()}))
