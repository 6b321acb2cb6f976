"""World-size-2 data-parallel gradient exchange on CPU (gloo): the N>1 path of the RL training step."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from adaptiveisp_amd import dist as adist
    from adaptiveisp_amd.config import cfg
    from adaptiveisp_amd.value import Value
    torch.set_num_threads(1)
    r, w, dev = adist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                       # ranks start different ...
    v = Value(cfg, shape=(19, 64, 64))
    adist.broadcast_parameters([v])                     # ... and are made identical
    flat0 = torch.cat([p.detach().reshape(-1) for p in v.parameters()])
    gathered = [torch.zeros_like(flat0) for _ in range(world)]
    dist.all_gather(gathered, flat0)
    assert all(torch.equal(gathered[0], g) for g in gathered)
    # rank-dependent synthetic gradients: after the bucketed all-reduce every rank holds their mean
    bucket = adist.GradBucket(v)
    for i, p in enumerate(v.parameters()):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    opt = torch.optim.SGD(v.parameters(), lr=1.0)
    before = [p.detach().clone() for p in v.parameters()]
    adist.synced_step([v], [opt], [bucket], max_grad_norm=1e9)
    mean = sum(range(1, world + 1)) / world
    for i, (p, b) in enumerate(zip(v.parameters(), before)):
        torch.testing.assert_close(b - p.detach(), torch.full_like(p, mean * (i + 1)))
    flat1 = torch.cat([p.detach().reshape(-1) for p in v.parameters()])
    gathered = [torch.zeros_like(flat1) for _ in range(world)]
    dist.all_gather(gathered, flat1)
    assert all(torch.equal(gathered[0], g) for g in gathered)        # replicas stay in lock-step
    # the clip sees the GLOBAL gradient: norm of the averaged bucket, identical on every rank
    for i, p in enumerate(v.parameters()):
        p.grad = torch.full_like(p, float(rank + 1))
    adist.synced_step([v], [torch.optim.SGD(v.parameters(), lr=0.0)], [bucket], max_grad_norm=1e-5)
    out.put((rank, bucket.numel))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_all_reduce():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(out.get(timeout=5) for _ in range(2))
    assert got == [(0, 1223841), (1, 1223841)]


def test_single_process_is_a_no_op():
    from adaptiveisp_amd import dist as adist
    lin = torch.nn.Linear(4, 2)
    lin.weight.grad = torch.ones_like(lin.weight)
    lin.bias.grad = torch.ones_like(lin.bias)
    b = adist.GradBucket(lin)
    b.finish(b.all_reduce_mean())
    assert torch.equal(lin.weight.grad, torch.ones_like(lin.weight))


# ---- two ranks run the whole RL optimisation step on DIFFERENT batches and stay in lock-step ---------------------

class _StubDetector(torch.nn.Module):
    """Frozen stand-in for the reward model: three raw head maps [B, 3, ny, nx, 5+nc] with autograd to the image."""

    def __init__(self, nc=4):
        super().__init__()
        self.no = 5 + nc
        self.convs = torch.nn.ModuleList([torch.nn.Conv2d(3, 3 * self.no, 3, stride=s, padding=1) for s in (8, 16, 32)])
        for p in self.parameters():
            p.requires_grad_(False)

    def forward(self, x):
        outs = []
        for c in self.convs:
            y = c(x)
            B, _, ny, nx = y.shape
            outs.append(y.view(B, 3, self.no, ny, nx).permute(0, 1, 3, 4, 2).contiguous())
        return outs


def _torch_isp(img, packed, op_ids):
    """Differentiable stand-in for the HIP filters (this test is about the data-parallel step, not pixels): a gain and
    an offset driven by the first two packed parameters."""
    g = torch.exp(packed[:, 0].clamp(-2, 2) * 0.1)[:, None, None, None]
    o = (packed[:, 1].clamp(-2, 2) * 0.01)[:, None, None, None]
    return torch.clip(img * g + o, 0.0, 1.0)


def _rl_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from adaptiveisp_amd import dist as adist
    from adaptiveisp_amd.agent import Agent
    from adaptiveisp_amd.config import cfg
    from adaptiveisp_amd.rl import train_iteration
    from adaptiveisp_amd.value import Value
    from adaptiveisp_amd.yolo.loss import DetectionLoss, default_hyp
    torch.set_num_threads(1)
    adist.init_from_env("gloo")
    torch.manual_seed(7 + rank)                         # different initial weights, batches and noise per rank
    agent = Agent(cfg, shape=(6 + len(cfg.filters), 64, 64), device="cpu")
    value = Value(cfg, shape=(9 + len(cfg.filters), 64, 64))
    for m in (agent, value):
        m.down_sample = torch.nn.AdaptiveAvgPool2d((64, 64))
    agent._apply_isp = _torch_isp
    agent.use_fast_eval = False
    agent.train(); value.train()
    adist.broadcast_parameters([agent, value])
    torch.manual_seed(1)
    det = _StubDetector(nc=4)
    anchors = torch.tensor([[[1.2, 1.6], [2.0, 3.7], [4.1, 2.9]]] * 3)
    loss_fn = DetectionLoss(anchors, nc=4, hyp=default_hyp(4, 64), device="cpu")
    opts = [torch.optim.Adam(agent.parameters(), lr=3e-5), torch.optim.Adam(value.parameters(), lr=3e-5)]
    buckets = [adist.GradBucket(agent), adist.GradBucket(value)]
    g = torch.Generator().manual_seed(100 + rank)
    B = 2
    norms = []
    for it in range(2):
        imgs = torch.rand(B, 3, 64, 64, generator=g) * 0.6 + 0.1
        z = torch.rand(B, cfg.z_dim, generator=g) * 0.98 + 0.01
        states = torch.zeros(B, cfg.num_state_dim)
        labels = [torch.tensor([[0, (rank + b) % 4, 0.5, 0.5, 0.3, 0.4]]) for b in range(B)]
        captured = {}
        real_clip = torch.nn.utils.clip_grad_norm_

        def spy(params, max_norm, *a, **k):
            params = list(params)
            n = real_clip(params, max_norm, *a, **k)
            captured.setdefault("norms", []).append(float(n))
            return n

        torch.nn.utils.clip_grad_norm_ = spy
        try:
            res = train_iteration(cfg, agent, value, det, loss_fn, imgs, z, states, labels, 0.1, opts, buckets=buckets)
        finally:
            torch.nn.utils.clip_grad_norm_ = real_clip
        assert torch.isfinite(res["agent_loss"]) and torch.isfinite(res["value_loss"])
        norms.append(captured["norms"])
    flat = torch.cat([p.detach().reshape(-1) for m in (agent, value) for p in m.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert all(torch.equal(gathered[0], t) for t in gathered), "replicas diverged"
    nt = torch.tensor(norms, dtype=torch.float64)
    allnorms = [torch.zeros_like(nt) for _ in range(world)]
    dist.all_gather(allnorms, nt)
    assert all(torch.equal(allnorms[0], t) for t in allnorms), "the clip must see the same (global) gradient norm"
    dead = [n for n, p in agent.named_parameters() if p.grad is None]
    out.put((rank, float(nt.sum()), len(dead)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_train_iteration_lock_step():
    """Config 4's N>1 path: each rank runs rl.train_iteration on its own batch; after the bucketed all-reduce + global
    clip + Adam the replicas hold bit-identical parameters and saw the same gradient norm (train.py:341-351)."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rl_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    got = sorted(out.get(timeout=5) for _ in range(2))
    assert got[0][1] == got[1][1] and got[0][1] > 0
    assert got[0][2] == got[1][2] and got[0][2] > 0        # fc_mask heads: no gradient, .grad stays None (as the reference)
