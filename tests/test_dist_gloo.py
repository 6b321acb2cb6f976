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
    # a gradient that exists on ONE rank only (a data-dependent branch): the presence mask travels in the same collective,
    # every rank materialises the averaged gradient and applies the same update; a parameter with no gradient anywhere
    # keeps .grad = None on every rank
    lin = torch.nn.Linear(3, 2)
    with torch.no_grad():
        lin.weight.fill_(1.0); lin.bias.fill_(1.0)
    b2 = adist.GradBucket(lin)
    lin.weight.grad = torch.full_like(lin.weight, 4.0) if rank == 0 else None
    seen = {}
    real_clip = torch.nn.utils.clip_grad_norm_

    def spy(params, max_norm, *a, **k):
        seen["grads"] = [None if p.grad is None else p.grad.clone() for p in params]
        return real_clip([p for p in lin.parameters()], max_norm, *a, **k)

    torch.nn.utils.clip_grad_norm_ = spy
    try:
        adist.synced_step([lin], [torch.optim.SGD(lin.parameters(), lr=1.0)], [b2], max_grad_norm=1e9)
    finally:
        torch.nn.utils.clip_grad_norm_ = real_clip
    assert torch.equal(seen["grads"][0], torch.full_like(lin.weight, 4.0 / world)) and seen["grads"][1] is None
    assert torch.equal(lin.weight.detach(), torch.full_like(lin.weight, 1.0 - 4.0 / world))
    assert torch.equal(lin.bias.detach(), torch.ones_like(lin.bias))
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


def _mask_worker(rank, world, port, out):
    """World 4, unequal presence masks: parameter k of a 6-layer stack has a gradient on rank r iff bit r of PATTERN[k]."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    os.environ.pop("ADAISP_DP_FETCH_MASK", None)
    import torch.distributed as dist
    from adaptiveisp_amd import dist as adist
    torch.set_num_threads(1)
    adist.init_from_env("gloo")
    net = torch.nn.ModuleList([torch.nn.Linear(3, 2, bias=False) for _ in range(6)])
    with torch.no_grad():
        for l in net:
            l.weight.fill_(1.0)
    PATTERN = [0b1111, 0b0001, 0b1010, 0b0000, 0b0110, 0b0000]      # two parameters never get a gradient anywhere
    bucket = adist.GradBucket(net)
    opt = torch.optim.SGD(net.parameters(), lr=1.0)
    for it in range(4):
        for k, l in enumerate(net):
            l.weight.grad = torch.full_like(l.weight, float(rank + 1)) if (PATTERN[k] >> rank) & 1 else None
        adist.synced_step([net], [opt], [bucket], max_grad_norm=1e9)
    for k, l in enumerate(net):
        ranks = [r for r in range(world) if (PATTERN[k] >> r) & 1]
        step = sum(r + 1 for r in ranks) / world                   # mean over ALL ranks of the gradients that exist
        torch.testing.assert_close(l.weight.detach(), torch.full_like(l.weight, 1.0 - 4 * step))
    # the reduced mask was read ONCE (first iteration: the set of never-used parameters is learned); ranks that lack a
    # gradient only inside that set — or lack none — do not read it again. Rank 0 lacks parameter 2 (present on ranks 1, 3)
    # every iteration, so it reads every iteration.
    lacks_live = any(PATTERN[k] and not (PATTERN[k] >> rank) & 1 for k in range(6))
    fetches = getattr(bucket, "mask_fetches", 0)
    assert fetches == (4 if lacks_live else 1), (rank, fetches)
    # a rank whose missing gradients all lie in the learned never-set skips the read; when such a parameter then DOES get a
    # gradient elsewhere, the device-side check raises on the next iteration instead of letting the replicas drift
    net2 = torch.nn.ModuleList([torch.nn.Linear(2, 2, bias=False) for _ in range(2)])
    b2 = adist.GradBucket(net2)
    opt2 = torch.optim.SGD(net2.parameters(), lr=0.1)

    def run(second_on_rank0):
        net2[0].weight.grad = torch.ones_like(net2[0].weight)
        net2[1].weight.grad = torch.ones_like(net2[1].weight) if (second_on_rank0 and rank == 0) else None
        adist.synced_step([net2], [opt2], [b2], max_grad_norm=1e9)

    run(False); run(False)
    assert getattr(b2, "mask_fetches", 0) == 1
    run(True)                                           # rank 0 alone produces a gradient for the "never" parameter
    run(False)                                          # the ranks that skipped it found out on the device; the finding rides in this bucket
    raised = False
    try:
        b2._check_pending()                             # what the next iteration's all_reduce_mean() starts with
    except RuntimeError as e:
        raised = "ADAISP_DP_FETCH_MASK" in str(e)
    assert raised, (rank, raised)                       # EVERY rank raises, at the same point (no rank is left in a collective)
    out.put((rank, fetches))
    dist.destroy_process_group()


def test_four_ranks_unequal_presence_masks():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mask_worker, args=(r, 4, port, out)) for r in range(4)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    got = sorted(out.get(timeout=5) for _ in range(4))
    assert [g[0] for g in got] == [0, 1, 2, 3]


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
    buckets = [adist.GradBucket(agent, value)]
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
    # synced_step ends with zero_grad(set_to_none=True), so .grad says nothing here: the optimizer state does. Adam holds
    # state for exactly the parameters that ever had a gradient: all but the filters' fc_mask heads (masking is off).
    no_state = sorted(n for n, p in agent.named_parameters() if p not in opts[0].state)
    assert no_state and all(".fc_mask." in n for n in no_state), no_state
    assert all(p in opts[1].state for p in value.parameters())
    out.put((rank, float(nt.sum()), len(no_state)))
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
    assert got[0][2] == got[1][2] == 20                    # 10 filters x (fc_mask.weight, fc_mask.bias): no gradient, no Adam state


# ---- `bench.py --gpus N` / `python -m adaptiveisp_amd.train --gpus N` start N ranks themselves -------------------------

def _run_launcher(cmd, env_extra):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, *cmd], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                      # ONE line, from rank 0
    return json.loads(lines[0])


def test_bench_gpus_n_launches_n_ranks():
    """The driver calls `python3 bench.py --gpus N ...` WITHOUT torchrun: the parent must start N ranks as a child process
    and relay rank 0's line. BENCH_REHEARSAL=dry: same harness (barrier, max over ranks, one line), no device."""
    line = _run_launcher(["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"], {"BENCH_REHEARSAL": "dry"})
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["config"]["global_batch"] == 16 and line["config"]["per_gpu_batch"] == 8
    assert line["config"]["parallelism"] == "replicas x2"
    assert line["metric"].startswith("ISP+YOLO forward images/sec @1280x720 bs8") and "REHEARSAL" in line["data"]
    one = _run_launcher(["bench.py", "--steps", "2", "--warmup", "0"], {"BENCH_REHEARSAL": "dry"})
    assert one["n_gpus"] == 1 and one["config"]["global_batch"] == 8


def test_bench_rejects_a_world_that_is_not_gpus():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_REHEARSAL="dry", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1"], cwd=root, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_train_gpus_n_launches_n_ranks():
    line = _run_launcher(["-m", "adaptiveisp_amd.train", "--gpus", "2", "--iters", "2", "--warmup", "0", "--batch", "4"],
                         {"ADAISP_DP_REHEARSAL": "dry"})
    assert line["n_gpus"] == 2 and line["global_batch"] == 8 and line["per_gpu_batch"] == 4
    assert line["sync_bn"] is False and line["grad_buckets"] == 1 and line["iters"] == 2
    # the same one-line shape as bench.py, plus the collective's own figures
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "config"):
        assert k in line, k
    assert line["steps"] == 2 and line["scaling"] == "weak" and line["config"]["parallelism"] == "dp2"
    assert line["grad_bucket_bytes"] == 4 * (8 * 8 + 8 + 2 + 1) and line["all_reduce_ms"] > 0
