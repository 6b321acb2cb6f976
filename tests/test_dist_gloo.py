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
