"""Data-parallel glue for the RL training step: one process per GPU, `torch.distributed` (backend "nccl" = RCCL
over xGMI on the MI355X node, "gloo" on CPU for tests).

The reference trains on one GPU (train.py:45 WORLD_SIZE = 1). Sharding is by replay records: every rank draws its own
batch, runs the whole hot path locally, and the ONLY collective per iteration is an all-reduce (mean) of the agent
and value gradients — issued after backward() and BEFORE the 1e-5 grad-norm clip so the clip sees the global
gradient (train.py:341-349). Both models' gradients are flattened into ONE bucket (28.7 MB + 4.9 MB fp32, SURVEY
8(e)): one large transfer suits the point-to-point xGMI links better than hundreds of small ones.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun). Returns (rank, world, device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available()
    if use_gpu and os.environ.get("ADAISP_DP_REHEARSAL") == "1":
        # every rank on device 0 and gloo instead of RCCL (which refuses two ranks on one device): the N-rank code path on a
        # one-GPU box — a rehearsal of the collectives' placement, not a measurement
        local, backend = 0, backend or "gloo"
    device = torch.device("cuda", local) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend or ("nccl" if use_gpu else "gloo"), rank=rank, world_size=world)
    return rank, world, device


def broadcast_parameters(modules, src=0):
    """Make every rank start from rank `src`'s weights and buffers (one flat broadcast per module)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    for m in modules:
        tensors = [t.data for t in list(m.parameters()) + list(m.buffers()) if t.is_floating_point()]
        if not tensors:
            continue
        flat = torch.cat([t.reshape(-1) for t in tensors])
        dist.broadcast(flat, src)
        off = 0
        for t in tensors:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n


class GradBucket:
    """Flat fp32 view of the gradients of one or MORE modules; `all_reduce_mean()` averages them across ranks in ONE
    collective (agent 28.7 MB + value 4.9 MB as a single flattened bucket). The last `len(params)` floats of
    the buffer are a presence mask (1 where this rank's backward produced a gradient): after the sum every rank knows
    which parameters received a gradient ANYWHERE, so a parameter whose local .grad is None on one rank only still gets
    the averaged gradient (and its Adam update) on every rank — the replicas cannot drift apart over a data-dependent
    branch. Parameters that never enter the loss on any rank (the filters' fc_mask heads: masking is hard-wired off,
    isp/filters.py:161-162) keep .grad = None as in the reference, so Adam creates no state for them."""

    def __init__(self, *modules):
        self.modules = list(modules)
        self.per_module = [[p for p in m.parameters() if p.requires_grad] for m in modules]   # (walked once, not per iteration)
        self.params = [p for ps in self.per_module for p in ps]
        self.numel = sum(p.numel() for p in self.params)
        self.flat = None
        self.views = None

    @staticmethod
    def _active():
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def all_reduce_mean(self, async_op=False):
        if not self._active():
            return None                       # single process: the gradients stay where autograd put them
        self._check_pending()
        dev = self.params[0].device
        if self.flat is None or self.flat.device != dev:
            # [gradients | presence mask | 1 float: "some rank skipped an update last iteration" (see _check_pending)]
            self.flat = torch.zeros(self.numel + len(self.params) + 1, dtype=torch.float32, device=dev)
            self._late_dev = torch.zeros(1, dtype=torch.float32, device=dev)
            self._late_host = torch.zeros(1, dtype=torch.float32, pin_memory=dev.type == "cuda")
            self.views, off = [], 0
            for p in self.params:                                # one view of the bucket per parameter, made once
                self.views.append(self.flat[off:off + p.numel()].view_as(p))
                off += p.numel()
            # the presence mask is built in ONE persistent host buffer (pinned when the bucket lives on a GPU: the copy is
            # then asynchronous; a fresh pageable tensor per iteration made it a blocking one)
            self._present_host = torch.zeros(len(self.params), dtype=torch.float32, pin_memory=dev.type == "cuda")
        ph = self._present_host
        for i, p in enumerate(self.params):
            ph[i] = 0.0 if p.grad is None else 1.0
        self.flat[self.numel:self.numel + len(self.params)].copy_(ph, non_blocking=True)
        self.flat[-1:].copy_(self._late_dev)                     # last iteration's local finding travels with this one's gradients
        self._late_dev.zero_()
        have = [(v, p.grad) for v, p in zip(self.views, self.params) if p.grad is not None]
        absent = [v for v, p in zip(self.views, self.params) if p.grad is None]
        if have:                                                 # two multi-tensor launches instead of one copy per parameter
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        if absent:
            torch._foreach_zero_(absent)
        work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=async_op)
        if not async_op:
            self.flat[:self.numel].div_(dist.get_world_size())
        return work

    # ---- which parameters received a gradient on ANY rank ------------------------------------------------------------
    # Reading the reduced mask is a device-to-host copy that waits for the whole backward AND the all-reduce: done every
    # iteration it serialises the host behind the GPU (the next iteration's ~10 ms of enqueue work cannot start), for the one
    # configuration data parallelism exists for. It is needed only when THIS rank lacks a gradient for a parameter that may
    # have one elsewhere. So: the set of parameters without a gradient on any rank is learned by a (blocking) read the first
    # time this rank lacks one, and afterwards a rank whose missing gradients all lie in that set reads nothing. That the
    # set is still right is verified on the DEVICE: (reduced mask of the set) > 0 anywhere means this rank skipped an update
    # another rank made. That finding (one float) rides in the NEXT iteration's bucket, so after that all-reduce every rank
    # holds the same verdict; it is copied to pinned memory behind an event and read at the start of the iteration after —
    # when it has long landed — where ALL ranks raise together (a raise on some ranks only would leave the others hanging in
    # the next collective). Two iterations late, loud, and only for a case this model does not produce (the only parameters
    # without gradients are the fc_mask heads, on every rank). ADAISP_DP_FETCH_MASK=1: read the mask every iteration.
    def _check_pending(self):
        pend, self._pending = getattr(self, "_pending", None), None
        if pend is not None:
            flag, event = pend
            if event is not None:
                event.synchronize()
            if float(flag[0]) > 0:
                raise RuntimeError("GradBucket: a parameter that had no gradient on any rank received one on some rank two "
                                   "iterations ago and the ranks that lacked it skipped its update (replicas diverged by one "
                                   "step). Set ADAISP_DP_FETCH_MASK=1 to read the presence mask every iteration.")

    def _fetch_anywhere(self):
        self.mask_fetches = getattr(self, "mask_fetches", 0) + 1
        anywhere = (self.flat[self.numel:self.numel + len(self.params)].cpu() > 0).tolist()
        self._never = frozenset(i for i, a in enumerate(anywhere) if not a)
        self._never_idx = None
        return anywhere

    def finish(self, work=None):
        """Complete an async all-reduce (if any) and scatter the averaged bucket back into .grad (materialising .grad
        where another rank had a gradient and this one did not)."""
        if not self._active():
            return
        if work is not None:
            work.wait()
            self.flat[:self.numel].div_(dist.get_world_size())
        local_absent = [i for i, p in enumerate(self.params) if p.grad is None]
        never = getattr(self, "_never", None)
        if not local_absent:
            anywhere = None                                      # every gradient exists here: nothing to learn from the mask
        elif os.environ.get("ADAISP_DP_FETCH_MASK") == "1" or never is None or not never.issuperset(local_absent):
            anywhere = self._fetch_anywhere()
        else:
            anywhere = None                                      # all of them are known to be absent everywhere: verify, do not wait
            if getattr(self, "_never_idx", None) is None:
                self._never_idx = torch.tensor(sorted(never), dtype=torch.int64, device=self.flat.device) + self.numel
            self._late_dev.copy_((self.flat.index_select(0, self._never_idx) > 0).any().reshape(1))
        # the verdict every rank reduced in THIS iteration (what any rank found in the previous one), read one iteration on
        self._late_host.copy_(self.flat[-1:], non_blocking=True)
        event = None
        if self.flat.is_cuda:
            event = torch.cuda.Event()
            event.record()
        self._pending = (self._late_host, event)
        dst, src = [], []
        for i, (p, v) in enumerate(zip(self.params, self.views)):
            if p.grad is None:
                if anywhere is None or not anywhere[i]:
                    continue
                p.grad = torch.empty_like(p)
            dst.append(p.grad)
            src.append(v)
        if dst:
            torch._foreach_copy_(dst, src)


def _same_params(model, opt, cached):
    """The optimizer steps exactly the parameters of `model` that the clip would see (the norm is taken over them)."""
    key = "_adaisp_same_%d" % id(model)
    hit = opt.__dict__.get(key)
    if hit is None:
        mine = {id(p) for p in (cached.get(id(model)) or [p for p in model.parameters() if p.requires_grad])}
        theirs = {id(p) for g in opt.param_groups for p in g["params"] if p.requires_grad}
        hit = opt.__dict__[key] = mine == theirs
    return hit


def reduce_gradients(buckets):
    """The iteration's collective: every bucket all-reduced (mean) and scattered back into .grad."""
    works = [b.all_reduce_mean(async_op=GradBucket._active()) for b in buckets]
    for b, w in zip(buckets, works):
        b.finish(w)


def synced_step(models, optimizers, buckets, max_grad_norm=1e-5, lr_dev=None, collective=True, release=True):
    """After loss.backward(): all-reduce(mean) the gradients (one bucket = one collective), clip, step.
    Mirrors train.py:341-351 with the collective inserted before the clip. `lr_dev`: per optimizer a 1-element float64 device
    tensor holding its learning rate (captured iterations: optim.clip_adam_step); such an optimizer MUST be served by the
    kernels — torch's step would bake the host-side rate into the capture. `collective=False`: the caller has already reduced
    the gradients (`reduce_gradients`); `release=False`: the gradients stay where they are after the step (a captured backward
    writes them to the same addresses at every replay) instead of zero_grad(set_to_none=True)."""
    optimizers = list(optimizers)
    if collective:
        reduce_gradients(buckets)
    cached = {id(m): ps for b in buckets for m, ps in zip(getattr(b, "modules", ()), getattr(b, "per_module", ()))}
    from . import optim as aoptim
    paired = len(models) == len(optimizers)
    for i, m in enumerate(models):
        # clip + Adam of one model as three launches where the kernels serve the optimizer (optim.clip_adam_step), else torch's
        if paired and _same_params(m, optimizers[i], cached) and aoptim.clip_adam_step(
                optimizers[i], max_grad_norm, lr_dev=None if lr_dev is None else lr_dev[i]):
            if release:
                optimizers[i].zero_grad(set_to_none=True)
            optimizers[i] = None
            continue
        if lr_dev is not None:
            raise RuntimeError("synced_step: lr_dev given but the clip + Adam kernels do not serve this optimizer "
                               "(optim.clip_adam_step): a captured step needs them")
        torch.nn.utils.clip_grad_norm_(cached.get(id(m)) or list(m.parameters()), max_grad_norm)
    for o in optimizers:
        if o is None:
            continue
        o.step()
        if release:
            o.zero_grad(set_to_none=True)   # (the reference's default too: the next backward writes the gradients instead of adding to zeros)


def launch_ranks(n, target, argv, module=False):
    """`--gpus N` without torchrun: start N ranks of `target` (a script path, or a module name with module=True) under
    `python -m torch.distributed.run` on 127.0.0.1 as a CHILD process and return its exit code. The caller must not have
    touched the GPU (importing torch does not): the parent only waits, the children own the devices. stdout/stderr are
    inherited, so rank 0's JSON line is the parent's output."""
    import socket
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n)}",
           "--master-addr", "127.0.0.1", "--master-port", str(port)]
    cmd += (["-m", target] if module else [target]) + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)
