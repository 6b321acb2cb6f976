"""clip_grad_norm_ + Adam.step of one model in three launches (csrc/isp_rl_train.hip: adaisp_clip_adam_step).

The reference clips the agent's and the critic's gradients to a norm of 1e-5 and steps two Adam optimizers every iteration
(train.py:341-351). Through torch that is ~14 launches per model (foreach norms, stack, norm, clamp, foreach mul, two fused-Adam
launches that move the 235 MB of parameters / moments at 1.3 TB/s). The kernels read the optimizer's OWN state tensors
(`exp_avg`, `exp_avg_sq`, `step`: state_dict / checkpoints unchanged) and torch's update arithmetic; an optimizer the kernels do
not serve (other options, CPU tensors, state not yet created) is left to torch by the caller (`dist.synced_step`).
"""
import os

import numpy as np
import torch

from . import _lib
from .util import to_device_async

CHUNK = 4096


def enabled():
    return os.environ.get("ADAISP_ADAM_KERNEL", "1") == "1"


def _served(opt):
    if type(opt) is not torch.optim.Adam or len(opt.param_groups) != 1:
        return False
    g = opt.param_groups[0]
    # registered optimizer-step hooks run inside opt.step(), which the kernel path never calls: leave such an optimizer to torch
    if getattr(opt, "_optimizer_step_pre_hooks", None) or getattr(opt, "_optimizer_step_post_hooks", None):
        return False
    return (not g.get("amsgrad") and g.get("weight_decay", 0) == 0 and not g.get("maximize") and not g.get("differentiable")
            and not g.get("capturable") and not isinstance(g["lr"], torch.Tensor)
            and not isinstance(g["betas"][0], torch.Tensor))


def reserve_capture_table(opt):
    """One pinned block for the pointer table of a clip_adam_step that is about to be CAPTURED into a hipGraph (each capture
    keeps its own: the graph's copy node reads it at every replay)."""
    n = sum(1 for p in opt.param_groups[0]["params"] if p in opt.state and "exp_avg" in opt.state[p])
    opt.__dict__.setdefault("_adaisp_capture_blocks", []).append([torch.empty((n, 7), dtype=torch.int64, pin_memory=True), False])


def clip_adam_step(opt, max_norm, lr_dev=None):
    """clip_grad_norm_(the optimizer's parameters, max_norm) + opt.step() on the kernels. Returns False — nothing done — when
    the kernels do not serve this optimizer in this state (the caller then runs torch's clip and step). `lr_dev`: a 1-element
    float64 DEVICE tensor the kernels read the learning rate from instead of `param_groups[0]["lr"]` (an iteration captured in
    a hipGraph: the host refreshes the scalar, the launch stays the same)."""
    if not (enabled() and _served(opt)):
        return False
    group = opt.param_groups[0]
    params = [p for p in group["params"] if p.grad is not None]
    if not params:
        return False
    dev = params[0].device
    if dev.type != "cuda":
        return False
    cache = opt.__dict__.get("_adaisp_table")
    sig = tuple(map(id, params))
    if cache is None or cache["sig"] != sig:
        rows, steps, chunk0 = [], [], 0
        for p in params:
            st = opt.state.get(p)
            if not st or "exp_avg" not in st or not isinstance(st["step"], torch.Tensor) or st["step"].device != dev:
                return False                                   # state not created yet (first step), or a host-side step count
            m, v = st["exp_avg"], st["exp_avg_sq"]
            if not (p.dtype == m.dtype == v.dtype == torch.float32 and st["step"].dtype == torch.float32
                    and p.is_contiguous() and m.is_contiguous() and v.is_contiguous()):
                return False
            n = p.numel()
            rows.append([p.data_ptr(), 0, m.data_ptr(), v.data_ptr(), st["step"].data_ptr(), n, chunk0])
            steps.append(st["step"])
            chunk0 += (n + CHUNK - 1) // CHUNK
        cache = opt.__dict__["_adaisp_table"] = dict(sig=sig, rows=np.array(rows, dtype=np.int64), steps=steps, nchunks=chunk0,
                                                     ws=torch.empty((chunk0 + 2,), dtype=torch.float32, device=dev))
    table = cache["rows"].copy()
    for i, p in enumerate(params):
        g, st = p.grad, opt.state[p]
        if g.dtype != torch.float32 or not g.is_contiguous() or g.is_sparse:
            return False
        # load_state_dict (a resumed checkpoint) replaces the state tensors: the cached addresses must still be theirs
        if st["exp_avg"].data_ptr() != table[i, 2] or st["exp_avg_sq"].data_ptr() != table[i, 3] or st["step"].data_ptr() != table[i, 4]:
            opt.__dict__.pop("_adaisp_table", None)
            return clip_adam_step(opt, max_norm, lr_dev=lr_dev)
        table[i, 1] = g.data_ptr()
    L = _lib.load()
    torch._foreach_add_(cache["steps"], 1)
    if torch.cuda.is_current_stream_capturing():
        # the copy becomes a node of the graph that reads this block at every replay: it stays as it is (the gradients of a
        # replay live where the capture's did), belongs to THIS capture and lives as long as the optimizer. A pinned allocation
        # inside a capture invalidates it: the caller reserved the block beforehand (reserve_capture_table)
        blocks = opt.__dict__.get("_adaisp_capture_blocks") or []
        free = [b for b in blocks if not b[1] and tuple(b[0].shape) == table.shape]
        if not free:
            raise RuntimeError("optim.clip_adam_step inside a stream capture: call optim.reserve_capture_table(opt) before the capture")
        free[0][1] = True
        free[0][0].copy_(torch.from_numpy(table))
        dtable = free[0][0].to(dev, non_blocking=True)
    else:
        dtable = to_device_async(table, dev)
    b1, b2 = group["betas"]
    mn = float(max_norm) if max_norm is not None else 0.0
    with torch.cuda.device(dev):
        if lr_dev is not None:
            if lr_dev.dtype != torch.float64 or lr_dev.numel() != 1 or lr_dev.device != dev:
                raise ValueError("lr_dev: a 1-element float64 tensor on the parameters' device")
            rc = L.adaisp_clip_adam_step_dev(dtable.data_ptr(), len(params), cache["nchunks"], cache["ws"].data_ptr(), mn,
                                             lr_dev.data_ptr(), float(b1), float(b2), float(group["eps"]), _lib._stream())
        else:
            rc = L.adaisp_clip_adam_step(dtable.data_ptr(), len(params), cache["nchunks"], cache["ws"].data_ptr(), mn,
                                         float(group["lr"]), float(b1), float(b2), float(group["eps"]), _lib._stream())
    _lib._check(rc, "adaisp_clip_adam_step")
    for p in params:                                           # written through raw pointers
        _lib._wrote(p)
    # what Optimizer.step's wrapper would have recorded: LambdaLR warns ("lr_scheduler.step() before optimizer.step()") when the
    # first step after a resume went through here (the state exists, so the kernels serve it at once)
    opt._opt_called = True
    if hasattr(opt, "_step_count"):
        opt._step_count += 1
    opt.__dict__["_adaisp_keep"] = (dtable, [p.grad for p in params])   # alive until the next step has been enqueued behind this one
    return True
