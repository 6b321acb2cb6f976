"""The policy's tail in training mode on one HIP launch each way (csrc/isp_rl_train.hip: k_policy_tail_fwd / _bwd).

Between the fully-connected layers and the pixels, Agent.forward (agent.py:103-149, 234-280) is ~70 element-wise ATen
launches on [B, F] / [B, F, width] tensors — regressors of every filter, softmax, exploration mix, entropy, pdf_sample,
one-hot, surrogate, gather of the selected parameters, state update, penalties — and ~50 more in the backward. The kernel pair
does the same arithmetic in the order of the fused eval kernel (adaisp_policy_finish) and returns what `Agent.policy_heads`
returns; gradients reach the heads' pre-activations (selected filter's row) and the selector's logits.
"""
import ctypes
import math
import os

import torch

from . import _lib
from .policy_fast import MAX_FILTERS, _Regressor


class _TailArgs(ctypes.Structure):
    _fields_ = ([(k, ctypes.c_int32) for k in ("B", "num_filters", "param_width", "noise_stride", "sample", "forced_id")] +
                [(k, ctypes.c_float) for k in ("one_minus_exploration", "exploration_over_f", "entropy_coef", "log_num_filters",
                                               "test_steps", "filter_usage_penalty", "early_stop_penalty", "runtime_lambda")] +
                [("reg", _Regressor * MAX_FILTERS)] +
                [(k, ctypes.c_void_p) for k in ("x", "logits", "noise", "states", "runtime", "table", "packed", "op_ids",
                                                "selected", "pdf", "surrogate", "new_states", "penalty", "d_packed",
                                                "d_surrogate", "d_penalty", "d_x", "d_logits", "entropy_coef_dev")])


def enabled():
    return os.environ.get("ADAISP_POLICY_TAIL_KERNEL", "1") == "1"


class _TailFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, agent, entropy_coef, sample, forced_id, x, logits, noise, states):
        L = _lib.load()
        cfg = agent.cfg
        B, F, pw = (int(v) for v in x.shape)
        dev = x.device
        x, logits, states = x.contiguous(), logits.contiguous(), states.contiguous()
        if noise.stride(-1) != 1 and noise.shape[-1] != 1:
            noise = noise.contiguous()
        a = _TailArgs()
        a.B, a.num_filters, a.param_width = B, F, pw
        a.noise_stride = int(noise.stride(0)) if noise.dim() > 1 else 1
        a.sample, a.forced_id = 1 if sample else 0, -1 if forced_id is None else int(forced_id)
        a.one_minus_exploration, a.exploration_over_f = 1 - cfg.exploration, cfg.exploration * 1.0 / F
        if isinstance(entropy_coef, torch.Tensor):       # a device scalar read by the kernels at run time (captured iterations)
            a.entropy_coef, a.entropy_coef_dev = 0.0, entropy_coef.data_ptr()
        else:
            a.entropy_coef, a.entropy_coef_dev = float(entropy_coef), None
        a.log_num_filters, a.test_steps = math.log(F), cfg.test_steps
        a.filter_usage_penalty, a.early_stop_penalty = cfg.filter_usage_penalty, cfg.early_stop_penalty
        a.runtime_lambda = cfg.filter_runtime_penalty_lambda if cfg.filter_runtime_penalty else 0.0
        for j, f in enumerate(agent.filters):
            a.reg[j] = _Regressor(*f.regressor_spec())
        runtime = agent.runtime.to(dev) if cfg.filter_runtime_penalty else None
        e = lambda *s, dt=torch.float32: torch.empty(s, dtype=dt, device=dev)  # noqa: E731
        table, packed, op_ids, selected = e(B, F, pw), e(B, pw), e(B, dt=torch.int32), e(B, dt=torch.int64)
        pdf, surrogate, new_states, penalty = e(B, F), e(B, 1), e(B, 3 + F), e(B, 1)
        a.x, a.logits, a.noise, a.states = x.data_ptr(), logits.data_ptr(), noise.data_ptr(), states.data_ptr()
        a.runtime = None if runtime is None else runtime.data_ptr()
        a.table, a.packed, a.op_ids, a.selected = table.data_ptr(), packed.data_ptr(), op_ids.data_ptr(), selected.data_ptr()
        a.pdf, a.surrogate, a.new_states, a.penalty = pdf.data_ptr(), surrogate.data_ptr(), new_states.data_ptr(), penalty.data_ptr()
        with torch.cuda.device(dev):
            _lib._check(L.adaisp_policy_tail_fwd(ctypes.byref(a), _lib._stream()), "adaisp_policy_tail_fwd")
        ctx.a, ctx.keep = a, (x, logits, noise, states, runtime, pdf, selected, entropy_coef)
        ctx.mark_non_differentiable(op_ids, selected, new_states, pdf, table)
        return packed, op_ids, selected, surrogate, penalty, new_states, pdf, table

    @staticmethod
    def backward(ctx, d_packed, _o, _s, d_sur, d_pen, _n, _p, _t):
        L = _lib.load()
        a = ctx.a
        x, logits = ctx.keep[0], ctx.keep[1]
        d_x, d_logits = torch.empty_like(x), torch.empty_like(logits)
        keep = [None if g is None else g.contiguous() for g in (d_packed, d_sur, d_pen)]
        a.d_packed, a.d_surrogate, a.d_penalty = (None if g is None else g.data_ptr() for g in keep)
        a.d_x, a.d_logits = d_x.data_ptr(), d_logits.data_ptr()
        with torch.cuda.device(x.device):
            _lib._check(L.adaisp_policy_tail_bwd(ctypes.byref(a), _lib._stream()), "adaisp_policy_tail_bwd")
        return None, None, None, None, d_x, d_logits, None, None


def _is_dev_scalar(t, like):
    return isinstance(t, torch.Tensor) and t.numel() == 1 and t.dtype == torch.float32 and t.device == like.device and not t.requires_grad


def serves(agent, x, logits, entropy_coef):
    return (enabled() and x.is_cuda and x.dtype == torch.float32 and logits.dtype == torch.float32
            and (isinstance(entropy_coef, (int, float)) or _is_dev_scalar(entropy_coef, x)) and len(agent.filters) <= MAX_FILTERS
            and x.shape[2] <= 24 and all(f._regressor is not None for f in agent.filters))


def policy_tail(agent, x, logits, noise, states, entropy_coef, sample=True, forced_id=None):
    """(packed, op_ids, selected, surrogate, penalty, new_states, pdf, table) of Agent.policy_heads from the heads'
    pre-activations x [B,F,width] and the selector's logits [B,F]."""
    return _TailFn.apply(agent, entropy_coef, sample, forced_id, x, logits, noise, states)
