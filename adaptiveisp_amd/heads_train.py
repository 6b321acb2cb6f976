"""The policy's parameter heads in training mode on the HIP kernels (csrc/isp_heads_train.hip: adaisp_heads_fwd / _bwd).

`Agent.policy_heads` (agent.py:103-121 of the reference: per filter fc1 -> LeakyReLU -> fc_filter, the selector's fc1 -> LeakyReLU ->
fc2) needs, per iteration, the pre-activations x [B,F,width] of every filter's regressor and the selector's logits [B,F]. As batched
ATen ops that is 19 launches forward and ~25 backward around four M = 8 GEMMs (Agent._heads_pre); here one autograd node over the
filters' OWN parameters: 2 launches forward, 4 backward, the gradients written straight into one buffer per parameter kind whose
views become the parameters' .grad. Same formulas; the summation order differs from rocBLAS's, so results agree with the ATen path to
fp32 rounding (tests/test_gpu_heads_train.py), and two runs agree bit for bit.
"""
import ctypes
import os

import torch

from . import _lib
from .policy_fast import MAX_FILTERS

MAX_B = 8
_P = ctypes.c_void_p


class _HeadsArgs(ctypes.Structure):
    _fields_ = ([(k, ctypes.c_int32) for k in ("B", "F", "D", "hid", "pw")] + [("n", ctypes.c_int32 * MAX_FILTERS)] +
                [("feat_f", _P), ("feat_s", _P)] +
                [(k, _P * MAX_FILTERS) for k in ("w1", "b1", "wf", "bf")] +
                [(k, _P) for k in ("ws1", "bs1", "ws2", "bs2", "hidden", "x", "logits", "dx", "dlogits", "dhid", "part")] +
                [(k, _P * MAX_FILTERS) for k in ("dw1", "db1", "dwf", "dbf")] +
                [(k, _P) for k in ("dws1", "dbs1", "dws2", "dbs2", "dfeat_f", "dfeat_s")])


def enabled():
    return os.environ.get("ADAISP_HEADS_KERNEL", "1") == "1"


def _params(agent):
    """(fc1.weight, fc1.bias, fc_filter.weight, fc_filter.bias) per filter, then the selector's four."""
    ps = []
    for f in agent.filters:
        ps += [f.fc1.weight, f.fc1.bias, f.fc_filter.weight, f.fc_filter.bias]
    return ps + [agent.fc1.weight, agent.fc1.bias, agent.fc2.weight, agent.fc2.bias]


def serves(agent, filter_features, selector_features):
    if not (enabled() and filter_features.is_cuda and selector_features is not None and filter_features.dtype == torch.float32
            and selector_features.dtype == torch.float32 and filter_features.shape == selector_features.shape
            and filter_features.dim() == 2 and 1 <= filter_features.shape[0] <= MAX_B and len(agent.filters) <= MAX_FILTERS):
        return False
    D, hid = filter_features.shape[1], agent.cfg.fc1_size
    if D % 1024 or hid % 8 or hid > 256 or agent._param_width > 24:
        return False
    ps = _params(agent)
    if not all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in ps):
        return False
    F = len(agent.filters)
    ok = all(tuple(ps[4 * j].shape) == (hid, D) and tuple(ps[4 * j + 1].shape) == (hid,) and ps[4 * j + 2].shape[1] == hid
             and ps[4 * j + 2].shape[0] == ps[4 * j + 3].shape[0] <= agent._param_width for j in range(F))
    return ok and tuple(ps[-4].shape) == (hid, D) and tuple(ps[-2].shape) == (F, hid)


class _HeadsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, agent, feat_f, feat_s, *ps):
        L = _lib.load()
        F, hid, pw = len(agent.filters), agent.cfg.fc1_size, agent._param_width
        B, D = (int(v) for v in feat_f.shape)
        dev = feat_f.device
        feat_f, feat_s = feat_f.contiguous(), feat_s.contiguous()
        a = _HeadsArgs()
        a.B, a.F, a.D, a.hid, a.pw = B, F, D, hid, pw
        for j in range(F):
            a.n[j] = int(ps[4 * j + 2].shape[0])
            a.w1[j], a.b1[j], a.wf[j], a.bf[j] = (ps[4 * j + k].data_ptr() for k in range(4))
        a.ws1, a.bs1, a.ws2, a.bs2 = (p.data_ptr() for p in ps[-4:])
        a.feat_f, a.feat_s = feat_f.data_ptr(), feat_s.data_ptr()
        hidden = torch.empty((B, F + 1, hid), dtype=torch.float32, device=dev)
        x = torch.empty((B, F, pw), dtype=torch.float32, device=dev)
        logits = torch.empty((B, F), dtype=torch.float32, device=dev)
        a.hidden, a.x, a.logits = hidden.data_ptr(), x.data_ptr(), logits.data_ptr()
        with torch.cuda.device(dev):
            _lib._check(L.adaisp_heads_fwd(ctypes.byref(a), _lib._stream()), "adaisp_heads_fwd")
        ctx.a, ctx.keep, ctx.shapes = a, (feat_f, feat_s, hidden) + tuple(ps), [tuple(p.shape) for p in ps]
        return x, logits

    @staticmethod
    def backward(ctx, dx, dlogits):
        L = _lib.load()
        a = ctx.a
        feat_f, feat_s, hidden = ctx.keep[:3]
        B, F, D, hid = a.B, a.F, a.D, a.hid
        dev = feat_f.device
        zero = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)  # noqa: E731
        dx = zero(B, F, a.pw) if dx is None else dx.contiguous()
        dlogits = zero(B, F) if dlogits is None else dlogits.contiguous()
        # one buffer per parameter kind: the gradients of the ten filters are views of it (what AccumulateGrad keeps)
        nrows = [s[0] for s in ctx.shapes[2:4 * F:4]]
        dw1 = torch.empty((F + 1, hid, D), dtype=torch.float32, device=dev)
        db1 = torch.empty((F + 1, hid), dtype=torch.float32, device=dev)
        dwf = torch.empty((sum(nrows) + F, hid), dtype=torch.float32, device=dev)
        dbf = torch.empty((sum(nrows) + F,), dtype=torch.float32, device=dev)
        dfeat = torch.empty((2, B, D), dtype=torch.float32, device=dev)
        dhid = torch.empty((B, F + 1, hid), dtype=torch.float32, device=dev)
        part = torch.empty((F + 1, B, D), dtype=torch.float32, device=dev)
        grads, r0 = [], 0
        for j in range(F):
            g = (dw1[j], db1[j], dwf[r0:r0 + nrows[j]], dbf[r0:r0 + nrows[j]])
            r0 += nrows[j]
            a.dw1[j], a.db1[j], a.dwf[j], a.dbf[j] = (t.data_ptr() for t in g)
            grads += g
        sel = (dw1[F], db1[F], dwf[r0:r0 + F], dbf[r0:r0 + F])
        a.dws1, a.dbs1, a.dws2, a.dbs2 = (t.data_ptr() for t in sel)
        grads += sel
        a.dx, a.dlogits, a.dhid, a.part = dx.data_ptr(), dlogits.data_ptr(), dhid.data_ptr(), part.data_ptr()
        a.dfeat_f, a.dfeat_s = dfeat[0].data_ptr(), dfeat[1].data_ptr()
        with torch.cuda.device(dev):
            _lib._check(L.adaisp_heads_bwd(ctypes.byref(a), _lib._stream()), "adaisp_heads_bwd")
        return (None, dfeat[0], dfeat[1]) + tuple(grads)


def heads(agent, filter_features, selector_features):
    """(x [B,F,width] pre-activations of every filter's fc_filter, zero in the padded slots; logits [B,F] of the selector)."""
    return _HeadsFn.apply(agent, filter_features, selector_features, *_params(agent))
