"""Fused eval path of Agent.forward: 4 trunk-conv launches (both CNN trunks per launch), one launch for the
eleven 4096->128 hidden layers, one "finish" launch (fc_filter + regressors, selector softmax, sampling /
argmax / forced id, one-hot bookkeeping, state update, penalty, packed parameters) and then adaisp_forward.
~250 ATen launches per step in the reference formulation become 7. Used when the agent is in eval mode with
autograd off; training keeps the PyTorch head path so gradients reach the heads.

Weights are snapshotted (BatchNorm folded with its running statistics) and re-snapshotted whenever a
parameter or buffer changes (tensor version counters), so optimizer steps / load_state_dict are picked up.
"""
import ctypes
import os
import math

import torch

from . import _lib

MAX_FILTERS = 16
REG_TANH_RANGE, REG_EXP_TANH_RANGE, REG_SIGMOID, REG_TANH, REG_WB = 0, 1, 2, 3, 4


class _Regressor(ctypes.Structure):
    _fields_ = [("op", ctypes.c_int32), ("n", ctypes.c_int32), ("kind", ctypes.c_int32),
                ("lo", ctypes.c_float), ("scale", ctypes.c_float), ("bias", ctypes.c_float)]


class _FinishArgs(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in
                ("hidden", "w_filter", "b_filter", "row_filter", "row_slot", "w_sel", "b_sel", "noise", "states",
                 "runtime", "params_all", "packed", "op_ids", "selected", "pdf_out", "surrogate", "new_states",
                 "penalty")] + \
               [(n, ctypes.c_int32) for n in
                ("num_filters", "num_rows", "hid", "param_width", "noise_stride", "train_mode", "forced_id")] + \
               [(n, ctypes.c_float) for n in
                ("one_minus_exploration", "exploration_over_f", "entropy_coef", "log_num_filters", "test_steps",
                 "filter_usage_penalty", "early_stop_penalty", "runtime_lambda")] + \
               [("reg", _Regressor * MAX_FILTERS)]


def _fold(conv, bn):
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    w = conv.weight * scale[:, None, None, None]
    b = (conv.bias if conv.bias is not None else 0.0) * scale + bn.bias - bn.running_mean * scale
    return w.detach().float().contiguous(), b.detach().float().contiguous()


class FastPolicy:
    def __init__(self, agent):
        self.agent = agent
        self._stamp = None
        self._bufs = {}
        L = _lib.load()
        vp, ci = ctypes.c_void_p, ctypes.c_int
        L.adaisp_policy_conv.argtypes = [vp, vp, ci, vp, vp, vp, ci, ci, ci, ci, ci, vp]
        L.adaisp_policy_fc1.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, vp]
        L.adaisp_policy_finish.argtypes = [ctypes.POINTER(_FinishArgs), ci, vp]
        for n in ("adaisp_policy_conv", "adaisp_policy_fc1", "adaisp_policy_finish"):
            getattr(L, n).restype = ci
        self.L = L

    # -- weights -------------------------------------------------------------------------------------------
    def _current_stamp(self):
        a = self.agent
        return sum(t._version for t in a.parameters()) + sum(t._version for t in a.buffers()), \
            next(a.parameters()).device

    def refresh(self):
        a = self.agent
        dev = next(a.parameters()).device
        trunks = [a.feature_extractor, a.action_selection]
        self.layers = []
        for li in range(0, len(trunks[0].layers), 3):
            ws, bs = zip(*[_fold(t.layers[li], t.layers[li + 1]) for t in trunks])
            self.layers.append((torch.stack(ws).contiguous(), torch.stack(bs).contiguous()))
        heads = list(a.filters)
        self.w1 = torch.stack([f.fc1.weight.detach() for f in heads] + [a.fc1.weight.detach()]).float().contiguous()
        self.b1 = torch.stack([f.fc1.bias.detach() for f in heads] + [a.fc1.bias.detach()]).float().contiguous()
        self.head_src = torch.tensor([0] * len(heads) + [1], dtype=torch.int32, device=dev)
        self.w_filter = torch.cat([f.fc_filter.weight.detach() for f in heads]).float().contiguous()
        self.b_filter = torch.cat([f.fc_filter.bias.detach() for f in heads]).float().contiguous()
        rows_f, rows_s = [], []
        for j, f in enumerate(heads):
            n = f.get_num_filter_parameters()
            rows_f += [j] * n
            rows_s += list(range(n))
        self.row_filter = torch.tensor(rows_f, dtype=torch.int32, device=dev)
        self.row_slot = torch.tensor(rows_s, dtype=torch.int32, device=dev)
        self.w_sel = a.fc2.weight.detach().float().contiguous()
        self.b_sel = a.fc2.bias.detach().float().contiguous()
        self.runtime = torch.tensor(a.cfg.filters_runtime, dtype=torch.float32, device=dev) \
            if a.cfg.filter_runtime_penalty else None
        self.specs = [f.regressor_spec() for f in heads]
        self.hid = a.fc1.out_features
        self.D = a.fc1.in_features
        self._stamp = self._current_stamp()

    def _buffers(self, B, dev):
        key = (B, str(dev))
        if key not in self._bufs:
            a = self.agent
            F = len(a.filters)
            acts, size = [], 64
            for w, _ in self.layers:
                size //= 2
                acts.append(torch.empty((2, B, w.shape[1], size, size), dtype=torch.float32, device=dev))
            e = lambda *s, dt=torch.float32: torch.empty(s, dtype=dt, device=dev)  # noqa: E731
            self._bufs[key] = dict(acts=acts, hidden=e(B, F + 1, self.hid))
        return self._bufs[key]

    # -- one step --------------------------------------------------------------------------------------------
    def run(self, pooled, z, states, progress, forced_id, train_mode=False):
        """pooled [B,3,64,64], z [B,>=1], states [B,3+F] (fp32, device) -> dict of step outputs."""
        if self._stamp != self._current_stamp():
            self.refresh()
        a, L = self.agent, self.L
        cfg = a.cfg
        B, dev, F = pooled.shape[0], pooled.device, len(a.filters)
        bufs = self._buffers(B, dev)
        pooled, z, states = pooled.contiguous(), z.contiguous().float(), states.contiguous().float()
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
        with torch.cuda.device(dev):
            src, size, cin = pooled, 64, 3 + states.shape[1]
            for li, (w, b) in enumerate(self.layers):
                rc = L.adaisp_policy_conv(P(src), P(states) if li == 0 else None, states.shape[1] if li == 0 else 0,
                                          P(w), P(b), P(bufs["acts"][li]), 2, B, cin, size, w.shape[1], st)
                _lib._check(rc, "adaisp_policy_conv")
                src, size, cin = bufs["acts"][li], size // 2, w.shape[1]
            feats = bufs["acts"][-1]                                   # [2][B][256][4][4] == [2][B][4096]
            rc = L.adaisp_policy_fc1(P(feats), P(self.head_src), P(self.w1), P(self.b1), P(bufs["hidden"]), B, self.D,
                                     F + 1, self.hid, st)
            _lib._check(rc, "adaisp_policy_fc1")
            pw = a._param_width
            out = dict(
                params_all=torch.empty((B, F, pw), dtype=torch.float32, device=dev),
                packed=torch.empty((B, pw), dtype=torch.float32, device=dev),
                op_ids=torch.empty((B,), dtype=torch.int32, device=dev),
                selected=torch.empty((B,), dtype=torch.int64, device=dev),
                pdf=torch.empty((B, F), dtype=torch.float32, device=dev),
                surrogate=torch.empty((B, 1), dtype=torch.float32, device=dev),
                new_states=torch.empty((B, 3 + F), dtype=torch.float32, device=dev),
                penalty=torch.empty((B, 1), dtype=torch.float32, device=dev))
            fa = _FinishArgs()
            fa.hidden, fa.w_filter, fa.b_filter = bufs["hidden"].data_ptr(), self.w_filter.data_ptr(), self.b_filter.data_ptr()
            fa.row_filter, fa.row_slot = self.row_filter.data_ptr(), self.row_slot.data_ptr()
            fa.w_sel, fa.b_sel = self.w_sel.data_ptr(), self.b_sel.data_ptr()
            fa.noise, fa.states = z.data_ptr(), states.data_ptr()
            fa.runtime = self.runtime.data_ptr() if self.runtime is not None else None
            fa.params_all, fa.packed, fa.op_ids = out["params_all"].data_ptr(), out["packed"].data_ptr(), out["op_ids"].data_ptr()
            fa.selected, fa.pdf_out, fa.surrogate = out["selected"].data_ptr(), out["pdf"].data_ptr(), out["surrogate"].data_ptr()
            fa.new_states, fa.penalty = out["new_states"].data_ptr(), out["penalty"].data_ptr()
            fa.num_filters, fa.num_rows, fa.hid, fa.param_width = F, self.row_filter.numel(), self.hid, pw
            fa.noise_stride, fa.train_mode = z.shape[1], 1 if train_mode else 0
            fa.forced_id = -1 if forced_id is None else int(forced_id)
            fa.one_minus_exploration = 1 - cfg.exploration
            fa.exploration_over_f = cfg.exploration * 1.0 / F
            fa.entropy_coef = (1.0 - progress) * cfg.exploration_penalty
            fa.log_num_filters = math.log(F)
            fa.test_steps = cfg.test_steps
            fa.filter_usage_penalty, fa.early_stop_penalty = cfg.filter_usage_penalty, cfg.early_stop_penalty
            fa.runtime_lambda = cfg.filter_runtime_penalty_lambda
            for j, (op, n, kind, lo, scale, bias) in enumerate(self.specs):
                fa.reg[j] = _Regressor(op, n, kind, lo, scale, bias)
            rc = L.adaisp_policy_finish(ctypes.byref(fa), B, st)
            _lib._check(rc, "adaisp_policy_finish")
        out["_keep"] = (pooled, z, states)        # inputs stay alive until the stream has consumed them
        n_extra = int(os.environ.get("ADAISP_EXPERIMENT_EXTRA_LAUNCHES", "0"))      # measurement aid: N trivial
        if n_extra:                                                                   # one-workgroup launches behind every policy step
            if not hasattr(self, "_extra"):
                self._extra = torch.zeros(64, device=dev)
            for _ in range(n_extra):
                self._extra.add_(1.0)
        return out
