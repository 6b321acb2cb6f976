"""Policy network — host-side mirror of the reference's agent.py (Agent, FeatureExtractor,
pdf_sample, one_hot), re-organised around the HIP ISP kernels.

The reference runs ALL filters on the full-resolution batch every step, stacks the ten results and
keeps one per image with a one-hot multiply-sum (agent.py:103-116,154). Here the heads still regress
every filter's parameters (tiny [B,n] tensors), but only the SELECTED filter touches pixels: the
per-image op ids stay on the device and one `adaisp_forward` call filters the batch. Everything the
caller can observe is unchanged: argument/return tuples, debug_info keys, state-dict keys, the
state update, the penalty terms and the selection arithmetic (int64, bit-exact).
"""
import math
import weakref

import torch
import torch.nn as nn

from .isp.isp_function import isp_apply_selected
from .nets import FeatureExtractor as _Trunk
from .nets import Pool64
from . import _lib, trunk_train
from .util import STATE_DROPOUT_BEGIN, STATE_REWARD_DIM, STATE_STEP_DIM, STATE_STOPPED_DIM, enrich_image_input


def pdf_sample(pdf, uniform_noise):
    """Inverse-CDF sampling: index = #{k : cdf_exclusive[k] < u} - 1 (reference agent.py:12-16).
    u == 0 yields -1, i.e. an all-zero one-hot row."""
    pdf = pdf / (torch.sum(pdf, dim=1, keepdim=True) + 1e-36)
    below = torch.less(torch.cumsum(pdf, dim=1) - pdf, uniform_noise)
    return torch.sum(below.to(torch.int32), dim=1) - 1


def one_hot(num_class, index):
    """int64 [B,num_class]; rows for indices outside 0..num_class-1 are all zero (reference agent.py:18-23)."""
    classes = torch.arange(num_class, device=index.device, dtype=index.dtype)
    return (index[:, None] == classes[None, :]).to(torch.int64)


class FeatureExtractor(_Trunk):
    def __init__(self, shape=(14, 64, 64), mid_channels=32, output_dim=4096, dropout_prob=0.5):
        super().__init__(shape=shape, mid_channels=mid_channels, output_dim=output_dim, dropout_prob=dropout_prob)


class _LazyFilterDebug:
    """The reference's per-filter debug list [{'filter_parameters': p[0], 'mask': mask[0]}, ...] as a sequence whose
    entries are made on first access (same values, same shapes)."""

    def __init__(self, filters, table):
        self._filters, self._table, self._items = list(filters), table, None
        self._masks = [f.mask for f in filters]

    def _build(self):
        if self._items is None:
            items = []
            for j, flt in enumerate(self._filters):
                p0 = self._table[0, j, :flt.get_num_filter_parameters()]
                if hasattr(flt, "curve_steps"):                    # curve filters keep the reference's [steps,ch,1,1] view
                    p0 = p0.reshape(flt.curve_steps, -1, 1, 1)
                items.append({'filter_parameters': p0, 'mask': self._masks[j][0]})
            self._items = items
        return self._items

    def __len__(self):
        return len(self._filters)

    def __getitem__(self, i):
        return self._build()[i]

    def __iter__(self):
        return iter(self._build())


class Agent(nn.Module):
    def __init__(self, cfg, shape=(16, 64, 64), device='cuda'):
        super().__init__()
        self.cfg = cfg
        self.feature_extractor = FeatureExtractor(shape=shape, mid_channels=cfg.base_channels,
                                                  output_dim=cfg.feature_extractor_dims,
                                                  dropout_prob=1.0 - cfg.dropout_keep_prob)
        self.filters = []
        for make in self.cfg.filters:
            flt = make(self.cfg, predict=True).to(device)
            setattr(self, flt.get_short_name(), flt)     # state-dict prefix = short name ("E.", "S+.", ...)
            self.filters.append(flt)
        self.action_selection = FeatureExtractor(shape=shape, mid_channels=cfg.base_channels,
                                                 output_dim=cfg.feature_extractor_dims,
                                                 dropout_prob=1.0 - cfg.dropout_keep_prob)
        self.fc1 = nn.Linear(cfg.feature_extractor_dims, cfg.fc1_size)
        self.lrelu = nn.LeakyReLU(negative_slope=0.2)
        self.fc2 = nn.Linear(cfg.fc1_size, len(self.filters))
        self.softmax = nn.Softmax(dim=1)
        self.down_sample = Pool64((shape[1], shape[2]))
        self.runtime = torch.tensor(cfg.filters_runtime, requires_grad=False).to(device)
        # action id -> kernel op code; entry 0 serves the all-zero one-hot (id -1)
        self._op_table_host = [_lib.OP_ZERO] + [int(f.op_code) for f in self.filters]
        self._op_table = None
        self._param_width = max(f.get_num_filter_parameters() for f in self.filters)
        self._fast = None            # fused eval path (policy_fast.FastPolicy), built on first use
        self._ones_mask = None
        self._head_cache = None      # per-filter regressor constants of the batched training heads
        self._pool_cache = None      # (weakref to the image the last eval step returned, its version, its 64x64 pooling)
        self.use_fast_eval = True

    # ------------------------------------------------------------------------------------------
    def _op_ids(self, selected):
        if self._op_table is None or self._op_table.device != selected.device:
            self._op_table = torch.tensor(self._op_table_host, dtype=torch.int32, device=selected.device)
        return self._op_table[(selected + 1).clamp(0, len(self.filters))]

    def _packed_params(self, params, selected):
        """[B,width] row b = flattened parameters of filter selected[b] (zeros for id -1). Differentiable."""
        B = selected.shape[0]
        table = torch.stack([torch.nn.functional.pad(p.reshape(B, -1), (0, self._param_width - p[0].numel()))
                             for p in params], dim=1)                       # [B,F,width]
        idx = selected.clamp(0, len(params) - 1).view(B, 1, 1).expand(B, 1, self._param_width)
        return table.gather(1, idx).squeeze(1)

    def _apply_isp(self, img, packed, op_ids):
        """The one place pixels are touched: selected filter + clip to [0,1] (Filter.forward semantics)."""
        return isp_apply_selected(img, packed, op_ids, clip=True)

    # ------------------------------------------------------------------------------------------
    def plan_step(self, inp, progress, selected_filter_id=None, pooled=None):
        """First half of an eval step on the fused kernels: 64x64 pooling + the policy (trunks, heads, selection, state
        update, penalties). Returns the step's decision — a dict of device tensors (`op_ids`, `packed` parameter rows,
        `new_states`, `selected`, `pdf`, `params_all`, `surrogate`, `penalty`) plus `host_op` (the kernel op code when
        `selected_filter_id` forces it, else None) — without touching the full-resolution image. `pooled`: the
        [B,3,64,64] pooling of x if the caller already has it (the previous step's `apply_step(..., pooled_next=)`
        produced it in the filter launch); otherwise x is pooled here (adaisp_pool64, agent.py:97). `apply_step` is the
        other half; `forward` in eval mode is the two in sequence (agent.py:88-285)."""
        x, z, states = inp
        if self._fast is None:
            from .policy_fast import FastPolicy
            self._fast = FastPolicy(self)
        if pooled is None:
            pooled = self.down_sample(x)
        plan = self._fast.run(pooled, z, states, progress, selected_filter_id)
        plan["host_op"] = None if selected_filter_id is None else self._op_table_host[int(selected_filter_id) + 1]
        return plan

    def apply_step(self, x, plan, out=None, pooled_next=None):
        """Second half: the selected filter of every image on the full-resolution batch (one adaisp_forward call; `plan`
        needs `op_ids` and `packed` only — and `host_op` when the step was teacher-forced — so a caller may carry just
        those between the halves). `pooled_next` ([B,3,64,64]) receives the 64x64 pooling of the result: the next step's
        policy input, written by the same launch (agent.py:97 of the NEXT step)."""
        no_usm = _lib.OP_USM not in self._op_table_host
        return _lib.forward(x, plan["op_ids"], plan["packed"], clip=True, no_usm=no_usm, out=out, pooled=pooled_next,
                            host_op=plan.get("host_op"))

    @staticmethod
    def _version_of(t):
        """The tensor's version counter, or None when it has none: inference tensors (`torch.inference_mode()`, the mode
        the reference's eval entry runs in, val_adaptiveisp.py:104) do not track versions, so nothing could tell a later
        in-place write — such a tensor is never cached."""
        if torch.is_inference(t):
            return None
        try:
            return t._version
        except RuntimeError:
            return None

    # The one piece of state the eval path keeps between calls: the 64x64 planes of the image the previous step returned, reused
    # when THAT tensor object comes back with an unchanged version counter (torch bumps it for its own in-place ops, _lib for
    # every kernel of this package). A writer that goes AROUND both — another extension writing through the raw pointer —
    # is invisible to it: such a caller sets `agent.reuse_pooled_planes = False` (every step then pools its input with a launch
    # of its own, ~17 us at 8 x 720 x 1280) or calls `agent.forget_pooled_planes()` after writing.
    reuse_pooled_planes = True

    def forget_pooled_planes(self):
        self._pool_cache = None

    def _cached_pool(self, x):
        """The pooling of x if x IS the tensor the previous eval step returned, unmodified since."""
        c = self._pool_cache
        if not self.reuse_pooled_planes:
            return None
        if c is not None and c[1] is not None and c[0]() is x and self._version_of(x) == c[1]:
            return c[2]
        return None

    def _forward_fast(self, x, z, states, progress, high_res, selected_filter_id, out=None):
        """Eval-mode step on the fused kernels: 7 launches instead of ~250 (see policy_fast.py)."""
        o = self.plan_step((x, z, states), progress, selected_filter_id, pooled=self._cached_pool(x))
        # the filter launch also pools its result: an eval loop that feeds the returned image back in (val_adaptiveisp.py
        # :293-304, the bench) never runs a separate pooling pass after the first step
        pooled_next = torch.empty((x.shape[0], 3, 64, 64), dtype=torch.float32, device=x.device)
        x_out = self.apply_step(x, o, out=out, pooled_next=pooled_next)
        ver = self._version_of(x_out)                 # (_lib bumps the version of every tensor a kernel writes)
        self._pool_cache = (weakref.ref(x_out), ver, pooled_next) if ver is not None else None
        hr_out = self.apply_step(high_res, o) if high_res is not None else None
        mask = self._ones_mask                      # Filter.get_mask with masking off (isp/filters.py:161-173): ones(1,1,1,1)
        if mask is None or mask.device != x.device:
            mask = self._ones_mask = torch.ones((1, 1, 1, 1), dtype=torch.float32, device=x.device)
        fdi = []
        for j, flt in enumerate(self.filters):
            n = flt.get_num_filter_parameters()
            p0 = o["params_all"][0, j, :n]
            if hasattr(flt, "curve_steps"):                    # curve filters keep the reference's [steps,ch,1,1] view
                p0 = p0.reshape(flt.curve_steps, -1, 1, 1)
            flt.mask, flt.mask_parameters = mask, None
            fdi.append({'filter_parameters': p0, 'mask': mask[0]})
        sel = o["selected"]
        debug_info = {'state': states, 'selected_filter_id': sel[0], 'filter_debug_info': fdi, 'pdf': o["pdf"][0],
                      'selected_filter': sel}

        def debugger(debug_info, combined=True):
            raise NotImplementedError("the drawing debugger needs cv2 and is outside the ISP hot path")

        debugger.width = int(x_out.shape[2])
        if self.cfg.clamp:
            x_out = torch.clip(x_out, min=0.0, max=5.0)
        if high_res is None:
            return (x_out, o["new_states"], o["surrogate"], o["penalty"]), debug_info, debugger
        return (x_out, o["new_states"], hr_out), debug_info, debugger

    def forward(self, inp, progress, high_res=None, selected_filter_id=None, out=None):
        """Reference signature (agent.py:88) plus `out`: an optional preallocated [B,3,H,W] tensor the retouched image is
        written into (eval path; a caller that double-buffers the hand-over to the detector saves a full-frame copy)."""
        train = 1 if self.training else 0
        x, z, states = inp
        if (self.use_fast_eval and not self.training and not torch.is_grad_enabled() and x.is_cuda
                and len(self.filters) <= 16 and all(f._regressor is not None for f in self.filters)):
            return self._forward_fast(x, z, states, progress, high_res, selected_filter_id, out=out)
        if out is not None:
            raise ValueError("`out` is only supported on the fused eval path (eval mode, autograd off)")
        if not self.cfg.shared_feature_extractor:
            raise ValueError("current just support shared_feature_extractor")
        x_down = self.down_sample(x)
        coef = (1.0 - progress) * self.cfg.exploration_penalty
        if train and self.entropy_coef_dev is not None:
            coef = self.entropy_coef_dev                     # a captured iteration: the coefficient is a device scalar the host refreshes
        res = self.policy_heads(
            x_down, z[:, 0:1], states, coef, train=bool(train), forced_id=selected_filter_id, with_masks=not train)
        packed, op_ids, selected, surrogate, penalty, new_states, pdf, table = res[:8]
        masks = res[8] if len(res) > 8 else None            # fc_mask outputs: eval only (unused while masking is off)

        if self._ones_mask is None or self._ones_mask.device != x.device:       # Filter.get_mask with masking off: ones(1,1,1,1),
            self._ones_mask = torch.ones((1, 1, 1, 1), dtype=torch.float32, device=x.device)   # one shared tensor, not a fill per filter
        for j, flt in enumerate(self.filters):
            flt.mask_parameters = masks[j] if masks is not None else None
            flt.mask = flt.get_mask(x, flt.mask_parameters) if flt.use_masking() else self._ones_mask
        # debug_info['filter_debug_info'] (agent.py:138-146: per filter the first image's parameters and mask) is built when
        # somebody reads it — ~50 view ops per call that the training loop never looks at
        filter_debug_info = _LazyFilterDebug(self.filters, table)

        # pixels: only the selected filter runs
        x = self._apply_isp(x, packed, op_ids)
        if high_res is not None:
            high_res_output = self._apply_isp(high_res, packed, op_ids)

        debug_info = {
            'state': states,
            'selected_filter_id': selected[0],
            'filter_debug_info': filter_debug_info,
            'pdf': pdf[0],
            'selected_filter': selected,
        }

        def debugger(debug_info, combined=True):
            raise NotImplementedError("the drawing debugger needs cv2 and is outside the ISP hot path")

        debugger.width = int(x.shape[2])
        if self.cfg.clamp:
            x = torch.clip(x, min=0.0, max=5.0)
        if high_res is None:
            return (x, new_states, surrogate, penalty), debug_info, debugger
        return (x, new_states, high_res_output), debug_info, debugger

    # ------------------------------------------------------------------------------------------
    # Training path: the ten filters' heads as ONE chain of batched ops. Per filter the reference runs fc1 (4096 -> 128),
    # LeakyReLU, fc_filter (128 -> n) and a regressor of a handful of element-wise ops on a [B, n] tensor: ~40 tiny launches
    # forward and ~90 backward per filter pair of passes, i.e. most of the ~750 launches of the agent in one RL iteration,
    # each a few microseconds of dependent latency on an otherwise idle GPU. Here: the fc1 weights concatenated into one
    # [F*128, 4096] matrix (one GEMM), the fc_filter weights scattered into a zero-padded [F, width, 128] stack (one batched
    # GEMM), and the five regressor forms (include/adaisp.h: adaisp_regressor_kind — the same table the fused eval kernel
    # uses) evaluated on the whole [B, F, width] tensor with per-filter constants and masks. Same parameters (the stacks are
    # built from them inside autograd: gradients reach every nn.Linear as before), same formulas per element; matmul
    # summation order differs from ten separate nn.Linear calls, so results agree to fp32 rounding, not bit for bit
    # (tests/test_host_logic.py::test_batched_heads_match_the_per_filter_heads). `agent.batched_heads = False` restores the loop.
    batched_heads = True
    # Training under a captured hipGraph (train.Trainer, graph mode): a 1-element fp32 DEVICE tensor that holds
    # (1 - progress) * cfg.exploration_penalty of the current iteration; forward() then ignores `progress` for that coefficient
    entropy_coef_dev = None

    def _head_consts(self, device):
        c = self._head_cache
        if c is not None and c["device"] == device:
            return c
        if not all(f._regressor is not None for f in self.filters):
            return None
        F, pw = len(self.filters), self._param_width
        specs = [f.regressor_spec() for f in self.filters]                 # (op, n, kind, lo, scale, bias)
        t = lambda v: torch.tensor(v, dtype=torch.float32, device=device).view(1, F, 1)   # noqa: E731
        kind = [sp[2] for sp in specs]
        keep = torch.ones(1, F, pw, device=device)
        for j, sp in enumerate(specs):
            if sp[2] == 4:                                                  # white balance: the R gain is pinned (features * [0,1,1])
                keep[0, j, 0] = 0.0
        rows = [j * pw + k for j, sp in enumerate(specs) for k in range(sp[1])]
        valid = torch.zeros(1, F, pw, device=device)
        for j, sp in enumerate(specs):
            valid[0, j, :sp[1]] = 1.0                                       # slots beyond a filter's n stay zero, as in the padded loop
        b = lambda ks: torch.tensor([k in ks for k in kind], device=device).view(1, F, 1)   # noqa: E731
        c = self._head_cache = dict(device=device, lo=t([sp[3] for sp in specs]), scale=t([sp[4] for sp in specs]),
                                    bias=t([sp[5] for sp in specs]), keep=keep, is_exp=b((1, 4)), is_sig=b((2,)),
                                    is_tanh=b((3,)), is_wb=b((4,)).view(1, F), valid=valid,
                                    rows=torch.tensor(rows, dtype=torch.int64, device=device))
        return c

    def _heads_pre(self, features):
        """Pre-activations [B,F,width] of every filter's fc_filter (zero-padded slots), ten heads as three matmuls."""
        c = self._head_consts(features.device)
        F, pw, B = len(self.filters), self._param_width, features.shape[0]
        hid = self.cfg.fc1_size
        w1 = torch.cat([f.fc1.weight for f in self.filters], 0)            # [F*hid, D]
        b1 = torch.cat([f.fc1.bias for f in self.filters], 0)
        hidden = torch.nn.functional.leaky_relu(torch.nn.functional.linear(features, w1, b1), 0.2).view(B, F, hid)
        wr = torch.cat([f.fc_filter.weight for f in self.filters], 0)      # [sum n, hid]
        br = torch.cat([f.fc_filter.bias for f in self.filters], 0)
        wf = wr.new_zeros(F * pw, hid).index_copy(0, c["rows"], wr).view(F, pw, hid)
        bf = br.new_zeros(F * pw).index_copy(0, c["rows"], br).view(F, 1, pw)
        return torch.baddbmm(bf, hidden.transpose(0, 1), wf.transpose(1, 2)).transpose(0, 1)      # [B,F,pw]

    def _heads_batched(self, features):
        c = self._head_consts(features.device)
        x = self._heads_pre(features)
        base = (torch.tanh(x * c["keep"] + c["bias"]) * 0.5 + 0.5) * c["scale"] + c["lo"]       # tanh_range (isp/filters.py:25-34)
        out = torch.where(c["is_exp"], torch.exp(base), base)               # gamma, white balance
        out = torch.where(c["is_sig"], torch.sigmoid(x), out)               # NLM, S+, BW
        out = torch.where(c["is_tanh"], torch.tanh(x), out)                 # contrast
        lum = 1e-5 + 0.27 * out[..., 0] + 0.67 * out[..., 1] + 0.06 * out[..., 2]               # isp/filters.py:204-206
        # masked BEFORE the reciprocal: the other filters' "lum" can land on exactly 0 (negative outputs), and the unselected
        # branch of a where() still back-propagates 0 * inf = NaN into that filter's heads
        norm = 1.0 / torch.where(c["is_wb"], lum, torch.ones_like(lum))
        return out * (norm[..., None] * c["valid"])

    def policy_heads(self, x_down, noise, states, entropy_coef, train=True, forced_id=None, with_masks=False):
        """Everything of a step between the 64x64-pooled image and the pixels (agent.py:97-149, 234-280): both CNN trunks,
        every filter's heads and regressor, the selector's pdf, sampling / arg-max, one-hot bookkeeping, state update and
        penalties — tensors in, tensors out, no host-side data dependence. `noise` [B,1] = z[:, 0:1]; `entropy_coef` a 1-element tensor = (1 - progress) *
        cfg.exploration_penalty. Returns (packed [B,w], op_ids int32 [B], selected int64 [B], surrogate [B,1], penalty
        [B,1], new_states [B,3+F], pdf [B,F], params_table [B,F,w]) and, with `with_masks`, the list of fc_mask outputs
        (unused while masking is off, isp/filters.py:161-162)."""
        num_filters = len(self.filters)
        sv = states if self.cfg.img_include_states else None
        fe, sel_trunk = self.feature_extractor, self.action_selection
        if (train and trunk_train.serves(fe, x_down, sv) and trunk_train.serves(sel_trunk, x_down, sv)
                and not x_down.requires_grad and fe.droupout.p == sel_trunk.droupout.p):
            # both trunks in one autograd node on the HIP kernels (csrc/isp_trunk_train.hip): 8 launches forward, 11 backward
            # instead of ~300; the state planes are read from the vector (no concat tensor). One dropout call for both
            # feature rows: independent masks, as two calls would draw
            feats = torch.nn.functional.dropout(trunk_train.trunk_features([fe, sel_trunk], [x_down], [sv]), fe.droupout.p, True)
            filter_features, selector_features = feats.unbind(0)
        else:
            net_in = enrich_image_input(self.cfg, x_down, states)
            filter_features, selector_features = fe(net_in), None

        # every filter's heads (cheap), no pixels yet
        B = x_down.shape[0]
        masks = []
        batched = train and not with_masks and self.batched_heads and self._head_consts(x_down.device) is not None
        if batched and x_down.is_cuda and (isinstance(entropy_coef, (int, float)) or
                                           (isinstance(entropy_coef, torch.Tensor) and entropy_coef.numel() == 1)):
            # regressors, pdf, sampling, surrogate, gather, state update, penalties: one launch each way (policy_train.py)
            from . import heads_train, policy_train
            if selector_features is None:
                selector_features = sel_trunk(net_in)
            if heads_train.serves(self, filter_features, selector_features):
                # every filter's fc1 / fc_filter and the selector's fc1 / fc2 on the parameters themselves: 2 launches, 4 backward
                x, logits = heads_train.heads(self, filter_features, selector_features)
            else:
                x = self._heads_pre(filter_features)
                logits = self.fc2(self.lrelu(self.fc1(selector_features)))
            if policy_train.serves(self, x, logits, entropy_coef):
                return policy_train.policy_tail(self, x, logits, noise, states, entropy_coef, sample=True, forced_id=forced_id)
        if batched:
            table = self._heads_batched(filter_features)                    # [B,F,width], ten heads as three matmuls
        else:
            params = []
            for flt in self.filters:
                hidden = flt.lrelu(flt.fc1(filter_features))
                params.append(flt.filter_param_regressor(flt.fc_filter(hidden)))
                if with_masks:
                    masks.append(flt.fc_mask(hidden))
            table = torch.stack([torch.nn.functional.pad(p.reshape(B, -1), (0, self._param_width - p[0].numel()))
                                 for p in params], dim=1)                   # [B,F,width]

        # action selection
        if selector_features is None:
            selector_features = sel_trunk(net_in)
        selector = self.lrelu(self.fc1(selector_features))
        pdf = self.softmax(self.fc2(selector)) + 1e-37
        pdf = pdf * (1 - self.cfg.exploration) + self.cfg.exploration * 1.0 / num_filters
        pdf = pdf / (torch.sum(pdf, dim=1, keepdim=True) + 1e-30)
        entropy = torch.sum(-pdf * torch.log(pdf), dim=1)[:, None]
        if forced_id is not None:
            selected = torch.full((B,), int(forced_id), dtype=torch.int64, device=x_down.device)
        elif train:
            selected = pdf_sample(pdf, noise).to(torch.int64)
        else:
            selected = torch.argmax(pdf, dim=1).to(torch.int64)
        filter_one_hot = one_hot(num_filters, selected)
        surrogate = torch.sum(filter_one_hot * torch.log(pdf + 1e-10), dim=1, keepdim=True)
        op_ids = self._op_ids(selected)
        idx = selected.clamp(0, num_filters - 1).view(B, 1, 1).expand(B, 1, self._param_width)
        packed = table.gather(1, idx).squeeze(1)                # row b = parameters of filter selected[b] (zeros' row for -1 is never read: op ZERO)

        # state update (reference agent.py:234-259)
        step = states[:, STATE_STEP_DIM:STATE_STEP_DIM + 1]
        is_last_step = (torch.abs(step + 1 - self.cfg.test_steps) < 1e-4).to(torch.float32)
        submitted = is_last_step
        filter_usage = states[:, STATE_STEP_DIM + 1:]
        assert filter_usage.dim() == filter_one_hot.dim()
        early_stop_penalty = (1 - is_last_step) * submitted * self.cfg.early_stop_penalty
        usage_penalty = torch.sum(filter_usage * filter_one_hot, dim=1, keepdim=True)
        new_states = torch.cat([submitted, submitted, step + 1, torch.maximum(filter_usage, filter_one_hot)], dim=1)

        entropy_penalty = entropy_coef * (-entropy + math.log(num_filters))
        runtime_penalty = 0.0
        if self.cfg.filter_runtime_penalty:
            runtime_penalty = torch.sum(filter_one_hot * self.runtime.to(x_down.device), dim=1, keepdim=True)
            runtime_penalty = self.cfg.filter_runtime_penalty_lambda * runtime_penalty
        # mean(clip(x - 1, min=0)^2) of the reference (agent.py:279) is identically 0: the kernel has already
        # clipped x to [0,1], so the full-resolution pass that term would cost is skipped.
        over_range = torch.zeros_like(entropy)
        penalty = over_range + entropy_penalty + usage_penalty * self.cfg.filter_usage_penalty + \
            early_stop_penalty + runtime_penalty
        out = (packed, op_ids, selected, surrogate, penalty, new_states, pdf, table)
        return out + (masks,) if with_masks else out
