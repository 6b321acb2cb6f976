"""Reward / TD-target / loss arithmetic of one RL training iteration — the caller side of the hot path
(reference train.py:262-305, 341-351), written as pure functions so it can be tested without a dataset.

The literal semantics are kept, including the `truncated` flag: the bootstrap term is multiplied by
(1 - truncated) where truncated = 1 for images whose mean brightness is inside (0.01, max_bri), i.e. the value
bootstrap survives only for too-dark / too-bright results (train.py:287-291).
"""
import ctypes
import os

import torch

from .util import STATE_STEP_DIM, STATE_STOPPED_DIM


class _TdArgs(ctypes.Structure):
    _fields_ = ([(k, ctypes.c_int32) for k in ("B", "state_dim", "use_penalty", "use_truncated", "use_td")] +
                [(k, ctypes.c_float) for k in ("detect_loss_weight", "all_reward", "critic_logit_multiplier", "discount_factor",
                                               "parameter_lr_mul", "maximum_trajectory_length", "max_bri")] +
                [(k, ctypes.c_void_p) for k in ("l_in", "l_re", "penalty", "surrogate", "new_states", "old_value", "new_value",
                                                "retouch_mean", "reward", "q_value", "advantage", "losses", "dlosses",
                                                "d_l_re", "d_penalty", "d_surrogate", "d_old_value", "d_new_value")])


class _TdFn(torch.autograd.Function):
    """td_losses as one launch forward and one backward (csrc/isp_rl_train.hip): ~40 + ~60 ATen launches on B floats."""

    @staticmethod
    def forward(ctx, consts, l_in, l_re, penalty, surrogate, new_states, old_value, new_value, retouch_mean):
        from . import _lib
        L = _lib.load()
        B = int(new_states.shape[0])
        ins = [t.contiguous() for t in (l_in, l_re, penalty, surrogate, new_states, old_value, new_value, retouch_mean)]
        a = _TdArgs()
        a.B, a.state_dim = B, int(new_states.shape[1])
        for k, v in consts.items():
            setattr(a, k, v)
        for k, t in zip(("l_in", "l_re", "penalty", "surrogate", "new_states", "old_value", "new_value", "retouch_mean"), ins):
            setattr(a, k, t.data_ptr())
        out = torch.empty((3, B, 1), dtype=torch.float32, device=new_states.device)
        losses = torch.empty((2,), dtype=torch.float32, device=new_states.device)
        a.reward, a.q_value, a.advantage, a.losses = out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), losses.data_ptr()
        with torch.cuda.device(new_states.device):
            _lib._check(L.adaisp_td_fwd(ctypes.byref(a), _lib._stream()), "adaisp_td_fwd")
        ctx.a, ctx.keep = a, ins
        reward, q_value, advantage = out.unbind(0)
        value_loss, agent_loss = losses.unbind(0)
        ctx.mark_non_differentiable(reward, q_value, advantage)
        return reward, q_value, advantage, value_loss, agent_loss

    @staticmethod
    def backward(ctx, _r, _q, _a, g_value, g_agent):
        from . import _lib
        L = _lib.load()
        a, ins = ctx.a, ctx.keep
        dev = ins[4].device
        zero = None
        if g_value is None or g_agent is None:
            zero = torch.zeros((), dtype=torch.float32, device=dev)
        dl = torch.stack([zero if g_value is None else g_value, zero if g_agent is None else g_agent])
        d = torch.empty((5, a.B, 1), dtype=torch.float32, device=dev)
        a.dlosses = dl.data_ptr()
        a.d_l_re, a.d_penalty, a.d_surrogate, a.d_old_value, a.d_new_value = (d[i].data_ptr() for i in range(5))
        with torch.cuda.device(dev):
            _lib._check(L.adaisp_td_bwd(ctypes.byref(a), _lib._stream()), "adaisp_td_bwd")
        d_l_re, d_pen, d_sur, d_old, d_new = (d[i].view_as(t) for i, t in zip(range(5), (ins[1], ins[2], ins[3], ins[5], ins[6])))
        return None, None, d_l_re, d_pen, d_sur, None, d_old, d_new, None


def _td_kernel_serves(*tensors):
    return (os.environ.get("ADAISP_TD_KERNEL", "1") == "1"
            and all(t.is_cuda and t.dtype == torch.float32 for t in tensors)
            and all(t.numel() == tensors[0].numel() for t in tensors))


def td_losses(cfg, detect_input_loss, detect_retouch_loss, penalty, surrogate, new_states, old_value, new_value,
              retouch_mean, use_truncated=True, max_bri=0.9):
    """All inputs [B,1] except new_states [B,3+F]. Returns dict(reward, q_value, advantage, value_loss, agent_loss).

    detect_*_loss are the per-sample detection losses BEFORE weighting/clipping (train.py:264-271)."""
    if new_states.is_cuda and new_states.dtype == torch.float32 and _td_kernel_serves(
            detect_input_loss, detect_retouch_loss, penalty, surrogate, old_value, new_value, retouch_mean):
        consts = dict(use_penalty=int(bool(cfg.use_penalty)), use_truncated=int(bool(use_truncated)), use_td=int(bool(cfg.use_TD)),
                      detect_loss_weight=float(cfg.detect_loss_weight), all_reward=float(cfg.all_reward),
                      critic_logit_multiplier=float(cfg.critic_logit_multiplier), discount_factor=float(cfg.discount_factor),
                      parameter_lr_mul=float(cfg.parameter_lr_mul),
                      maximum_trajectory_length=float(cfg.maximum_trajectory_length), max_bri=float(max_bri))
        reward, q_value, advantage, value_loss, agent_loss = _TdFn.apply(
            consts, detect_input_loss, detect_retouch_loss, penalty, surrogate, new_states, old_value, new_value, retouch_mean)
        return dict(reward=reward, q_value=q_value, advantage=advantage, value_loss=value_loss, agent_loss=agent_loss)
    l_in = torch.clip(detect_input_loss * cfg.detect_loss_weight, 0, 1.0)
    l_re = torch.clip(detect_retouch_loss * cfg.detect_loss_weight, 0, 1.0)
    stopped = new_states[:, STATE_STOPPED_DIM:STATE_STOPPED_DIM + 1]
    reward = (cfg.all_reward + (1 - cfg.all_reward) * stopped) * (l_in.detach() - l_re) * cfg.critic_logit_multiplier
    if cfg.use_penalty:
        reward = reward - penalty
    clear_final = torch.gt(new_states[:, STATE_STEP_DIM:STATE_STEP_DIM + 1], cfg.maximum_trajectory_length).float()
    new_value = new_value * (1.0 - clear_final)
    if use_truncated:
        truncated = torch.where(0.01 < retouch_mean, 1.0, 0.0)
        truncated = torch.where(retouch_mean < max_bri, truncated, torch.zeros_like(truncated))
        q_value = reward + (1.0 - stopped) * cfg.discount_factor * new_value * (1.0 - truncated)
    else:
        q_value = reward + (1.0 - stopped) * cfg.discount_factor * new_value
    advantage = q_value.detach() - old_value
    value_loss = torch.mean(advantage ** 2)
    if cfg.use_TD:
        routine_loss, adv_for_policy = -q_value * cfg.parameter_lr_mul, -advantage
    else:
        routine_loss, adv_for_policy = -reward, -reward
    agent_loss = torch.mean(routine_loss + surrogate * adv_for_policy.detach())
    return dict(reward=reward, q_value=q_value, advantage=advantage, value_loss=value_loss, agent_loss=agent_loss)


_SIDE = {}


def _side_stream(device):
    """One extra stream per device for the work of an iteration that does not depend on the agent."""
    key = str(device)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


def lr_lambda(max_iter):
    """LambdaLR factor 0.1^(3*it/max_it) (train.py:206-218)."""
    return lambda it: 0.1 ** (3.0 * it / max_iter)


def retouch_stats(retouch):
    """[B,2]: per-image mean and number of non-finite values of the retouched batch — one pass (adaisp_image_stats) for the TD
    target's brightness test (train.py:287-291) and the trainer's replay guard (train.py:374-381)."""
    d = retouch.detach()
    if d.is_cuda and d.dtype == torch.float32:
        from . import _lib
        return _lib.image_stats(d)
    return torch.stack([d.mean(dim=tuple(range(1, d.dim()))), (~torch.isfinite(d)).flatten(1).sum(1).to(d.dtype)], dim=1)


def train_iteration(cfg, agent, value, detector, loss_fn, imgs, z, states, labels, progress, optimizers, buckets=None,
                    use_truncated=True, max_bri=0.9, on_retouch=None, assigned=None, lr_dev=None, step=True, one_stream=False):
    """One optimisation step (train.py:255-351). `detector(x)` must return the three raw head maps with autograd
    through to x (the frozen reward model); `buckets` (adaptiveisp_amd.dist.GradBucket per model) enables the
    data-parallel gradient all-reduce before the 1e-5 clip. `on_retouch(retouch)` is called as soon as the retouched batch
    is enqueued, with `retouch_stats(retouch)` and the new state vectors as further arguments (the trainer starts its NaN /
    brightness guard and the read-back of the states there, long before the iteration's backward is launched).
    `assigned` = (packed, packed_pair): the labels' target assignment already on the device (yolo.loss.StaticLabelTables — an
    iteration captured in a hipGraph reads tables of fixed address; `labels` is then unused); `lr_dev`: dist.synced_step;
    `step=False`: stop after backward() — the caller runs the collective and the optimizers (a capture split around an
    all-reduce that stays outside it). `one_stream`: nothing forks onto the second stream (a hipGraph with a fork / join inside
    costs ~0.3 ms of idle device per replay and ten times the launch work on this runtime: tools/graph_launch_gap.py).
    Returns the scalars of td_losses plus the retouched batch."""
    from . import dist as adist
    from .yolo.loss import batched_per_sample_loss as per_sample_loss
    from .yolo.loss import assign_labels, assign_labels_packed
    values = None
    if assigned is not None and getattr(detector, "per_sample_loss_pair", None) is None:
        raise ValueError("train_iteration: `assigned` tables are for the pair engine (yolo.YoloTrainPairEngine)")
    if getattr(detector, "per_sample_loss_pair", None) is not None:
        # HIP training engine, pair form (yolo.YoloTrainPairEngine): ONE detector forward over [input batch; retouched batch],
        # one loss launch for both, the backward over the retouched half only
        if assigned is not None:
            packed, packed_pair = assigned
        else:
            with torch.no_grad():
                packed, packed_pair = assign_labels_packed(loss_fn, detector.head_shapes(), labels, imgs.device, pair=True)
        if imgs.is_cuda and getattr(detector, "begin_input_half", None) is not None and not one_stream:
            # the detector's shallow layers on the INPUT batch need nothing the agent computes: beside the agent's forward
            # (yolo.YoloTrainPairEngine.begin_input_half), on the stream the critic uses later
            cur0, side0 = torch.cuda.current_stream(), _side_stream(imgs.device)
            side0.wait_stream(cur0)
            with torch.cuda.stream(side0):
                detector.begin_input_half(imgs)
            imgs.record_stream(side0)
        (retouch, new_states, surrogate, penalty), _, _ = agent((imgs, z, states), progress)
        stats = retouch_stats(retouch)
        if on_retouch is not None:
            on_retouch(retouch.detach(), stats, new_states.detach())
        if imgs.is_cuda and hasattr(value, "forward_pair") and os.environ.get("ADAISP_CRITIC_STREAM", "1") == "1" and not one_stream:
            # the critic needs the retouched batch, not the detector: its two calls run on a second stream beside the detector's
            # forward — and, autograd replaying every node on its forward's stream, its backward beside the detector's: chains of
            # 5-15 us launches next to launches that fill the chip (7.0 -> 6.7 ms per iteration, four interleaved pairs; the same
            # arrangement measured nothing while the loop still drained the GPU once per iteration). ADAISP_CRITIC_STREAM=0: one stream
            cur, side = torch.cuda.current_stream(), _side_stream(imgs.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                values = value.forward_pair(imgs, states, retouch, new_states)
            for t in (imgs, states, retouch, new_states):
                t.record_stream(side)
        l_in, l_re = detector.per_sample_loss_pair(loss_fn, imgs, retouch, packed, packed_pair)
        if values is not None:
            cur.wait_stream(side)
            for t in values:
                t.record_stream(cur)
    elif getattr(detector, "per_sample_loss", None) is not None and os.environ.get("ADAYOLO_FUSED_LOSS", "1") == "1":
        # HIP training engine: detector forward + one loss launch on its bf16 head maps (csrc/yolo_loss.hip), no fp32 copies.
        # The detection loss of the INPUT batch needs nothing the agent computes: it runs on a second stream beside the
        # agent's forward (a latency chain of ~250 small launches that leaves most CUs idle); the retouched batch's forward
        # — same engine, same buffers — waits for it (measured at 8 x 512 x 512, interleaved on one box: 15.9-16.3 -> 13.4-14.2 ms
        # per iteration; the two critic calls on that stream as well: no further gain). ADAISP_TRAIN_OVERLAP=0: one stream.
        with torch.no_grad():
            packed = assign_labels_packed(loss_fn, detector.head_shapes(), labels, imgs.device)   # host-side build_targets
        if imgs.is_cuda and os.environ.get("ADAISP_TRAIN_OVERLAP", "1") == "1":
            cur, side = torch.cuda.current_stream(), _side_stream(imgs.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side), torch.no_grad():
                l_in = detector.per_sample_loss(loss_fn, imgs, packed)
            (retouch, new_states, surrogate, penalty), _, _ = agent((imgs, z, states), progress)
            stats = retouch_stats(retouch)
            if on_retouch is not None:
                on_retouch(retouch.detach(), stats, new_states.detach())
            cur.wait_stream(side)                           # the engine's buffers are free again
            l_in.record_stream(cur)
        else:
            (retouch, new_states, surrogate, penalty), _, _ = agent((imgs, z, states), progress)
            stats = retouch_stats(retouch)
            if on_retouch is not None:
                on_retouch(retouch.detach(), stats, new_states.detach())
            with torch.no_grad():
                l_in = detector.per_sample_loss(loss_fn, imgs, packed)
        l_re = detector.per_sample_loss(loss_fn, retouch, packed)
    else:
        (retouch, new_states, surrogate, penalty), _, _ = agent((imgs, z, states), progress)
        stats = retouch_stats(retouch)
        if on_retouch is not None:
            on_retouch(retouch.detach(), stats, new_states.detach())
        with torch.no_grad():
            p_in = detector(imgs)
            assigned = assign_labels(loss_fn, p_in, labels)      # same labels, same map shapes for both batches
            l_in = per_sample_loss(loss_fn, p_in, labels, assigned)
        l_re = per_sample_loss(loss_fn, detector(retouch), labels, assigned)
    if values is not None:
        old_value, new_value = values
    elif hasattr(value, "forward_pair") and os.environ.get("ADAISP_CRITIC_PAIR", "1") == "1":
        old_value, new_value = value.forward_pair(imgs, states, retouch, new_states)     # one node for both trunk passes
    else:
        old_value = value(imgs, states)
        new_value = value(retouch, new_states)
    out = td_losses(cfg, l_in, l_re, penalty, surrogate, new_states, old_value, new_value,
                    stats[:, 0:1], use_truncated, max_bri)
    # train.py:341-342 calls backward() on the two losses in turn; gradients accumulate, so one engine pass over both roots
    # deposits the same sums (and the critic's two calls may share autograd nodes)
    torch.autograd.backward([out["value_loss"], out["agent_loss"]])
    models = [agent, value]
    if step:
        if buckets is None:
            buckets = [adist.GradBucket(*models)]
        adist.synced_step(models, optimizers, buckets, max_grad_norm=1e-5, lr_dev=lr_dev)
    out["retouch"] = retouch.detach()
    out["new_states"] = new_states.detach()
    out["detect_loss_input"], out["detect_loss_retouch"] = l_in.detach(), l_re.detach()
    return out
