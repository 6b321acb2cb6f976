"""Reward-side arithmetic: detection loss vs the reference's ComputeLossBatch (golden), TD/advantage/loss formulas."""
import numpy as np
import torch

from adaptiveisp_amd.config import cfg
from adaptiveisp_amd.rl import lr_lambda, td_losses
from adaptiveisp_amd.yolo.loss import DetectionLoss, ciou, per_sample_loss

T = torch.from_numpy


def _loss(g):
    hyp = dict(box=0.05, cls=0.5, obj=1.0 * (96 / 640) ** 2, anchor_t=4.0, cls_pw=1.0, obj_pw=1.0, fl_gamma=0.0,
               label_smoothing=0.0)
    return DetectionLoss(T(g["anchors"]), nc=80, hyp=hyp)


def test_detection_loss_matches_reference(golden):
    g = golden("detloss")
    preds = [T(g[f"p{i}"]) for i in range(3)]
    lbox, lobj, lcls = _loss(g)(preds, T(g["targets"]))
    np.testing.assert_allclose(lbox.numpy(), g["lbox"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(lobj.numpy(), g["lobj"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(lcls.numpy(), g["lcls"], rtol=1e-5, atol=1e-6)


def test_per_sample_loss_matches_reference(golden):
    g = golden("detloss")
    preds = [T(g[f"p{i}"]) for i in range(3)]
    tg = T(g["targets"])
    labels = [tg[tg[:, 0] == b] for b in range(2)]
    out = per_sample_loss(_loss(g), preds, labels)
    for b in range(2):
        np.testing.assert_allclose(out[b].item(), g[f"sample{b}"].sum(), rtol=1e-5)


def test_ciou_basics():
    a = torch.tensor([[0.5, 0.5, 0.2, 0.4]])
    assert abs(ciou(a, a).item() - 1.0) < 1e-5                        # identical boxes
    far = torch.tensor([[5.0, 5.0, 0.2, 0.4]])
    assert ciou(a, far).item() < 0.0                                  # disjoint boxes are penalised by distance


def test_td_losses_literal_semantics():
    B, F = 4, 10
    ns = torch.zeros(B, 3 + F)
    ns[:, 2] = torch.tensor([1.0, 5.0, 8.0, 3.0])                      # step counter; 8 > maximum_trajectory_length (7)
    ns[1, 1] = 1.0                                                    # sample 1 stopped
    l_in = torch.tensor([[0.30], [0.50], [0.20], [2.00]])              # the last one is clipped to 1
    l_re = torch.tensor([[0.10], [0.60], [0.20], [0.40]], requires_grad=True)
    pen = torch.tensor([[0.01], [0.0], [1.0], [0.0]])
    sur = torch.tensor([[-2.0], [-1.0], [-0.5], [-3.0]])
    v_old = torch.tensor([[0.1], [0.2], [0.3], [0.4]], requires_grad=True)
    v_new = torch.tensor([[1.0], [2.0], [3.0], [4.0]])
    mean = torch.tensor([[0.5], [0.5], [0.5], [0.95]])                 # only the last is OUTSIDE (0.01, 0.9)
    out = td_losses(cfg, l_in, l_re, pen, sur, ns, v_old, v_new, mean)
    reward = (torch.tensor([[0.30], [0.50], [0.20], [1.00]]) - l_re.detach()) * 100 - pen
    torch.testing.assert_close(out["reward"], reward)
    # bootstrap survives only for the out-of-range image (literal (1 - truncated)), not stopped, not past max length
    q = reward + torch.tensor([[0.0], [0.0], [0.0], [4.0]])
    torch.testing.assert_close(out["q_value"], q)
    adv = q - v_old.detach()
    torch.testing.assert_close(out["value_loss"], (adv ** 2).mean())
    torch.testing.assert_close(out["agent_loss"], (-q + sur * (-adv)).mean())
    out["value_loss"].backward()
    assert v_old.grad is not None and l_re.grad is None               # the value loss does not reach the reward
    assert abs(lr_lambda(100)(100) - 1e-3) < 1e-12


def td_case(g, tag, device="cpu"):
    """One switch setting of td.npz (tests/golden/gen_golden.py::gen_td — the reference's own statements,
    train.py:264-305, executed on seeded tensors) as td_losses arguments."""
    from adaptiveisp_amd.util import Dict
    c = Dict(cfg)
    use_td, use_truncated, use_penalty = (bool(v) for v in g[f"{tag}.switches"])
    (c.detect_loss_weight, c.all_reward, c.critic_logit_multiplier, c.discount_factor, c.parameter_lr_mul,
     c.maximum_trajectory_length, max_bri) = (float(v) for v in g[f"{tag}.consts"])
    c.use_TD, c.use_penalty = use_td, use_penalty
    leaves = {k: T(g[f"{tag}.{k}"]).to(device).requires_grad_(True)
              for k in ("l_re", "penalty", "surrogate", "old_value", "new_value")}
    fixed = {k: T(g[f"{tag}.{k}"]).to(device) for k in ("l_in", "new_states", "retouch_mean")}
    return c, leaves, fixed, use_truncated, max_bri


def td_run(c, leaves, fixed, use_truncated, max_bri):
    out = td_losses(c, fixed["l_in"], leaves["l_re"], leaves["penalty"], leaves["surrogate"], fixed["new_states"],
                    leaves["old_value"], leaves["new_value"], fixed["retouch_mean"], use_truncated=use_truncated, max_bri=max_bri)
    grads = {}
    for loss in ("value_loss", "agent_loss"):
        gr = torch.autograd.grad(out[loss], list(leaves.values()), retain_graph=True, allow_unused=True)
        for (k, v), x in zip(leaves.items(), gr):
            grads[f"d_{loss}.{k}"] = torch.zeros_like(v) if x is None else x
    return out, grads


def test_td_losses_match_the_reference_statements(golden):
    """a16, reward / TD half: rl.td_losses (the ATen path) against td.npz for all eight (use_TD, use_truncated, use_penalty)
    settings — values and every gradient of both losses, incl. a stopped sample, steps on both sides of
    maximum_trajectory_length, means at 0.005 / 0.01 / 0.5 / 0.9 / 0.95 and detection losses on both sides of the clip."""
    g = golden("td")
    for case in range(8):
        tag = f"c{case}"
        out, grads = td_run(*td_case(g, tag))
        for k in ("reward", "q_value", "value_loss", "agent_loss"):
            np.testing.assert_allclose(out[k].detach().numpy(), g[f"{tag}.out.{k}"], rtol=1e-6, atol=1e-6, err_msg=f"{tag} {k}")
        # the reference REBINDS `advantage` to the policy's multiplier (train.py:298 / 301): -(q - V_old) with use_TD, else
        # -reward; td_losses reports q - V_old (the critic's error) and applies the sign inside agent_loss (checked above)
        want = -g[f"{tag}.out.advantage"] if g[f"{tag}.switches"][0] else out["q_value"].detach().numpy() - g[f"{tag}.old_value"]
        np.testing.assert_allclose(out["advantage"].detach().numpy(), want, rtol=1e-6, atol=1e-6, err_msg=f"{tag} advantage")
        if not g[f"{tag}.switches"][0]:
            np.testing.assert_allclose(-out["reward"].detach().numpy(), g[f"{tag}.out.advantage"], rtol=1e-6, atol=1e-6)
        for k, v in grads.items():
            np.testing.assert_allclose(v.numpy(), g[f"{tag}.{k}"], rtol=1e-6, atol=1e-9, err_msg=f"{tag} {k}")
