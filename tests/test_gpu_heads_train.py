"""adaisp_heads_fwd / _bwd (csrc/isp_heads_train.hip): the policy's parameter heads in training mode — every filter's fc1 -> LeakyReLU
-> fc_filter and the selector's fc1 -> LeakyReLU -> fc2 (agent.py:103-121) — against the ATen formulation they replace
(Agent._heads_pre + the selector's two nn.Linear), forward and backward, at fp32 rounding; bit-reproducible."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _agent():
    from _synth import synth_state_dict
    from adaptiveisp_amd.agent import Agent
    from adaptiveisp_amd.config import cfg
    torch.manual_seed(0)
    agent = Agent(cfg, shape=(16, 64, 64), device=DEV)
    agent.load_state_dict(synth_state_dict(agent, seed=0))
    return cfg, agent.to(DEV).train()


def _run(agent, ff, fs, kernel, wx, wl):
    from adaptiveisp_amd import heads_train
    agent.zero_grad(set_to_none=True)
    ff, fs = ff.clone().requires_grad_(True), fs.clone().requires_grad_(True)
    if kernel:
        assert heads_train.serves(agent, ff, fs)
        x, logits = heads_train.heads(agent, ff, fs)
    else:
        x = agent._heads_pre(ff)
        logits = agent.fc2(agent.lrelu(agent.fc1(fs)))
    ((x * wx).sum() + (logits * wl).sum()).backward()
    grads = {n: p.grad.clone() for n, p in agent.named_parameters() if p.grad is not None}
    return x.detach().clone(), logits.detach().clone(), ff.grad.clone(), fs.grad.clone(), grads


@pytest.mark.parametrize("B", [8, 3, 1])
def test_heads_kernels_match_the_aten_formulation(B):
    from _margins import close_scaled
    cfg, agent = _agent()
    F, pw = len(agent.filters), agent._param_width
    g = torch.Generator(device=DEV).manual_seed(B)
    ff = torch.randn(B, 4096, generator=g, device=DEV)
    fs = torch.randn(B, 4096, generator=g, device=DEV)
    wx = torch.randn(B, F, pw, generator=g, device=DEV)
    wl = torch.randn(B, F, generator=g, device=DEV)
    ref = _run(agent, ff, fs, False, wx, wl)
    got = _run(agent, ff, fs, True, wx, wl)
    again = _run(agent, ff, fs, True, wx, wl)
    for a, b in zip(got[:4], again[:4]):
        assert torch.equal(a, b)                                            # fixed summation order: bit-reproducible
    assert got[4].keys() == again[4].keys() and all(torch.equal(got[4][k], again[4][k]) for k in got[4])
    for name, a, b in zip(("x", "logits", "d filter features", "d selector features"), got[:4], ref[:4]):
        close_scaled("heads_train." + name.replace(" ", "_"), a, b, 2e-5, err_msg=f"B={B}")
    assert got[4].keys() == ref[4].keys() and len(got[4]) == 4 * F + 4      # every head parameter and nothing else
    for k in got[4]:
        kind = k.split(".")[-2] + "." + k.split(".")[-1]
        close_scaled("heads_train.grad." + kind, got[4][k], ref[4][k], 2e-5, err_msg=f"{k} B={B}")
    # the padded slots of x are exact zeros, as the padded ATen layout gives them
    for j, flt in enumerate(agent.filters):
        assert float(got[0][:, j, flt.get_num_filter_parameters():].abs().max() if flt.get_num_filter_parameters() < pw else 0.0) == 0.0


def test_agent_training_step_uses_the_heads_kernels_and_matches_the_aten_path():
    """Agent.forward in training mode end to end (trunks, heads, tail, filters) with ADAISP_HEADS_KERNEL on and off: the same
    selections, retouched images and parameter gradients to fp32 rounding."""
    from _margins import close_scaled
    from adaptiveisp_amd import heads_train
    cfg, agent = _agent()
    agent.feature_extractor.droupout.p = agent.action_selection.droupout.p = 0.0
    B = 4
    g = torch.Generator(device=DEV).manual_seed(7)
    x = torch.rand(B, 3, 64, 96, generator=g, device=DEV)
    z = torch.rand(B, cfg.z_dim, generator=g, device=DEV)
    st = torch.zeros(B, cfg.num_state_dim, device=DEV)
    outs, calls = {}, []
    orig = heads_train.heads
    heads_train.heads = lambda *a: (calls.append(1), orig(*a))[1]
    try:
        for sw in ("1", "0"):
            os.environ["ADAISP_HEADS_KERNEL"] = sw
            agent.zero_grad(set_to_none=True)
            (ret, ns, sur, pen), dbg, _ = agent((x, z, st), 0.3)
            (ret.mean() + sur.sum() + pen.sum()).backward()
            outs[sw] = (ret.detach().clone(), ns.detach().clone(), sur.detach().clone(), pen.detach().clone(), dbg["selected_filter"].clone(),
                        {n: p.grad.clone() for n, p in agent.named_parameters() if p.grad is not None})
    finally:
        heads_train.heads = orig
        del os.environ["ADAISP_HEADS_KERNEL"]
    assert len(calls) == 1
    a, b = outs["1"], outs["0"]
    assert torch.equal(a[4], b[4]) and torch.equal(a[1], b[1])
    for name, u, v in zip(("retouch", "surrogate", "penalty"), (a[0], a[2], a[3]), (b[0], b[2], b[3])):
        close_scaled("heads_train.step." + name, u, v, 2e-5)
    assert a[5].keys() == b[5].keys()
    for k in a[5]:
        close_scaled("heads_train.step.grad", a[5][k], b[5][k], 2e-4, err_msg=k)


def test_heads_kernels_refuse_what_they_do_not_serve():
    import ctypes

    from adaptiveisp_amd import _lib, heads_train
    cfg, agent = _agent()
    ff = torch.randn(9, 4096, device=DEV)                                     # more images than the kernels hold in LDS
    assert not heads_train.serves(agent, ff, ff)
    assert not heads_train.serves(agent, ff[:4, :4000].contiguous(), ff[:4, :4000].contiguous())
    a = heads_train._HeadsArgs()
    assert _lib.load().adaisp_heads_fwd(ctypes.byref(a), None) != 0 and _lib.load().adaisp_heads_bwd(None, None) != 0
