"""Agent / Value on the MI355X (heads in PyTorch-ROCm, pixels in HIP) against the reference's golden vectors."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _agent():
    from _synth import synth_state_dict
    from adaptiveisp_amd.agent import Agent
    from adaptiveisp_amd.config import cfg
    dev = torch.device("cuda:0")
    ag = Agent(cfg, shape=(6 + len(cfg.filters), 64, 64), device=dev)
    ag.load_state_dict(synth_state_dict(ag, seed=0))
    return ag.to(dev).eval(), cfg, dev


@pytest.fixture(autouse=True)
def _fp32_convs():
    old = torch.backends.cudnn.allow_tf32
    torch.backends.cudnn.allow_tf32 = False
    yield
    torch.backends.cudnn.allow_tf32 = old


@pytest.mark.parametrize("k", range(10))
def test_teacher_forced_step(golden, k):
    g = golden("agent")
    ag, cfg, dev = _agent()
    with torch.no_grad():
        (x, ns, sur, pen), dbg, _ = ag((T(g["x"]).to(dev), T(g["z"]).to(dev), T(g["s0"]).to(dev)), 1.0,
                                       selected_filter_id=k)
    assert np.array_equal(ns.cpu().numpy(), g[f"forced{k}.new_states"])
    np.testing.assert_allclose(pen.cpu().numpy(), g[f"forced{k}.penalty"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(dbg["filter_debug_info"][k]["filter_parameters"].reshape(-1).cpu().numpy(),
                               g[f"forced{k}.param0"], rtol=1e-4, atol=1e-5)
    # the heads run on MIOpen/rocBLAS (different summation order than the CPU reference): parameters agree to
    # ~1e-5, which the filters amplify slightly
    np.testing.assert_allclose(x.cpu().numpy(), g[f"forced{k}.x"], rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("tag", ["s0", "s1"])
def test_policy_selection(golden, tag):
    g = golden("agent")
    ag, cfg, dev = _agent()
    with torch.no_grad():
        (x, ns, sur, pen), dbg, _ = ag((T(g["x"]).to(dev), T(g["z"]).to(dev), T(g[tag]).to(dev)),
                                       float(g[f"{tag}.progress"]))
    np.testing.assert_allclose(dbg["pdf"].cpu().numpy(), g[f"{tag}.pdf0"], rtol=1e-3, atol=1e-5)
    assert np.array_equal(dbg["selected_filter"].cpu().numpy(), g[f"{tag}.selected"])
    assert np.array_equal(ns.cpu().numpy(), g[f"{tag}.new_states"])
    np.testing.assert_allclose(x.cpu().numpy(), g[f"{tag}.x"], rtol=2e-4, atol=2e-5)


def test_value(golden):
    from _synth import synth_state_dict
    from adaptiveisp_amd.config import cfg
    from adaptiveisp_amd.value import Value
    g = golden("agent")
    dev = torch.device("cuda:0")
    va = Value(cfg, shape=(9 + len(cfg.filters), 64, 64))
    va.load_state_dict(synth_state_dict(va, seed=1))
    va = va.to(dev).eval()
    with torch.no_grad():
        np.testing.assert_allclose(va(T(g["x"]).to(dev), T(g["s0"]).to(dev)).cpu().numpy(), g["value.s0"], rtol=1e-3,
                                   atol=1e-4)


def test_filter_module_api(golden):
    """Filter.process / Filter.forward on device tensors, the reference's own entry points."""
    from adaptiveisp_amd.config import cfg
    from adaptiveisp_amd.isp import filters as F
    g = golden("filters")
    dev = torch.device("cuda:0")
    img = T(g["img"]).to(dev)
    for name, cls in (("T", F.ToneFilter), ("CCM", F.CCMFilter), ("NLM", F.DenoiseFilter), ("Shr", F.SharpenFilter)):
        f = cls(cfg, predict=False)
        p = f.filter_param_regressor(T(g[f"{name}.feat"]).to(dev))
        np.testing.assert_allclose(f.process(img, p).cpu().numpy(), g[f"{name}.process"], rtol=1e-5, atol=2e-6)
        low, high, dbg = f.forward(img, specified_parameter=p, high_res=img)
        np.testing.assert_allclose(low.cpu().numpy(), g[f"{name}.forward"], rtol=1e-5, atol=2e-6)
        assert torch.equal(low, high) and set(dbg) == {"filter_parameters", "mask"}
