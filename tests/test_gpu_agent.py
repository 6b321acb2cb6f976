"""Agent / Value on the MI355X (heads in PyTorch-ROCm, pixels in HIP) against the reference's golden vectors."""
import numpy as np
import pytest
import torch

from _margins import close

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
    close("teacher_forced_step#1", pen.cpu().numpy(), g[f"forced{k}.penalty"], rtol=1e-4, atol=1e-5)
    close(f"teacher_forced_step:param:f{k}", dbg["filter_debug_info"][k]["filter_parameters"].reshape(-1).cpu().numpy(), g[f"forced{k}.param0"], rtol=1e-4, atol=1e-5)
    # the heads run on MIOpen/rocBLAS (different summation order than the CPU reference): parameters agree to
    # ~1e-5, which the filters amplify slightly
    close(f"teacher_forced_step:x:f{k}", x.cpu().numpy(), g[f"forced{k}.x"], rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("tag", ["s0", "s1"])
def test_policy_selection(golden, tag):
    g = golden("agent")
    ag, cfg, dev = _agent()
    with torch.no_grad():
        (x, ns, sur, pen), dbg, _ = ag((T(g["x"]).to(dev), T(g["z"]).to(dev), T(g[tag]).to(dev)),
                                       float(g[f"{tag}.progress"]))
    close("policy_selection#1", dbg["pdf"].cpu().numpy(), g[f"{tag}.pdf0"], rtol=1e-3, atol=1e-5)
    assert np.array_equal(dbg["selected_filter"].cpu().numpy(), g[f"{tag}.selected"])
    assert np.array_equal(ns.cpu().numpy(), g[f"{tag}.new_states"])
    close("policy_selection#2", x.cpu().numpy(), g[f"{tag}.x"], rtol=2e-4, atol=2e-5)


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
        close("value#1", va(T(g["x"]).to(dev), T(g["s0"]).to(dev)).cpu().numpy(), g["value.s0"], rtol=1e-3, atol=1e-4)


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
        close("filter_module_api#1", f.process(img, p).cpu().numpy(), g[f"{name}.process"], rtol=1e-5, atol=2e-6)
        low, high, dbg = f.forward(img, specified_parameter=p, high_res=img)
        close("filter_module_api#2", low.cpu().numpy(), g[f"{name}.forward"], rtol=1e-5, atol=2e-6)
        assert torch.equal(low, high) and set(dbg) == {"filter_parameters", "mask"}


@pytest.mark.parametrize("tag", ["s0", "s1"])
def test_fused_eval_path_matches_torch_path(golden, tag):
    """The 7-launch fused policy step (policy_fast.py) against the PyTorch head path and the golden vectors."""
    g = golden("agent")
    ag, cfg, dev = _agent()
    inp = (T(g["x"]).to(dev), T(g["z"]).to(dev), T(g[tag]).to(dev))
    prog = float(g[f"{tag}.progress"])
    with torch.no_grad():
        ag.use_fast_eval = True
        (xf, nsf, surf, penf), dbgf, _ = ag(inp, prog)
        assert ag._fast is not None                                   # the fused path really ran
        ag.use_fast_eval = False
        (xt, nst, surt, pent), dbgt, _ = ag(inp, prog)
    assert torch.equal(dbgf["selected_filter"], dbgt["selected_filter"]) and dbgf["selected_filter"].dtype == torch.int64
    assert np.array_equal(dbgf["selected_filter"].cpu().numpy(), g[f"{tag}.selected"])
    assert torch.equal(nsf, nst) and np.array_equal(nsf.cpu().numpy(), g[f"{tag}.new_states"])
    close("fused_eval_path_matches_torch_path#1", dbgf["pdf"], dbgt["pdf"], rtol=1e-4, atol=1e-6)
    close("fused_eval_path_matches_torch_path#2", dbgf["pdf"].cpu().numpy(), g[f"{tag}.pdf0"], rtol=1e-3, atol=1e-5)
    close("fused_eval_path_matches_torch_path#3", surf, surt, rtol=1e-4, atol=1e-5)
    close("fused_eval_path_matches_torch_path#4", penf, pent, rtol=1e-4, atol=1e-5)
    close("fused_eval_path_matches_torch_path#5", penf.cpu().numpy(), g[f"{tag}.penalty"], rtol=1e-4, atol=1e-5)
    for a, b in zip(dbgf["filter_debug_info"], dbgt["filter_debug_info"]):
        assert a["filter_parameters"].shape == b["filter_parameters"].shape
        close("fused_eval_path_matches_torch_path#6", a["filter_parameters"], b["filter_parameters"], rtol=1e-4, atol=1e-5)
    close("fused_eval_path_matches_torch_path#7", xf.cpu().numpy(), g[f"{tag}.x"], rtol=2e-4, atol=2e-5)


def test_fused_eval_forced_and_highres(golden):
    g = golden("agent")
    ag, cfg, dev = _agent()
    inp = (T(g["x"]).to(dev), T(g["z"]).to(dev), T(g["s0"]).to(dev))
    with torch.no_grad():
        for k in range(10):
            (x, ns, sur, pen), dbg, _ = ag(inp, 1.0, selected_filter_id=k)
            assert np.array_equal(ns.cpu().numpy(), g[f"forced{k}.new_states"])
            close("fused_eval_forced_and_highres#1", pen.cpu().numpy(), g[f"forced{k}.penalty"], rtol=1e-4, atol=1e-5)
            close(f"fused_eval_forced:x:f{k}", x.cpu().numpy(), g[f"forced{k}.x"], rtol=2e-4, atol=2e-5)
        (x, ns, hr), _, _ = ag(inp, 1.0, high_res=T(g["hr.in"]).to(dev), selected_filter_id=5)
        close("fused_eval_forced_and_highres#3", hr.cpu().numpy(), g["hr.out"], rtol=2e-4, atol=2e-5)
        # `out=`: the retouched batch lands in the caller's buffer (the bench's double-buffered hand-over), same values
        buf = torch.full_like(inp[0], float("nan"))
        (x2, _, _, _), _, _ = ag(inp, 1.0, selected_filter_id=3, out=buf)
        (x3, _, _, _), _, _ = ag(inp, 1.0, selected_filter_id=3)
        assert x2.data_ptr() == buf.data_ptr() and torch.equal(buf, x3)
        with pytest.raises(ValueError):
            ag(inp, 1.0, selected_filter_id=3, out=buf[:, :, :-1])
        # the two halves of the eval step (pooling + policy, then the filter launch) are the step
        for k in (None, 4):
            plan = ag.plan_step(inp, 1.0, selected_filter_id=k)
            (xr, nsr, surr, penr), dbgr, _ = ag(inp, 1.0, selected_filter_id=k)
            assert torch.equal(ag.apply_step(inp[0], plan), xr) and torch.equal(plan["new_states"], nsr)
            assert torch.equal(plan["selected"], dbgr["selected_filter"]) and torch.equal(plan["penalty"], penr)
            carried = {"op_ids": plan["op_ids"].clone(), "packed": plan["packed"].clone()}      # what a caller must carry
            assert torch.equal(ag.apply_step(inp[0], carried, out=buf), xr) and torch.equal(buf, xr)
        # weights are re-snapshotted after an in-place update
        with torch.no_grad():
            ag.fc2.bias.add_(torch.tensor([0, 0, 0, 0, 0, 0, 0, 0, 50.0, 0], device=dev))
        (_, _, _, _), dbg, _ = ag(inp, 1.0)
        assert (dbg["selected_filter"] == 8).all()


def test_training_iteration_end_to_end():
    """One RL optimisation step on the device: policy heads (PyTorch) -> HIP ISP forward -> frozen detector with
    autograd -> per-sample detection loss -> TD losses -> HIP parameter-gradient kernels -> clip -> Adam."""
    from _synth import synth_state_dict, synth_yolo_state_dict, test_image
    from adaptiveisp_amd import dist as adist
    from adaptiveisp_amd.agent import Agent
    from adaptiveisp_amd.config import cfg
    from adaptiveisp_amd.rl import train_iteration
    from adaptiveisp_amd.value import Value
    from adaptiveisp_amd.yolo import yolov3
    from adaptiveisp_amd.yolo.loss import DetectionLoss, default_hyp
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    agent = Agent(cfg, shape=(16, 64, 64), device=dev)
    agent.load_state_dict(synth_state_dict(agent, seed=0))
    agent = agent.to(dev).train()
    value = Value(cfg, shape=(19, 64, 64))
    value.load_state_dict(synth_state_dict(value, seed=1))
    value = value.to(dev).train()
    det = yolov3()
    det.load_state_dict(synth_yolo_state_dict(det))
    det = det.to(dev).train()
    for m in det.modules():                                   # frozen reward model: BN statistics fixed, no grads
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()
    for p in det.parameters():
        p.requires_grad_(False)
    loss_fn = DetectionLoss(det.model[-1].anchors, nc=80, hyp=default_hyp(80, 96), device=dev)
    B = 4
    imgs = T(test_image(B, 64, 96, seed=3, special=False)).to(dev)
    z = torch.rand(B, cfg.z_dim, device=dev) * 0.98 + 0.01
    states = torch.zeros(B, cfg.num_state_dim, device=dev)
    labels = [torch.tensor([[0, 1 + b, 0.5, 0.5, 0.3, 0.4]]) for b in range(B)]
    opts = [torch.optim.Adam(agent.parameters(), lr=3e-5), torch.optim.Adam(value.parameters(), lr=3e-5)]
    before = torch.cat([p.detach().reshape(-1) for p in agent.parameters()]).clone()
    out = train_iteration(cfg, agent, value, det, loss_fn, imgs, z, states, labels, 0.1, opts,
                          buckets=[adist.GradBucket(agent, value)])
    torch.cuda.synchronize()
    for k in ("reward", "q_value", "value_loss", "agent_loss"):
        assert torch.isfinite(out[k]).all(), k
    after = torch.cat([p.detach().reshape(-1) for p in agent.parameters()])
    assert (after != before).any()                            # the heads moved
    assert out["retouch"].shape == imgs.shape and float(out["retouch"].min()) >= 0 and float(out["retouch"].max()) <= 1


def test_device_side_sampling_train_mode(golden):
    """k_finish with train_mode=1: the inverse-CDF sampler of agent.py:12-16 on the device. The selection must equal
    `pdf_sample` (pinned to the reference bit-exactly by tests/golden/select.npz on the CPU) applied to the pdf the
    same launch returned; u = 0 gives id -1 -> all-zero one-hot -> op ZERO -> an all-zero image and an unchanged usage
    vector; u just below 1 gives the last filter with non-zero probability."""
    from adaptiveisp_amd import _lib
    from adaptiveisp_amd.agent import one_hot, pdf_sample
    from adaptiveisp_amd.policy_fast import FastPolicy
    from _synth import test_image
    g = golden("select")
    ag, cfg, dev = _agent()
    B = g["u"].shape[0]
    x = T(test_image(B, 40, 56, seed=21, special=False)).to(dev)
    z = torch.rand(B, cfg.z_dim, generator=torch.Generator().manual_seed(3)).to(dev)
    z[:, 0] = T(g["u"][:, 0]).to(dev)                     # includes u = 0, u = 1 - ulp
    states = torch.zeros(B, cfg.num_state_dim, device=dev)
    states[:, 3:] = (torch.rand(B, 10, generator=torch.Generator().manual_seed(4)) < 0.3).float().to(dev)
    fp = FastPolicy(ag)
    o = fp.run(_lib.pool64(x), z, states, 0.25, None, train_mode=True)
    torch.cuda.synchronize()
    pdf = o["pdf"].cpu()
    want = pdf_sample(pdf, z[:, 0:1].cpu()).to(torch.int64)
    sel = o["selected"].cpu()
    assert torch.equal(sel, want), (sel, want)
    assert sel[0].item() == -1 and sel[1].item() == 9
    assert len(set(sel.tolist())) > 2, "sampling should not collapse onto the arg-max"
    ops = torch.tensor([-1] + [int(f.op_code) for f in ag.filters], dtype=torch.int32)
    assert torch.equal(o["op_ids"].cpu(), ops[sel + 1])
    hot = one_hot(10, sel).float()
    ns = o["new_states"].cpu()
    assert torch.equal(ns[:, 3:], torch.maximum(states[:, 3:].cpu(), hot))
    assert torch.equal(ns[:, 2], states[:, 2].cpu() + 1)
    sur = torch.sum(hot * torch.log(pdf + 1e-10), dim=1, keepdim=True)
    close("device_side_sampling_train_mode#1", o["surrogate"].cpu(), sur, rtol=1e-5, atol=1e-6)
    y = _lib.forward(x, o["op_ids"], o["packed"], clip=True)
    assert not y[0].any() and y[1].any()
    # the same launch in eval mode is the arg-max
    e = fp.run(_lib.pool64(x), z, states, 0.25, None, train_mode=False)
    assert torch.equal(e["selected"].cpu(), torch.argmax(e["pdf"].cpu(), dim=1))


@pytest.mark.parametrize("shape", [(2, 72, 100), (1, 30, 50), (2, 128, 64), (1, 720, 1280)])
def test_pool64_backward_matches_aten(shape):
    """adaisp_pool64_backward against autograd of nn.AdaptiveAvgPool2d((64,64)) (the reference's down_sample,
    agent.py:85 / value.py:61): same window rule, same g/kh/kw, same accumulation order -> a few ulp at most."""
    from adaptiveisp_amd import _lib
    from adaptiveisp_amd.nets import Pool64
    dev = torch.device("cuda:0")
    B, H, W = shape
    g = torch.Generator().manual_seed(H + W)
    x = torch.rand(B, 3, H, W, generator=g).to(dev).requires_grad_(True)
    go = torch.randn(B, 3, 64, 64, generator=g).to(dev)
    y = Pool64()(x)
    y.backward(go)
    xr = x.detach().cpu().requires_grad_(True)
    torch.nn.AdaptiveAvgPool2d((64, 64))(xr).backward(go.cpu())
    close("pool64_backward_matches_aten#1", x.grad.cpu(), xr.grad, rtol=2e-6, atol=1e-9)
    assert torch.equal(y.detach(), _lib.pool64(x.detach()))


@pytest.mark.parametrize("k", range(10))
def test_critic_to_actor_gradient_matches_reference(golden, k):
    """train.py:281-305 with cfg.use_TD: agent_loss contains -V(retouch, new_states), so the critic back-propagates
    through the 64x64 pooling and the selected filter into that filter's heads. Golden: the reference's autograd of
    L = -mean(V(retouch, new_states)) (tests/golden/gen_golden.py::gen_value_path), teacher-forced filter k."""
    from _synth import synth_state_dict
    from adaptiveisp_amd.config import cfg
    from adaptiveisp_amd.value import Value
    g = golden("value_path")
    ag, _, dev = _agent()
    va = Value(cfg, shape=(9 + len(cfg.filters), 64, 64))
    va.load_state_dict(synth_state_dict(va, seed=1))
    va = va.to(dev).eval()
    ag.zero_grad(set_to_none=True)
    (xo, ns, sur, pen), dbg, _ = ag((T(g["x"]).to(dev), T(g["z"]).to(dev), T(g["s0"]).to(dev)), 1.0, selected_filter_id=k)
    assert xo.requires_grad
    v = va(xo, ns)
    close(f"critic_to_actor_value:f{k}", v.detach().cpu().numpy(), g[f"f{k}.value"], rtol=1e-3, atol=1e-4)
    (-v.mean()).backward()
    flt = ag.filters[k]
    for got, key in ((flt.fc_filter.weight.grad, "gw"), (flt.fc_filter.bias.grad, "gb")):
        ref = g[f"f{k}.{key}"]
        assert got is not None and np.abs(ref).max() > 0
        close(f"critic_to_actor_gradient:f{k}", got.cpu().numpy(), ref, rtol=2e-3, atol=2e-3 * np.abs(ref).max())
    for j, other in enumerate(ag.filters):                 # only the selected filter's heads see this gradient
        if j != k and other.fc_filter.bias.grad is not None:
            assert not other.fc_filter.bias.grad.any()


def test_config1_single_640_three_steps_on_the_hip_path(oracle_mod):
    """BASELINE config 1 (one 640x640 frame, steps=3, no detector) on the device: every step of the fused eval path
    equals the oracle's filter applied with the parameters the heads regressed on the device."""
    from adaptiveisp_amd import _lib
    ag, cfg, dev = _agent()
    rng = np.random.default_rng(1234)
    x = torch.from_numpy((rng.random((1, 3, 640, 640)) ** 2.2 * 0.5).astype(np.float32)).to(dev)
    z = torch.from_numpy(rng.random((1, cfg.z_dim)).astype(np.float32)).to(dev)
    st = torch.zeros(1, cfg.num_state_dim, device=dev)
    ops = {0: _lib.OP_EXPOSURE, 9: _lib.OP_WB, 5: _lib.OP_TONE, 4: _lib.OP_NLM}
    with torch.no_grad():
        for step, k in enumerate((0, 4, 5)):
            (y, st2, _, pen), dbg, _ = ag((x, z, st), 1.0, selected_filter_id=k)
            assert ag._fast is not None
            p = dbg["filter_debug_info"][k]["filter_parameters"].reshape(1, -1).cpu().numpy()
            ref = oracle_mod.forward(x.cpu().numpy(), ops[k], p, clip=True)
            close("config1_single_640_three_steps_on_the_hip_path#1", y.cpu().numpy(), ref, rtol=1e-5, atol=2e-6)
            assert float(st2[0, 2]) == step + 1 and float(st2[0, 3 + k]) == 1.0 and int(dbg["selected_filter"][0]) == k
            x, st = y, st2


def test_eval_loop_reuses_the_pooling_of_the_filter_launch(golden):
    """Agent.forward in an eval loop (val_adaptiveisp.py:293-304 feeds the returned image back in): from the second step on
    the policy reads the 64x64 planes the previous filter launch wrote. Same selections, states, pdf and pixels as the
    loop that pools every input with a launch of its own — and an image modified in between is pooled again."""
    g = golden("agent")
    ag, cfg, dev = _agent()
    z = T(g["z"]).to(dev)

    def loop(use_cache, forced):
        x, st = T(g["x"]).to(dev), T(g["s0"]).to(dev)
        trace = []
        with torch.no_grad():
            for k in range(4):
                if not use_cache:
                    ag._pool_cache = None
                hit = ag._cached_pool(x) is not None
                (x, st, sur, pen), dbg, _ = ag((x, z, st), 0.5, selected_filter_id=forced[k] if forced else None)
                trace.append((x.clone(), st.clone(), dbg["selected_filter"].clone(), dbg["pdf"].clone(), pen.clone(), hit))
        return trace

    for forced in (None, [4, 3, 0, 5]):
        a, b = loop(True, forced), loop(False, forced)
        assert [t[5] for t in a] == [False, True, True, True] and not any(t[5] for t in b)
        for ta, tb in zip(a, b):
            assert all(torch.equal(u, v) for u, v in zip(ta[:5], tb[:5]))
    # a caller that edits the returned image in place gets a fresh pooling, not the cached one
    with torch.no_grad():
        (x, st, _, _), _, _ = ag((T(g["x"]).to(dev), z, T(g["s0"]).to(dev)), 0.5)
        assert ag._cached_pool(x) is not None
        x.mul_(0.5)
        assert ag._cached_pool(x) is None
        from adaptiveisp_amd import _lib
        (y, _, _, _), _, _ = ag((T(g["x"]).to(dev), z, T(g["s0"]).to(dev)), 0.5)
        _lib.process(_lib.OP_EXPOSURE, T(g["x"]).to(dev), torch.ones(y.shape[0], 1, device=dev), out=y)   # raw-pointer write
        assert ag._cached_pool(y) is None
        # the two switches for callers whose writers go around torch AND this package
        (y, _, _, _), _, _ = ag((T(g["x"]).to(dev), z, T(g["s0"]).to(dev)), 0.5)
        assert ag._cached_pool(y) is not None
        ag.forget_pooled_planes()
        assert ag._cached_pool(y) is None
        (y, _, _, _), _, _ = ag((T(g["x"]).to(dev), z, T(g["s0"]).to(dev)), 0.5)
        ag.reuse_pooled_planes = False
        assert ag._cached_pool(y) is None


def test_eval_loop_under_inference_mode(golden):
    """The reference's eval entry runs under `torch.inference_mode()` (val_adaptiveisp.py:104 `@smart_inference_mode()`):
    inference tensors have no version counter, so the pooled-plane cache must stand aside — same selections, states and
    pixels as the `no_grad` loop, every step pooled by a launch of its own."""
    g = golden("agent")
    ag, cfg, dev = _agent()
    z = T(g["z"]).to(dev)

    def loop(ctx):
        trace = []
        with ctx():
            x, st = T(g["x"]).to(dev), T(g["s0"]).to(dev)
            zz = z.clone()
            for k in range(4):
                (x, st, sur, pen), dbg, _ = ag((x, zz, st), 0.5)
                trace.append((x.clone(), st.clone(), dbg["selected_filter"].clone(), dbg["pdf"].clone(), pen.clone()))
                if ctx is torch.inference_mode:
                    assert torch.is_inference(x) and ag._cached_pool(x) is None
        return trace

    a, b = loop(torch.inference_mode), loop(torch.no_grad)
    for ta, tb in zip(a, b):
        assert all(torch.equal(u.clone(), v) for u, v in zip(ta, tb))
    # a normal tensor handed in while inference mode is on still has its version: the cache keeps working for it
    with torch.no_grad():
        (x, st, _, _), _, _ = ag((T(g["x"]).to(dev), z, T(g["s0"]).to(dev)), 0.5)
    with torch.inference_mode():
        assert ag._cached_pool(x) is not None
