"""Training-mode trunk kernels (csrc/isp_trunk_train.hip) against the module path they replace: nn.Conv2d ->
nn.BatchNorm2d (batch statistics) -> nn.LeakyReLU x 4 under torch autograd, fp32 (reference agent.py:26-60,
value.py:6-44 as train.py:258,282-283 runs them). Features, every parameter gradient, the input gradients the critic
needs, running statistics and num_batches_tracked; bit-reproducibility of two runs."""
import copy

import pytest
import torch

from _margins import NOTES, close_scaled

pytestmark = pytest.mark.gpu


def _trunk(cin, seed, dev, mid=32, out_dim=4096):
    from adaptiveisp_amd.nets import FeatureExtractor
    torch.manual_seed(seed)
    t = FeatureExtractor(shape=(cin, 64, 64), mid_channels=mid, output_dim=out_dim, dropout_prob=None).to(dev)
    with torch.no_grad():                                   # non-trivial affine parameters and running statistics
        for m in t.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.3, 0.3)
                m.running_mean.uniform_(-0.2, 0.2)
                m.running_var.uniform_(0.5, 2.0)
    return t.train()


def _inputs(B, S, seed, dev, grad=False):
    g = torch.Generator(device="cpu").manual_seed(seed)
    img = torch.rand((B, 3, 64, 64), generator=g).to(dev)
    sv = torch.rand((B, S), generator=g).to(dev) if S else None
    if grad:
        img.requires_grad_(True)
        if sv is not None:
            sv.requires_grad_(True)
    return img, sv


def _module_path(trunk, img, sv):
    x = img if sv is None else torch.cat([img, sv[:, :, None, None].expand(-1, -1, 64, 64)], dim=1)
    return trunk(x)


def _compare_params(tag, got, ref):
    gg, rr = dict(got.named_parameters()), dict(ref.named_parameters())
    for name, pg in gg.items():
        pr = rr[name]
        assert pg.grad is not None and pr.grad is not None, name
        if name.split(".")[-2] in ("0", "3", "6", "9") and name.endswith(".bias"):
            # a conv bias in front of a batch-statistics BatchNorm has gradient exactly 0 in exact arithmetic: both paths hold
            # the rounding noise of a sum of B x H x W conv-output gradients (measured against the layer's weight gradient)
            scale = max(1.0, float(rr[name[:-4] + "weight"].grad.abs().max()))
            NOTES.append(f"trunk_train.{tag} {name}: |grad| ours {float(pg.grad.abs().max()):.2e} module "
                         f"{float(pr.grad.abs().max()):.2e} (weight-gradient scale {scale:.2e})")
            assert float(pg.grad.abs().max()) < 1e-4 * scale and float(pr.grad.abs().max()) < 1e-4 * scale, name
            continue
        close_scaled(f"trunk_train.{tag}.grad", pg.grad, pr.grad, frac=2e-4, floor=1e-6, err_msg=name)
    for (name, bg), (_, br) in zip(got.named_buffers(), ref.named_buffers()):
        if name.endswith("num_batches_tracked"):
            assert int(bg) == int(br), name
        else:
            close_scaled(f"trunk_train.{tag}.running", bg, br, frac=1e-5, floor=1e-3, err_msg=name)


@pytest.mark.parametrize("B", [8, 3])
def test_agent_pair_matches_the_module_path(B):
    """Two trunks, one input (feature_extractor + action_selection of Agent, agent.py:97-101)."""
    from adaptiveisp_amd import trunk_train
    dev = torch.device("cuda:0")
    ours = [_trunk(16, 1, dev), _trunk(16, 2, dev)]
    refs = [copy.deepcopy(t) for t in ours]
    img, sv = _inputs(B, 13, 3, dev)
    assert all(trunk_train.serves(t, img, sv) for t in ours)
    feat = trunk_train.trunk_features(ours, [img], [sv])
    ref = torch.stack([_module_path(t, img, sv) for t in refs])
    assert feat.shape == ref.shape == (2, B, 4096)
    close_scaled("trunk_train.agent.features", feat, ref, frac=2e-5)
    torch.manual_seed(7)
    dfeat = torch.randn_like(ref)
    feat.backward(dfeat)
    ref.backward(dfeat)
    for o, r in zip(ours, refs):
        _compare_params("agent", o, r)


def test_critic_two_calls_share_one_parameter_set():
    """The critic's two calls of an iteration (value.py:64-81 called at train.py:282-283) as one node: statistics per call,
    running statistics updated call by call, parameter gradients summed, input gradients for the call that wants them."""
    from adaptiveisp_amd import trunk_train
    dev = torch.device("cuda:0")
    B = 8
    ours = _trunk(19, 5, dev)
    ref_t = copy.deepcopy(ours)
    img0, sv0 = _inputs(B, 16, 11, dev)
    img1, sv1 = _inputs(B, 16, 12, dev, grad=True)
    img1r, sv1r = img1.detach().clone().requires_grad_(True), sv1.detach().clone().requires_grad_(True)
    feat = trunk_train.trunk_features([ours, ours], [img0, img1], [sv0, sv1], share_params=True)
    ref = torch.stack([_module_path(ref_t, img0, sv0), _module_path(ref_t, img1r, sv1r)])
    close_scaled("trunk_train.critic.features", feat, ref, frac=2e-5)
    torch.manual_seed(9)
    dfeat = torch.randn_like(ref)
    feat.backward(dfeat)
    ref.backward(dfeat)
    _compare_params("critic", ours, ref_t)
    close_scaled("trunk_train.critic.grad_img", img1.grad, img1r.grad, frac=2e-4, floor=1e-9)
    close_scaled("trunk_train.critic.grad_svec", sv1.grad, sv1r.grad, frac=2e-4, floor=1e-9)


def test_two_runs_are_bit_identical():
    from adaptiveisp_amd import trunk_train
    dev = torch.device("cuda:0")
    outs = []
    for _ in range(2):
        t = _trunk(19, 5, dev)
        img0, sv0 = _inputs(8, 16, 11, dev)
        img1, sv1 = _inputs(8, 16, 12, dev, grad=True)
        feat = trunk_train.trunk_features([t, t], [img0, img1], [sv0, sv1], share_params=True)
        torch.manual_seed(9)
        feat.backward(torch.randn_like(feat))
        outs.append([feat.detach().clone(), img1.grad.clone(), sv1.grad.clone()] + [p.grad.clone() for p in t.parameters()]
                    + [b.clone() for b in t.buffers()])
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_large_batch_takes_the_streaming_statistics_kernels():
    """B x 32 x 32 > 8192 values per channel: the BatchNorm kernels re-read instead of keeping the channel in registers."""
    from adaptiveisp_amd import trunk_train
    dev = torch.device("cuda:0")
    ours = [_trunk(16, 1, dev)]
    refs = [copy.deepcopy(t) for t in ours]
    img, sv = _inputs(24, 13, 3, dev)
    feat = trunk_train.trunk_features(ours, [img], [sv])
    ref = torch.stack([_module_path(t, img, sv) for t in refs])
    close_scaled("trunk_train.b24.features", feat, ref, frac=2e-5)
    torch.manual_seed(7)
    dfeat = torch.randn_like(ref)
    feat.backward(dfeat)
    ref.backward(dfeat)
    _compare_params("b24", ours[0], refs[0])


def test_unserved_trunks_are_reported():
    from adaptiveisp_amd import trunk_train
    dev = torch.device("cuda:0")
    t = _trunk(16, 1, dev)
    img, sv = _inputs(4, 13, 3, dev)
    assert trunk_train.serves(t, img, sv)
    assert not trunk_train.serves(t, img, sv[:, :5])                     # channel count does not match the first conv
    assert not trunk_train.serves(t.eval(), img, sv)                     # eval mode: running statistics, the folded eval kernels
    t.train()
    sync = torch.nn.SyncBatchNorm.convert_sync_batchnorm(copy.deepcopy(t))
    assert not trunk_train.serves(sync, img, sv)
    with pytest.raises(Exception):
        trunk_train.trunk_features([sync], [img], [sv])


# ---- the critic's statistics + trunk as one node, and the TD arithmetic kernel (csrc/isp_rl_train.hip) --------------------
def _value_net(dev, seed=3):
    from adaptiveisp_amd.config import cfg
    from adaptiveisp_amd.value import Value
    torch.manual_seed(seed)
    return Value(cfg, shape=(9 + len(cfg.filters), 64, 64)).to(dev).train()


def _edge_images(B, H, seed, dev):
    """Images whose 64x64 pooling holds what the statistics' backward has rules for: saturated blocks (ties of max / min
    over channels at exactly 1.0 / 0.0), values outside [0,1], grey pixels (max == min)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.rand((B, 3, H, H), generator=g) * 1.3 - 0.15
    x[:, :, : H // 4, : H // 4] = 1.0
    x[:, :2, H // 4: H // 2, : H // 4] = 1.0
    x[:, :, : H // 4, H // 4: H // 2] = 0.0
    x[:, :, H // 2:, : H // 8] = x[:, :1, H // 2:, : H // 8]            # grey
    x[:, 0, -H // 8:, -H // 8:] = 0.75
    x[:, 1, -H // 8:, -H // 8:] = 0.25                                   # max + min == 1 exactly: the minimum's tie
    x[:, 2, -H // 8:, -H // 8:] = 0.5
    return x.to(dev)


@pytest.mark.parametrize("pair", [False, True])
def test_critic_node_matches_the_module_path(pair, monkeypatch):
    """Value.forward / forward_pair (statistics + trunk in one node) against the ATen path (value.py:61-81): values, the
    gradient at the full-resolution image (through the pooling), every parameter gradient, the buffers."""
    dev = torch.device("cuda:0")
    B, H, S = 8, 128, 13
    ours = _value_net(dev)
    ref = copy.deepcopy(ours)
    g = torch.Generator(device="cpu").manual_seed(5)
    img_a = torch.rand((B, 3, H, H), generator=g).to(dev)
    st_a, st_b = torch.rand((B, S), generator=g).to(dev), torch.rand((B, S), generator=g).to(dev)
    img_b = _edge_images(B, H, 6, dev).requires_grad_(True)
    img_br = img_b.detach().clone().requires_grad_(True)
    if pair:
        va, vb = ours.forward_pair(img_a, st_a, img_b, st_b)
    else:
        va, vb = ours(img_a, st_a), ours(img_b, st_b)
    monkeypatch.setenv("ADAISP_TRUNK_KERNELS", "0")
    ra, rb = ref(img_a, st_a), ref(img_br, st_b)
    monkeypatch.setenv("ADAISP_TRUNK_KERNELS", "1")
    close_scaled("trunk_train.value.out", torch.cat([va, vb]), torch.cat([ra, rb]), frac=2e-5)
    w = torch.linspace(-1.0, 1.0, 2 * B, device=dev)[:, None]
    (torch.cat([va, vb]) * w).sum().backward()
    (torch.cat([ra, rb]) * w).sum().backward()
    close_scaled("trunk_train.value.grad_image", img_b.grad, img_br.grad, frac=2e-4, floor=1e-12)
    for (name, pg), (_, pr) in zip(ours.named_parameters(), ref.named_parameters()):
        if name.startswith("feature_extractor"):
            continue
        close_scaled("trunk_train.value.grad_fc", pg.grad, pr.grad, frac=2e-4, floor=1e-9, err_msg=name)
    _compare_params("value", ours.feature_extractor, ref.feature_extractor)


def test_critic_statistics_kernel_alone():
    """adaisp_critic_planes_fwd / _bwd against Value._planes under autograd on the edge-case planes."""
    from adaptiveisp_amd import trunk_train
    dev = torch.device("cuda:0")
    B, S = 4, 5
    net = _value_net(dev)
    small = torch.nn.functional.adaptive_avg_pool2d(_edge_images(B, 128, 8, dev), (64, 64)).contiguous()
    st = torch.rand((B, S), device=dev)
    s_ref = small.clone().requires_grad_(True)
    _, ext_ref = net._planes(None, st, s_ref)
    L = trunk_train._lib.load()
    a = trunk_train._PlanesArgs()
    ext = torch.empty((B, S + 3), device=dev)
    a.G, a.B, a.n_state = 1, B, S
    a.small[0], a.states[0], a.svec[0] = small.data_ptr(), st.data_ptr(), ext.data_ptr()
    import ctypes
    assert L.adaisp_critic_planes_fwd(ctypes.byref(a), None) == 0
    torch.cuda.synchronize()
    assert torch.equal(ext[:, :S], st)
    close_scaled("trunk_train.planes.stats", ext[:, S:], ext_ref[:, S:], frac=2e-6)
    dext = torch.randn((B, S + 3), device=dev)
    ext_ref.backward(dext)
    base = torch.randn_like(small)
    dsmall = base.clone()
    a.dsvec[0], a.dsmall_in[0], a.dsmall[0] = dext.data_ptr(), dsmall.data_ptr(), dsmall.data_ptr()
    assert L.adaisp_critic_planes_bwd(ctypes.byref(a), None) == 0
    torch.cuda.synchronize()
    close_scaled("trunk_train.planes.grad", dsmall - base, s_ref.grad, frac=2e-5, floor=1e-12)


@pytest.mark.parametrize("use_td,use_truncated,use_penalty", [(True, True, True), (False, True, True), (True, False, False)])
def test_td_kernel_matches_the_elementwise_arithmetic(use_td, use_truncated, use_penalty, monkeypatch):
    """NOT the parity test of a16 (that is test_td_kernels_match_the_reference_fixture below: `td.npz`, generated by executing the
    reference's own statements of train.py:264-305). This one is product against product — adaisp_td_fwd / _bwd against
    rl.td_losses' element-wise ATen branch on random inputs — a consistency screen that keeps the two branches of OUR code equal."""
    from adaptiveisp_amd import rl
    from adaptiveisp_amd.config import cfg as base_cfg
    from adaptiveisp_amd.util import Dict
    dev = torch.device("cuda:0")
    cfg = Dict(base_cfg)
    cfg.use_TD, cfg.use_penalty, cfg.all_reward, cfg.detect_loss_weight, cfg.discount_factor = use_td, use_penalty, 0.7, 1.5, 0.9
    B, F = 16, len(base_cfg.filters)
    g = torch.Generator(device="cpu").manual_seed(3)
    r = lambda *sh: torch.rand(sh, generator=g)          # noqa: E731
    new_states = torch.cat([r(B, 1).round(), r(B, 1).round(), (r(B, 1) * 10).floor(), r(B, F).round()], dim=1).to(dev)
    retouch_mean = torch.tensor([0.005, 0.95] + [0.4] * (B - 2))[:, None].to(dev)
    leaves = {k: v.to(dev).requires_grad_(True) for k, v in dict(
        l_re=r(B, 1) * 1.2 - 0.1, penalty=r(B, 1), surrogate=-r(B, 1) * 3, old_value=r(B, 1) * 4 - 2, new_value=r(B, 1) * 4 - 2).items()}
    l_in = (r(B, 1) * 1.2 - 0.1).to(dev)
    outs = []
    for kernel in ("1", "0"):
        monkeypatch.setenv("ADAISP_TD_KERNEL", kernel)
        for v in leaves.values():
            v.grad = None
        o = rl.td_losses(cfg, l_in, leaves["l_re"], leaves["penalty"], leaves["surrogate"], new_states, leaves["old_value"],
                         leaves["new_value"], retouch_mean, use_truncated=use_truncated, max_bri=0.9)
        torch.autograd.backward([o["value_loss"] * 0.5, o["agent_loss"] * 2.0])
        outs.append(({k: o[k].detach().clone() for k in o},
                     {k: (torch.zeros_like(v) if v.grad is None else v.grad.clone()) for k, v in leaves.items()}))
    for k in outs[0][0]:
        close_scaled(f"trunk_train.td.{k}", outs[0][0][k], outs[1][0][k], frac=2e-6, floor=1e-6)
    for k in leaves:
        close_scaled(f"trunk_train.td.grad_{k}", outs[0][1][k], outs[1][1][k], frac=2e-6, floor=1e-9)


# ---- the agent's training step with trunk + tail kernels against the ATen path ---------------------------------------------
def _train_agent(dev):
    from _synth import synth_state_dict
    from adaptiveisp_amd.agent import Agent
    from adaptiveisp_amd.config import cfg
    ag = Agent(cfg, shape=(6 + len(cfg.filters), 64, 64), device=dev)
    ag.load_state_dict(synth_state_dict(ag, seed=0))
    ag = ag.to(dev).train()
    ag.feature_extractor.droupout.p = ag.action_selection.droupout.p = 0.0      # (the two paths draw different masks)
    return ag, cfg


@pytest.mark.parametrize("forced", [None, 9, 4])
def test_agent_training_step_kernels_match_the_module_path(forced, monkeypatch):
    """Agent.forward in train mode (agent.py:88-285) with the trunk pair and the policy tail on the HIP kernels, against the
    same step on ATen (ADAISP_TRUNK_KERNELS=0, ADAISP_POLICY_TAIL_KERNEL=0): identical selections, the outputs and — through a
    loss on (retouch, surrogate, penalty) — every parameter gradient."""
    from _synth import test_image
    dev = torch.device("cuda:0")
    ours, cfg = _train_agent(dev)
    ref = copy.deepcopy(ours)
    B = 8
    x = torch.from_numpy(test_image(B, 96, 128, seed=5, special=False)).to(dev)
    g = torch.Generator().manual_seed(3)
    z = torch.rand(B, cfg.z_dim, generator=g).to(dev)
    z[0, 0], z[1, 0] = 0.0, 0.999999                                    # the all-zero one-hot and the last filter
    states = torch.zeros(B, cfg.num_state_dim, device=dev)
    states[:, 2] = torch.arange(B, device=dev) % 5
    states[:, 3:] = (torch.rand(B, len(cfg.filters), generator=g) < 0.3).float().to(dev)
    w = torch.rand(x.shape, generator=g).to(dev)

    def run(ag):
        (y, ns, sur, pen), dbg, _ = ag((x, z, states), 0.25, selected_filter_id=forced)
        loss = (y * w).sum() * 1e-3 + (sur * torch.linspace(-1, 1, B, device=dev)[:, None]).sum() + pen.sum() * 0.7
        loss.backward()
        return y.detach(), ns.detach(), sur.detach(), pen.detach(), dbg

    yo, nso, so, po, dbo = run(ours)
    monkeypatch.setenv("ADAISP_TRUNK_KERNELS", "0")
    monkeypatch.setenv("ADAISP_POLICY_TAIL_KERNEL", "0")
    yr, nsr, sr, pr, dbr = run(ref)
    assert torch.equal(dbo["selected_filter"], dbr["selected_filter"])
    if forced is None:
        assert int(dbo["selected_filter"][0]) == -1 and len(set(dbo["selected_filter"].tolist())) > 2
    assert torch.equal(nso, nsr)
    close_scaled("trunk_train.agent_step.pdf", dbo["pdf"], dbr["pdf"], frac=2e-5)
    close_scaled("trunk_train.agent_step.retouch", yo, yr, frac=2e-5)
    close_scaled("trunk_train.agent_step.surrogate", so, sr, frac=2e-5)
    close_scaled("trunk_train.agent_step.penalty", po, pr, frac=2e-5)
    for i in range(len(ours.filters)):
        a, b = dbo["filter_debug_info"][i]["filter_parameters"], dbr["filter_debug_info"][i]["filter_parameters"]
        close_scaled("trunk_train.agent_step.params", a, b, frac=2e-5, err_msg=ours.filters[i].get_short_name())
    gg, rr = dict(ours.named_parameters()), dict(ref.named_parameters())
    for name, p in gg.items():
        if "fc_mask" in name:
            continue
        is_conv_bias = ("feature_extractor" in name or "action_selection" in name) and name.endswith(".bias") and \
            name.split(".")[-2] in ("0", "3", "6", "9")
        if is_conv_bias:
            continue                                                     # (exactly 0 in exact arithmetic: rounding noise on both sides)
        assert (p.grad is None) == (rr[name].grad is None), name
        if p.grad is not None:
            close_scaled("trunk_train.agent_step.grad", p.grad, rr[name].grad, frac=5e-4, floor=1e-7, err_msg=name)


@pytest.mark.parametrize("shape", [(8, 3, 512, 512), (3, 3, 30, 50), (2, 3, 63, 65)])
def test_image_stats(shape):
    """adaisp_image_stats: per-image mean and non-finite count (rl.retouch_stats), against torch on the same batch."""
    from adaptiveisp_amd import _lib, rl
    dev = torch.device("cuda:0")
    x = torch.rand(shape, generator=torch.Generator().manual_seed(2)).to(dev)
    x[1, 0, 3, 4:7] = float("nan")
    x[1, 2, 5, 1] = float("inf")
    st = _lib.image_stats(x)
    assert st.shape == (shape[0], 2)
    assert st[:, 1].tolist() == [0.0, 4.0] + [0.0] * (shape[0] - 2)
    ref = rl.retouch_stats(x.cpu())
    assert torch.equal(st[:, 1].cpu(), ref[:, 1]) and not torch.isfinite(st[1, 0])
    keep = [i for i in range(shape[0]) if i != 1]
    close_scaled("trunk_train.image_stats.mean", st[keep, 0], x[keep].double().mean(dim=(1, 2, 3)).float(), frac=2e-6)
    assert torch.equal(torch.nan_to_num(st, nan=-1.0), torch.nan_to_num(_lib.image_stats(x), nan=-1.0))


@pytest.mark.parametrize("hw", [(255, 255), (257, 257)])
def test_image_stats_tail_elements_of_odd_sizes(hw):
    """n % 4 != 0 with floor(n / 4) a multiple of the 64 chunks (3 x 255 x 255 = 195075): the last 1 - 3 values of every image
    are summed and checked for non-finite values too (they were not; the replay guard could miss a NaN there)."""
    from adaptiveisp_amd import _lib
    dev = torch.device("cuda:0")
    x = torch.rand((3, 3) + hw, generator=torch.Generator().manual_seed(4)).to(dev)
    x[0].view(-1)[-1] = 1000.0                       # a value the mean cannot miss
    x[1].view(-1)[-1] = float("nan")
    x[2].view(-1)[-3] = float("inf")
    st = _lib.image_stats(x)
    assert st[:, 1].tolist() == [0.0, 1.0, 1.0]
    assert abs(st[0, 0].item() - x[0].double().mean().item()) < 2e-6 * x[0].double().mean().item()


@pytest.mark.parametrize("max_norm", [1e-5, 1e3])
def test_clip_adam_kernels_match_torch(max_norm):
    """adaisp_clip_adam_step (optim.clip_adam_step) against torch.nn.utils.clip_grad_norm_ + torch.optim.Adam(fused=True).step()
    on the same gradients, step after step: parameters, both moments, step counts; tensors of odd sizes, larger than a chunk,
    one without a gradient. The first step creates the optimizer state through torch on both sides."""
    from adaptiveisp_amd import optim as aoptim
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(4)
    shapes = [(128, 4096), (7,), (33, 5, 3), (4099,), (256, 64, 4, 4), (1,), (12289,)]

    def make():
        ps = [torch.nn.Parameter(torch.randn(s, generator=torch.Generator().manual_seed(i)).to(dev)) for i, s in enumerate(shapes)]
        ps.append(torch.nn.Parameter(torch.zeros(5, device=dev)))                       # never receives a gradient
        return ps, torch.optim.Adam(ps, lr=3e-4, fused=True)

    (pa, oa), (pb, ob) = make(), make()
    for it in range(6):
        for q, r in zip(pa[:-1], pb[:-1]):
            gr = (torch.randn(q.shape, generator=g) * (10.0 ** (it % 3 - 1))).to(dev)
            q.grad, r.grad = gr.clone(), gr.clone()
        torch.nn.utils.clip_grad_norm_(pa, max_norm)
        oa.step()
        handled = aoptim.clip_adam_step(ob, max_norm)
        assert handled == (it > 0)
        if not handled:
            torch.nn.utils.clip_grad_norm_(pb, max_norm)
            ob.step()
        for i, (q, r) in enumerate(zip(pa[:-1], pb[:-1])):
            sa, sb = oa.state[q], ob.state[r]
            assert torch.equal(sa["step"], sb["step"])
            close_scaled("trunk_train.adam.param", r, q, frac=2e-6, err_msg=f"step {it} tensor {i}")
            close_scaled("trunk_train.adam.exp_avg", sb["exp_avg"], sa["exp_avg"], frac=2e-6, floor=1e-12, err_msg=f"step {it} tensor {i}")
            close_scaled("trunk_train.adam.exp_avg_sq", sb["exp_avg_sq"], sa["exp_avg_sq"], frac=2e-6, floor=1e-20, err_msg=f"step {it} tensor {i}")
        assert pb[-1].grad is None and len(ob.state.get(pb[-1], {})) == 0
    ws = ob.__dict__["_adaisp_table"]["ws"]
    total = torch.sqrt(sum((r.grad.double() ** 2).sum() for r in pb[:-1]))
    close_scaled("trunk_train.adam.total_norm", ws[-1:], total.float().reshape(1), frac=2e-6)
    assert abs(float(ws[-2]) - min(1.0, max_norm / (float(total) + 1e-6))) <= 2e-6 * max(1.0, float(ws[-2]))
    # options the kernels do not serve are left to torch
    assert not aoptim.clip_adam_step(torch.optim.Adam(pa, lr=1e-3, weight_decay=0.1, fused=True), 1.0)
    assert not aoptim.clip_adam_step(torch.optim.SGD(pa, lr=1e-3), 1.0)


def test_clip_adam_nan_norm_scheduler_bookkeeping_and_hooks():
    """(1) a non-finite gradient makes the total norm NaN and — like torch's clamp in clip_grad_norm_ — every
    gradient's coefficient NaN, so ALL parameters go NaN (fminf alone stepped the finite ones unclipped); (2) the kernel path
    records what Optimizer.step's wrapper records (`_opt_called`, `_step_count`): no LambdaLR warning after a resume;
    (3) an optimizer with registered step hooks is left to torch (the kernels never call opt.step())."""
    import warnings
    from adaptiveisp_amd import optim as aoptim
    dev = torch.device("cuda:0")

    def make():
        ps = [torch.nn.Parameter(torch.randn(300, 40, device=dev)), torch.nn.Parameter(torch.randn(17, device=dev))]
        opt = torch.optim.Adam(ps, lr=1e-3, fused=True)
        for p in ps:
            p.grad = torch.randn_like(p)
        torch.nn.utils.clip_grad_norm_(ps, 1.0)
        opt.step()                                                   # creates the state: the next step is the kernels'
        return ps, opt
    ps, opt = make()
    fresh = torch.optim.Adam(ps, lr=1e-3, fused=True)               # as after a resume: state loaded, step() never called
    fresh.load_state_dict(copy.deepcopy(opt.state_dict()))
    sched = torch.optim.lr_scheduler.LambdaLR(fresh, lambda it: 0.5 ** it)
    for p in ps:
        p.grad = torch.randn_like(p)
    assert aoptim.clip_adam_step(fresh, 1.0)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        sched.step()                                                 # would warn if optimizer.step() had "not been called"
    assert fresh._opt_called
    # (3)
    hooked = torch.optim.Adam(ps, lr=1e-3, fused=True)
    hooked.load_state_dict(copy.deepcopy(opt.state_dict()))
    hooked.register_step_post_hook(lambda o, a, k: None)
    assert not aoptim.clip_adam_step(hooked, 1.0)
    # (1)
    ps, opt = make()
    for p in ps:
        p.grad = torch.randn_like(p)
    ps[1].grad[3] = float("nan")
    assert aoptim.clip_adam_step(opt, 1e-5)
    torch.cuda.synchronize()
    assert all(torch.isnan(p).all() for p in ps)


def test_clip_adam_follows_a_reloaded_optimizer_state():
    """load_state_dict replaces exp_avg / exp_avg_sq / step: the kernels' cached address table is rebuilt, the update lands in
    the NEW tensors (what a resumed checkpoint relies on)."""
    from adaptiveisp_amd import optim as aoptim
    dev = torch.device("cuda:0")
    ps = [torch.nn.Parameter(torch.randn(300, 40, device=dev)), torch.nn.Parameter(torch.randn(17, device=dev))]
    opt = torch.optim.Adam(ps, lr=1e-3, fused=True)
    for it in range(3):
        for p in ps:
            p.grad = torch.randn_like(p)
        if not aoptim.clip_adam_step(opt, 1.0):
            torch.nn.utils.clip_grad_norm_(ps, 1.0)
            opt.step()
    saved = copy.deepcopy(opt.state_dict())
    old_m = opt.state[ps[0]]["exp_avg"]
    opt.load_state_dict(saved)
    new_m = opt.state[ps[0]]["exp_avg"]
    assert new_m.data_ptr() != old_m.data_ptr()
    before_old, before_new = old_m.clone(), new_m.clone()
    for p in ps:
        p.grad = torch.randn_like(p)
    assert aoptim.clip_adam_step(opt, 1.0)
    torch.cuda.synchronize()
    assert torch.equal(old_m, before_old) and not torch.equal(new_m, before_new)
    assert float(opt.state[ps[0]]["step"]) == 4.0


@pytest.mark.parametrize("case", range(8))
def test_td_kernels_match_the_reference_fixture(case, golden):
    """a16, reward / TD half, pinned on the reference itself: adaisp_td_fwd / adaisp_td_bwd (through rl.td_losses on the
    device) against td.npz — the statements of train.py:264-305 executed on seeded tensors by tests/golden/gen_golden.py
    (gen_td) — for every (use_TD, use_truncated, use_penalty) setting: reward, q_value, both losses, and autograd's
    gradients of each loss w.r.t. the retouch detection loss, penalty, surrogate and both critic values."""
    import numpy as np
    from test_rl_math import td_case, td_run
    g = golden("td")
    tag = f"c{case}"
    c, leaves, fixed, use_truncated, max_bri = td_case(g, tag, device="cuda:0")
    from adaptiveisp_amd import rl
    assert rl._td_kernel_serves(fixed["l_in"], *leaves.values(), fixed["retouch_mean"])        # the HIP path is the one that runs
    out, grads = td_run(c, leaves, fixed, use_truncated, max_bri)
    for k in ("reward", "q_value", "value_loss", "agent_loss"):
        np.testing.assert_allclose(out[k].detach().cpu().numpy(), g[f"{tag}.out.{k}"], rtol=2e-6, atol=2e-6, err_msg=f"{tag} {k}")
    want = -g[f"{tag}.out.advantage"] if g[f"{tag}.switches"][0] else g[f"{tag}.out.q_value"] - g[f"{tag}.old_value"]
    np.testing.assert_allclose(out["advantage"].detach().cpu().numpy(), want, rtol=2e-6, atol=2e-5, err_msg=f"{tag} advantage")
    for k, v in grads.items():
        ref = g[f"{tag}.{k}"]
        np.testing.assert_allclose(v.cpu().numpy(), ref, rtol=2e-6, atol=2e-6 * max(float(np.abs(ref).max()), 1e-6),
                                   err_msg=f"{tag} {k}")
