"""Detector training path on HIP: element-wise training kernels against PyTorch, and the whole backward (raw head
maps -> gradient of the input image) against fp32 autograd of the plain module tree."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def test_silu_fwd_bwd_kernels():
    from adaptiveisp_amd.yolo import _lib
    L, st = _lib.load(), _lib.stream_ptr
    g = torch.Generator().manual_seed(0)
    B, H, W, C = 2, 5, 7, 24
    pre = (torch.randn(B, H, W, C, generator=g) * 2).to(torch.bfloat16).to(DEV)
    res = torch.randn(B, H, W, C, generator=g).to(torch.bfloat16).to(DEV)
    wide = torch.zeros(B, H, W, C + 16, dtype=torch.bfloat16, device=DEV)           # write into a channel slice
    out = wide[..., 8:8 + C]
    _lib.check(L.adayolo_silu_fwd(_p(pre), C, _p(res), C, ctypes.c_void_p(wide.data_ptr() + 16), C + 16, B * H * W, C, st()), "silu")
    torch.cuda.synchronize()
    ref = F.silu(pre.float()).to(torch.bfloat16).float() + res.float()
    assert (out.float() - ref).abs().max() <= 2e-2 * max(1.0, ref.abs().max().item())
    assert wide[..., :8].abs().sum() == 0 and wide[..., 8 + C:].abs().sum() == 0
    gy = torch.randn(B, H, W, C, generator=g).to(torch.bfloat16).to(DEV)
    gp = torch.empty_like(gy)
    gres = torch.randn(B, H, W, C, generator=g).to(torch.bfloat16).to(DEV)
    gres0 = gres.clone()
    _lib.check(L.adayolo_silu_bwd(_p(gy), C, _p(pre), C, _p(gp), C, _p(gres), C, 1, B * H * W, C, st()), "dsilu")
    torch.cuda.synchronize()
    x = pre.float().requires_grad_(True)
    F.silu(x).backward(gy.float())
    assert (gp.float() - x.grad).abs().max() <= 2e-2 * max(1.0, x.grad.abs().max().item())
    assert (gres.float() - (gres0.float() + gy.float())).abs().max() <= 3e-2 * 4


def test_zero_insert_and_upsample_bwd():
    from adaptiveisp_amd.yolo import _lib
    L, st = _lib.load(), _lib.stream_ptr
    g = torch.Generator().manual_seed(1)
    B, Ho, Wo, C = 2, 3, 5, 8
    x = torch.randn(B, Ho, Wo, C, generator=g).to(torch.bfloat16).to(DEV)
    u = torch.full((B, 2 * Ho, 2 * Wo, C), 7.0, dtype=torch.bfloat16, device=DEV)
    _lib.check(L.adayolo_zero_insert2x(_p(x), C, _p(u), C, B, Ho, Wo, 2 * Ho, 2 * Wo, C, st()), "zins")
    ref = torch.zeros_like(u)
    ref[:, 0::2, 0::2] = x
    assert torch.equal(u, ref)
    gy = torch.randn(B, 2 * Ho, 2 * Wo, C, generator=g).to(torch.bfloat16).to(DEV)
    gx = torch.ones(B, Ho, Wo, C, dtype=torch.bfloat16, device=DEV)
    _lib.check(L.adayolo_upsample2x_bwd(_p(gy), C, _p(gx), C, 1, B, Ho, Wo, C, st()), "upbwd")
    r = gy.float().view(B, Ho, 2, Wo, 2, C).sum((2, 4)) + 1.0
    assert (gx.float() - r).abs().max() <= 3e-2 * max(1.0, r.abs().max().item())


def test_dgrad_is_conv_with_flipped_weights():
    """The identity the backward engine relies on, checked on the conv kernel itself (stride 1 and 2)."""
    from adaptiveisp_amd.yolo import _lib
    L, st = _lib.load(), _lib.stream_ptr
    g = torch.Generator().manual_seed(2)
    for (H, W, cin, cout, k, s) in [(12, 20, 16, 32, 3, 1), (12, 20, 32, 16, 1, 1), (12, 20, 16, 32, 3, 2)]:
        B = 2
        w = (torch.randn(cout, k, k, cin, generator=g) / (k * k * cin) ** 0.5).to(torch.bfloat16).to(DEV)
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        dP = torch.randn(B, Ho, Wo, cout, generator=g).to(torch.bfloat16).to(DEV)
        x = torch.zeros(B, cin, H, W, device=DEV, requires_grad=True)
        y = F.conv2d(x, w.float().permute(0, 3, 1, 2), stride=s, padding=k // 2)
        y.backward(dP.float().permute(0, 3, 1, 2))
        ref = x.grad.permute(0, 2, 3, 1)
        wt = w.flip(1, 2).permute(3, 1, 2, 0).contiguous()
        g_in = dP
        if s == 2:
            g_in = torch.empty(B, H, W, cout, dtype=torch.bfloat16, device=DEV)
            _lib.check(L.adayolo_zero_insert2x(_p(dP), cout, _p(g_in), cout, B, Ho, Wo, H, W, cout, st()), "zins")
        out = torch.empty(B, H, W, cin, dtype=torch.bfloat16, device=DEV)
        zb = torch.zeros(cin, device=DEV)
        _lib.check(L.adayolo_conv_fwd(_p(g_in), cout, _p(wt), _p(zb), None, 0, _p(out), cin, B, H, W, cout, cin, k, 1, 0, st()), "dgrad")
        torch.cuda.synchronize()
        assert (out.float() - ref).abs().max() <= 2e-2 * max(1.0, ref.abs().max().item()), (H, W, cin, cout, k, s)


@pytest.mark.parametrize("B,H,W", [(2, 64, 96), (1, 72, 128)])
def test_backward_to_image_vs_fp32_autograd(B, H, W):
    from _synth import synth_yolo_state_dict, test_image
    from adaptiveisp_amd.yolo import YoloTrainEngine, yolov3
    det = yolov3()
    det.load_state_dict(synth_yolo_state_dict(det, seed=2))
    det = det.to(DEV).train()
    for m in det.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()
    for p in det.parameters():
        p.requires_grad_(False)
    eng = YoloTrainEngine(det, B, H, W, device=DEV)
    x = torch.from_numpy(test_image(B, H, W, seed=61, special=False)).to(DEV)
    Hp = (H + 31) // 32 * 32
    top = (Hp - H) // 2
    g = torch.Generator().manual_seed(5)
    # reference: fp32 module tree on the letterboxed image
    xr = x.clone().requires_grad_(True)
    boxed = torch.full((B, 3, Hp, W), 114.0 / 255.0, device=DEV)
    boxed = torch.cat([boxed[:, :, :top], xr, boxed[:, :, top + H:]], 2) if Hp != H else xr
    raws_ref = det(boxed)
    R = [torch.randn(r.shape, generator=g).to(DEV) for r in raws_ref]
    sum((r * w).sum() for r, w in zip(raws_ref, R)).backward()
    # HIP path
    xh = x.clone().requires_grad_(True)
    raws = eng(xh)
    for a, b in zip(raws, raws_ref):
        assert (a - b.detach()).abs().max() <= 5e-2 * max(1.0, b.abs().max().item())
    sum((r * w).sum() for r, w in zip(raws, R)).backward()
    torch.cuda.synchronize()
    gh, gr = xh.grad, xr.grad
    assert torch.isfinite(gh).all()
    rel = ((gh - gr).norm() / gr.norm()).item()
    cos = F.cosine_similarity(gh.reshape(1, -1), gr.reshape(1, -1)).item()
    # what bf16 activations / gradients cost through 75 layers, measured with an INDEPENDENT bf16 implementation of the same
    # network: the PyTorch module tree under autocast (MIOpen bf16 convs, bf16 activations, fp32 master weights)
    xa = x.clone().requires_grad_(True)
    boxed_a = torch.cat([boxed.detach()[:, :, :top], xa, boxed.detach()[:, :, top + H:]], 2) if Hp != H else xa
    with torch.autocast("cuda", dtype=torch.bfloat16):
        raws_a = det(boxed_a)
    sum((r.float() * w).sum() for r, w in zip(raws_a, R)).backward()
    rel_a = ((xa.grad - gr).norm() / gr.norm()).item()
    cos_a = F.cosine_similarity(xa.grad.reshape(1, -1), gr.reshape(1, -1)).item()
    msg = (f"detector image gradient vs fp32 autograd at {B}x{H}x{W}: HIP engine rel {rel:.4f} cos {cos:.6f} | "
           f"torch autocast-bf16 rel {rel_a:.4f} cos {cos_a:.6f}")
    print(msg)
    import _margins
    _margins.NOTES.append(msg)
    # measured on the MI355X (profiles/round3_parity_margins.txt): HIP engine 1.24-1.26 % / cosine 0.99992, the independent
    # bf16 run 1.64-1.68 % / 0.99989 — the engine (fp32 accumulation in every kernel) is closer to fp32 than MIOpen's bf16
    assert rel < 0.025 and cos > 0.9995, (rel, cos)
    assert rel <= 1.25 * rel_a, (rel, rel_a)
    # a stale backward (another forward ran in between) must fail loudly, not use overwritten buffers
    xs = x.clone().requires_grad_(True)
    stale = eng(xs)
    eng.forward_train(x)
    with pytest.raises(RuntimeError):
        sum(r.sum() for r in stale).backward()


def _labels(B, g, dup=False, dense=0):
    """Per-image [n,6] labels (image, class, x, y, w, h) with 0..4 boxes; `dup`: two boxes of different classes on the same
    spot (the same cells / anchors are matched twice: last-match-wins objectness target, summed gradients)."""
    out = []
    for b in range(B):
        n = dense if dense else int(torch.randint(0, 5, (1,), generator=g))
        t = torch.zeros(n, 6)
        t[:, 1] = torch.randint(0, 80, (n,), generator=g).float()
        t[:, 2:4] = torch.rand(n, 2, generator=g) * 0.8 + 0.1
        t[:, 4:6] = torch.rand(n, 2, generator=g) * 0.5 + 0.04
        if dup and n:
            extra = t[:1].clone()
            extra[:, 1] = (extra[:, 1] + 7) % 80
            extra[:, 4:6] *= 1.05
            t = torch.cat([t, extra], 0)
        out.append(t)
    if all(t.shape[0] == 0 for t in out):
        out[0] = torch.tensor([[0, 3, 0.5, 0.5, 0.3, 0.4]])
    return out


@pytest.mark.parametrize("B,H,W,dup", [(2, 64, 96, False), (3, 96, 96, True), (8, 512, 512, True), (4, 128, 160, 120)])
def test_fused_detection_loss_matches_the_pytorch_loss(B, H, W, dup):
    """csrc/yolo_loss.hip (one forward launch, two backward launches on the bf16 head maps) against
    loss.batched_per_sample_loss — the PyTorch restatement of ComputeLossBatch that tests/test_yolo_cpu.py pins on the
    reference's own numbers (detloss.npz) — on the SAME engine: loss values to 2e-5, the head-map gradients element by
    element (both round to bf16: equal up to one bf16 ulp of rounding ties), the image gradient, bit-reproducible."""
    from _synth import synth_yolo_state_dict, test_image
    from adaptiveisp_amd.yolo import YoloTrainEngine, yolov3
    from adaptiveisp_amd.yolo.loss import DetectionLoss, assign_labels, batched_per_sample_loss, default_hyp, pack_assigned
    det = yolov3()
    det.load_state_dict(synth_yolo_state_dict(det, seed=2))
    det = det.to(DEV).train()
    for m in det.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()
    for p in det.parameters():
        p.requires_grad_(False)
    eng = YoloTrainEngine(det, B, H, W, device=DEV)
    loss_fn = DetectionLoss(det.model[-1].anchors, nc=80, hyp=default_hyp(80, max(H, W)), device=DEV)
    g = torch.Generator().manual_seed(B * 1000 + H)
    # (dup = 120: 120 boxes per image — more than 2048 matches per layer, so the kernels read the match list from memory
    # instead of LDS, and cells matched many times over)
    labels = _labels(B, g, bool(dup), dense=dup if dup not in (False, True) else 0)
    x = torch.from_numpy(test_image(B, H, W, seed=17, special=False)).to(DEV)
    wgt = (torch.rand(B, 1, generator=g) + 0.5).to(DEV)                     # upstream gradient of the per-image losses

    # PyTorch loss on fp32 copies of the engine's raw maps, evaluated on the CPU: a cell matched twice gets its objectness
    # target from a sequential index assignment there (the last match wins — what the kernels implement); on the GPU
    # torch's index_put_ leaves the winner undefined
    cpu_fn = DetectionLoss(det.model[-1].anchors.cpu(), nc=80, hyp=default_hyp(80, max(H, W)), device="cpu")
    raws = [r.cpu().requires_grad_(True) for r in eng.forward_train(x)]
    nthreads = torch.get_num_threads()
    torch.set_num_threads(1)                 # (above its grain size torch's CPU index_put_ is split over threads: not sequential either)
    try:
        assigned = assign_labels(cpu_fn, raws, labels)
        l_ref = batched_per_sample_loss(cpu_fn, raws, labels, assigned)
        (l_ref * wgt.cpu()).sum().backward()
    finally:
        torch.set_num_threads(nthreads)
    g_maps_ref = [r.grad.to(DEV) for r in raws]
    g_img_ref = eng.backward_image(g_maps_ref).clone()
    l_ref = l_ref.detach().to(DEV)

    # fused path
    packed = pack_assigned(assign_labels(loss_fn, eng.head_shapes(), labels))
    if dup not in (False, True):
        assert max(idx.shape[0] for idx, _ in packed) > 2048
    for (idx, box), m in zip(packed, assigned):
        assert idx.shape[0] == m["b"].shape[0] and idx.dtype == torch.int32 and box.shape[1] == 6 and idx.is_cuda
    with torch.no_grad():
        l_ng = eng.per_sample_loss(loss_fn, x, packed)
    xh = x.clone().requires_grad_(True)
    l_hip = eng.per_sample_loss(loss_fn, xh, packed)
    assert torch.equal(l_hip.detach(), l_ng)                                 # bit-reproducible, with or without autograd
    # (fp32 sums in another order: 2e-5 for a handful of matches per image, 1e-4 where a mean runs over thousands of terms)
    torch.testing.assert_close(l_hip.detach(), l_ref, rtol=2e-5 if dup in (False, True) else 1e-4, atol=1e-6)
    (l_hip * wgt).sum().backward()
    torch.cuda.synchronize()
    # head-map gradients as the kernels left them in the engine's bf16 buffers
    worst = 0.0
    for gv, v, gref in zip(eng._graw, eng.raw, g_maps_ref):
        got = gv.buf[..., : eng.na * eng.no].float().view(B, v.H, v.W, eng.na, eng.no).permute(0, 3, 1, 2, 4)
        want = gref.to(torch.bfloat16).float()                               # what backward_image feeds the detector's backward
        assert (gv.buf[..., eng.na * eng.no:] == 0).all()
        scale = want.abs().max().item()
        err = (got - want).abs().max().item()
        worst = max(worst, err / max(scale, 1e-30))
        # one bf16 ulp (2^-8 relative) of each element, plus the fp32 summation-order noise of the per-image means
        assert ((got - want).abs() <= want.abs() * 2 ** -7 + 1e-6 * scale).all(), (err, scale)
    rel = ((xh.grad - g_img_ref).norm() / g_img_ref.norm()).item()
    assert rel < 2e-2, (rel, worst)
    if dup:                                                                  # the duplicate-cell paths were exercised
        per_layer = [[tuple(r) for r in idx[:, :4].cpu().tolist()] for idx, _ in packed]
        assert any(len(k) != len(set(k)) for k in per_layer)
    # stale backward fails loudly here too
    xs = x.clone().requires_grad_(True)
    stale = eng.per_sample_loss(loss_fn, xs, packed)
    eng.forward_train(x)
    with pytest.raises(RuntimeError):
        stale.sum().backward()


@pytest.mark.parametrize("variant", [5, 22, 26, 27, 60, 80, 85])
def test_conv_keep_fwd_equals_conv_then_silu(variant):
    """adayolo_conv_keep_fwd (one launch: pre-activation stored, activation applied to its bf16 value, residual added) ==
    adayolo_conv_fwd_variant(ACT_NONE) into `pre` + adayolo_silu_fwd, bit for bit, on every shape the kernel serves."""
    from adaptiveisp_amd.yolo import _lib
    L = _lib.load()
    st = _lib.stream_ptr()
    served = 0
    for (B, H, W, cin, cout, k, s, use_res, act) in [(2, 16, 16, 512, 1024, 3, 1, True, 1), (8, 64, 64, 128, 256, 3, 1, False, 1),
                                                    (3, 33, 17, 256, 128, 1, 1, True, 1), (2, 32, 32, 256, 512, 3, 2, False, 1),
                                                    (1, 9, 7, 128, 128, 3, 1, True, 0), (2, 40, 24, 64, 128, 1, 1, False, 1)]:
        g = torch.Generator().manual_seed(H * 7 + cin)
        x = torch.randn(B, H, W, cin, generator=g).to(torch.bfloat16).to(DEV)
        w = (torch.randn(cout, k, k, cin, generator=g) / (k * k * cin) ** 0.5).to(torch.bfloat16).to(DEV)
        b = torch.randn(cout, generator=g).to(DEV)
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        res = torch.randn(B, Ho, Wo, cout, generator=g).to(torch.bfloat16).to(DEV) if use_res else None
        nan = lambda: torch.full((B, Ho, Wo, cout), float("nan"), dtype=torch.bfloat16, device=DEV)  # noqa: E731
        pre1, out1, pre2, out2 = nan(), nan(), nan(), nan()
        rc = L.adayolo_conv_keep_fwd(_p(x), cin, _p(w), _p(b), _p(res) if use_res else None, cout if use_res else 0, _p(out1), cout,
                                     _p(pre1), cout, B, H, W, cin, cout, k, s, act, variant, st)
        if rc == -2:                                    # this kernel does not take the shape: the engine keeps two launches
            continue
        assert rc == 0, (rc, variant)
        assert L.adayolo_conv_fwd_variant(_p(x), cin, _p(w), _p(b), None, 0, _p(pre2), cout, B, H, W, cin, cout, k, s, 0, variant, st) == 0
        if act:
            assert L.adayolo_silu_fwd(_p(pre2), cout, _p(res) if use_res else None, cout if use_res else 0, _p(out2), cout,
                                      B * Ho * Wo, cout, st) == 0
        else:
            out2 = pre2 if not use_res else (pre2.float() + res.float()).to(torch.bfloat16)
        torch.cuda.synchronize()
        assert torch.equal(pre1.view(torch.int16), pre2.view(torch.int16)), (variant, B, H, W, cin, cout)
        assert torch.equal(out1.view(torch.int16), out2.view(torch.int16)), (variant, B, H, W, cin, cout)
        served += 1
    assert served >= 3, (variant, served)
    # not a kernel with the second output
    assert L.adayolo_conv_keep_fwd(_p(x), cin, _p(w), _p(b), None, 0, _p(out1), cout, _p(pre1), cout, B, H, W, cin, cout, k, s, 1, 50, st) == -1


def test_graph_replay_and_fused_train_forward_change_nothing():
    """The training engine at the config-4 per-rank shape with the tuning table: (a) launch sequences replayed from
    hipGraphs, (b) conv + SiLU pairs of the forward replaced by adayolo_conv_keep_fwd where the tuned kernel has the
    second output, (c) the SiLU' launches of the backward absorbed by the data-gradient conv that completes their input
    (adayolo_conv_dsilu_fwd; the Bottleneck's gradient copy read in place) — against the same engine with all three
    switched off: raw head maps and image gradient bit for bit."""
    import os
    from _synth import synth_yolo_state_dict, test_image
    from adaptiveisp_amd.yolo import YoloTrainEngine, yolov3
    cache = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "adaptiveisp_amd", "yolo", "tuning", "mi355x.json")
    det = yolov3()
    det.load_state_dict(synth_yolo_state_dict(det, seed=2))
    det = det.to(DEV).train()
    for p in det.parameters():
        p.requires_grad_(False)
    B, H, W = 8, 512, 512
    x = torch.from_numpy(test_image(B, H, W, seed=5, special=False)).to(DEV)
    g = torch.Generator().manual_seed(9)
    results = []
    for graph, keep in (("0", "0"), ("1", "1")):
        os.environ["ADAYOLO_TRAIN_GRAPH"], os.environ["ADAYOLO_TRAIN_KEEP"] = graph, keep
        os.environ["ADAYOLO_TRAIN_FUSE_DSILU"] = keep                   # (c) SiLU' inside the conv that completes its gradient
        try:
            eng = YoloTrainEngine(det, B, H, W, device=DEV)
            eng.autotune(cache=cache, write=False)
            outs = []
            for rep in range(2):                                        # second pass = a replay when graphs are on
                xr = x.clone().requires_grad_(True)
                raws = eng(xr)
                if rep == 0:
                    R = [torch.randn(r.shape, generator=torch.Generator().manual_seed(3 + i)).to(DEV) for i, r in enumerate(raws)]
                sum((r * w).sum() for r, w in zip(raws, R)).backward()
                torch.cuda.synchronize()
                outs.append(([r.detach().clone() for r in raws], xr.grad.clone()))
            assert all(torch.equal(a, b) for a, b in zip(outs[0][0], outs[1][0])) and torch.equal(outs[0][1], outs[1][1])
            if keep == "1":
                assert eng.keep_fused >= 20, eng.keep_fused              # most Conv layers of the forward are single launches
                assert eng._graphs["fwd"] is not None and eng._graphs["bwd"] is not None
                # the SiLU' launches whose producing conv runs on a kernel with that epilogue are gone (the rest stay)
                assert eng.dsilu_fused >= 20, eng.dsilu_fused
                assert sum(1 for e in eng._backward_plan() if e[0] == "dsilu") == 72 - eng.dsilu_fused
            else:
                assert eng.dsilu_fused == 0 and sum(1 for e in eng._backward_plan() if e[0] == "dsilu") == 72
            results.append(outs[1])
        finally:
            os.environ.pop("ADAYOLO_TRAIN_GRAPH", None)
            os.environ.pop("ADAYOLO_TRAIN_KEEP", None)
            os.environ.pop("ADAYOLO_TRAIN_FUSE_DSILU", None)
    (raw_a, grad_a), (raw_b, grad_b) = results
    for a, b in zip(raw_a, raw_b):
        assert torch.equal(a, b)
    assert torch.equal(grad_a, grad_b)


@pytest.mark.parametrize("variant", [5, 22, 26, 27, 60, 80, 85, 104])
def test_conv_dsilu_fwd_equals_conv_then_silu_bwd(variant):
    """adayolo_conv_dsilu_fwd (one launch: g = conv + residual rounded to bf16, optionally stored; grad_pre = g * silu'(pre))
    == adayolo_conv_fwd_variant(ACT_NONE) + adayolo_silu_bwd, bit for bit, with and without the stored gradient, on
    channel slices, on every shape the kernel serves (the split-K variant with its workspace)."""
    from adaptiveisp_amd.yolo import _lib
    L = _lib.load()
    st = _lib.stream_ptr()
    served = 0
    for (B, H, W, cin, cout, k, use_res) in [(8, 16, 16, 1024, 512, 3, True), (8, 32, 32, 512, 256, 3, False), (3, 33, 17, 256, 128, 1, True),
                                             (1, 9, 7, 128, 128, 3, False), (2, 40, 24, 128, 64, 1, True), (2, 24, 40, 64, 32, 3, False)]:
        ws, nws = None, 0
        if variant >= 100:
            nws = int(L.adayolo_conv_splitk_workspace_bytes(B, H, W, cin, cout, k, 1, variant))
            if nws == 0:
                continue
            ws = torch.zeros(nws, dtype=torch.uint8, device=DEV)
        g = torch.Generator().manual_seed(H * 7 + cin + variant)
        x = torch.randn(B, H, W, cin, generator=g).to(torch.bfloat16).to(DEV)
        w = (torch.randn(cout, k, k, cin, generator=g) / (k * k * cin) ** 0.5).to(torch.bfloat16).to(DEV)
        zb = torch.zeros(cout, device=DEV)
        pre = (torch.randn(B, H, W, cout + 8, generator=g) * 2).to(torch.bfloat16).to(DEV)[..., 8:]      # a channel slice
        res = torch.randn(B, H, W, cout, generator=g).to(torch.bfloat16).to(DEV) if use_res else None
        nan = lambda c=cout: torch.full((B, H, W, c), float("nan"), dtype=torch.bfloat16, device=DEV)  # noqa: E731
        gy_ref, gp_ref = nan(), nan()
        if variant >= 100:
            rc = L.adayolo_conv_splitk_fwd(_p(x), cin, _p(w), _p(zb), _p(res) if use_res else None, cout if use_res else 0, _p(gy_ref), cout,
                                           None, 0, B, H, W, cin, cout, k, 1, 0, variant, _p(ws), nws, st)
        else:
            rc = L.adayolo_conv_fwd_variant(_p(x), cin, _p(w), _p(zb), _p(res) if use_res else None, cout if use_res else 0, _p(gy_ref), cout,
                                            B, H, W, cin, cout, k, 1, 0, variant, st)
        assert rc == 0
        assert L.adayolo_silu_bwd(_p(gy_ref), cout, _p(pre), cout + 8, _p(gp_ref), cout, None, 0, 0, B * H * W, cout, st) == 0
        for store in (True, False):
            gy, wide = nan(), nan(cout + 16)
            gp = wide[..., 8:8 + cout]
            rc = L.adayolo_conv_dsilu_fwd(_p(x), cin, _p(w), _p(zb), _p(res) if use_res else None, cout if use_res else 0,
                                          _p(gy) if store else None, cout if store else 0, _p(pre), cout + 8, _p(gp), cout + 16,
                                          B, H, W, cin, cout, k, 1, variant, _p(ws) if ws is not None else None, nws, st)
            if rc == -2:
                break
            assert rc == 0, (rc, variant)
            torch.cuda.synchronize()
            assert torch.equal(gp.contiguous().view(torch.int16), gp_ref.view(torch.int16)), (variant, B, H, W, cin, cout, store)
            if store:
                assert torch.equal(gy.view(torch.int16), gy_ref.view(torch.int16))
            assert torch.isnan(wide[..., :8].float()).all() and torch.isnan(wide[..., 8 + cout:].float()).all()   # neighbours untouched
        else:
            served += 1
    assert served >= 2, (variant, served)
    # a kernel without this epilogue / missing pointers
    assert L.adayolo_conv_dsilu_fwd(_p(x), cin, _p(w), _p(zb), None, 0, None, 0, _p(pre), cout + 8, _p(gp), cout + 16, B, H, W, cin, cout,
                                    k, 1, 50, None, 0, st) == -1
    assert L.adayolo_conv_dsilu_fwd(_p(x), cin, _p(w), _p(zb), None, 0, None, 0, None, 0, _p(gp), cout + 16, B, H, W, cin, cout,
                                    k, 1, 5, None, 0, st) == -1


@pytest.mark.parametrize("variant", [5, 27, 60, 104])
def test_stride2_data_gradient_without_zero_insertion(variant):
    """adayolo_conv_s2grad_fwd (2x2 conv over the output grid, depth-to-space stores) against fp32 autograd of the stride-2
    conv, with a residual, with the fused SiLU' (grad_in stored and not), on channel slices; ragged pixel counts."""
    from adaptiveisp_amd.yolo import _lib
    L = _lib.load()
    st = _lib.stream_ptr()
    served = 0
    for (B, Ho, Wo, cin, cout, use_res) in [(8, 16, 16, 512, 1024, True), (2, 24, 40, 64, 128, False), (3, 9, 11, 128, 256, True),
                                           (8, 32, 32, 256, 512, False), (1, 5, 7, 32, 64, True)]:
        H, W = 2 * Ho, 2 * Wo
        ws, nws = None, 0
        if variant >= 100:
            nws = int(L.adayolo_conv_splitk_workspace_bytes(B, Ho, Wo, cout, 4 * cin, 2, 1, variant))
            if nws == 0:
                continue
            ws = torch.zeros(nws, dtype=torch.uint8, device=DEV)
        g = torch.Generator().manual_seed(Ho * 7 + cin + variant)
        w = (torch.randn(cout, 3, 3, cin, generator=g) / (9 * cin) ** 0.5).to(torch.bfloat16).to(DEV)
        gy = torch.randn(B, Ho, Wo, cout, generator=g).to(torch.bfloat16).to(DEV)
        res = torch.randn(B, H, W, cin, generator=g).to(torch.bfloat16).to(DEV) if use_res else None
        pre = (torch.randn(B, H, W, cin + 8, generator=g) * 2).to(torch.bfloat16).to(DEV)[..., 8:]
        x = torch.zeros(B, cin, H, W, device=DEV, requires_grad=True)
        F.conv2d(x, w.float().permute(0, 3, 1, 2), stride=2, padding=1).backward(gy.float().permute(0, 3, 1, 2))
        ref = x.grad.permute(0, 2, 3, 1) + (res.float() if use_res else 0.0)
        w4 = torch.zeros(4 * cin, 2, 2, cout, dtype=torch.bfloat16, device=DEV)
        KH = {(0, 0): 1, (1, 0): 2, (1, 1): 0}
        for (pa, dh), kh in KH.items():
            for (pb, dw), kw in KH.items():
                w4[(2 * pa + pb) * cin:(2 * pa + pb + 1) * cin, dh, dw] = w[:, kh, kw, :].t()
        zb = torch.zeros(4 * cin, device=DEV)
        nan = lambda c=cin: torch.full((B, H, W, c), float("nan"), dtype=torch.bfloat16, device=DEV)  # noqa: E731
        wsp, rp = (_p(ws) if ws is not None else None), (_p(res) if use_res else None)
        gx = nan()
        rc = L.adayolo_conv_s2grad_fwd(_p(gy), cout, _p(w4), _p(zb), rp, cin if use_res else 0, _p(gx), cin, None, 0, None, 0,
                                       B, Ho, Wo, cout, cin, variant, wsp, nws, st)
        if rc == -2:
            continue
        assert rc == 0, (rc, variant)
        torch.cuda.synchronize()
        assert torch.isfinite(gx.float()).all(), "unwritten (NaN) gradient pixels"
        scale = max(1.0, ref.abs().max().item())
        d = (gx.float() - ref).abs()
        assert d.max().item() <= 2e-2 * scale and d.mean().item() <= 2e-3 * scale, (variant, B, Ho, Wo, cin, cout, d.max().item())
        gp_ref = nan()
        assert L.adayolo_silu_bwd(_p(gx), cin, _p(pre), cin + 8, _p(gp_ref), cin, None, 0, 0, B * H * W, cin, st) == 0
        for store in (True, False):
            gx2, wide = nan(), nan(cin + 16)
            gp = wide[..., 8:8 + cin]
            rc = L.adayolo_conv_s2grad_fwd(_p(gy), cout, _p(w4), _p(zb), rp, cin if use_res else 0, _p(gx2) if store else None,
                                           cin if store else 0, _p(pre), cin + 8, _p(gp), cin + 16, B, Ho, Wo, cout, cin, variant, wsp, nws, st)
            assert rc == 0, (rc, variant)
            torch.cuda.synchronize()
            assert torch.equal(gp.contiguous().view(torch.int16), gp_ref.view(torch.int16)), (variant, store)
            if store:
                assert torch.equal(gx2.view(torch.int16), gx.view(torch.int16))
            assert torch.isnan(wide[..., :8].float()).all() and torch.isnan(wide[..., 8 + cin:].float()).all()
        served += 1
    assert served >= 2, (variant, served)
    assert L.adayolo_conv_s2grad_fwd(_p(gy), cout, _p(w4), _p(zb), None, 0, None, 0, None, 0, None, 0, B, Ho, Wo, cout, cin, 5, None, 0, st) == -1
    assert L.adayolo_conv_s2grad_fwd(_p(gy), cout, _p(w4), _p(zb), None, 0, _p(gx), cin, None, 0, None, 0, B, Ho, Wo, cout, cin, 50, None, 0, st) == -1


def test_stride2_gradient_forms_agree_through_the_whole_backward():
    """The detector backward at the config-4 per-rank shape with the stride-2 layers back-propagated (a) by zero insertion +
    3x3 conv, (b) by adayolo_conv_s2grad_fwd: the same forward (bit for bit), image gradients that differ only by bf16
    rounding of differently ordered sums (relative L2 distance, cosine), five zero-insert launches gone."""
    import os
    from _synth import synth_yolo_state_dict, test_image
    from adaptiveisp_amd.yolo import YoloTrainEngine, yolov3
    cache = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "adaptiveisp_amd", "yolo", "tuning", "mi355x.json")
    det = yolov3()
    det.load_state_dict(synth_yolo_state_dict(det, seed=2))
    det = det.to(DEV).train()
    for p in det.parameters():
        p.requires_grad_(False)
    B, H, W = 8, 512, 512
    x = torch.from_numpy(test_image(B, H, W, seed=7, special=False)).to(DEV)
    res = {}
    for flag in ("0", "1"):
        os.environ["ADAYOLO_TRAIN_S2GRAD"] = flag
        try:
            eng = YoloTrainEngine(det, B, H, W, device=DEV)
            eng.autotune(cache=cache, write=False)
            xr = x.clone().requires_grad_(True)
            raws = eng(xr)
            R = [torch.randn(r.shape, generator=torch.Generator().manual_seed(3 + i)).to(DEV) for i, r in enumerate(raws)]
            sum((r * w).sum() for r, w in zip(raws, R)).backward()
            torch.cuda.synchronize()
            nz = sum(1 for e in eng._backward_plan() if e[0] == "zins")
            res[flag] = ([r.detach().clone() for r in raws], xr.grad.clone(), nz)
            del eng
        finally:
            os.environ.pop("ADAYOLO_TRAIN_S2GRAD", None)
    assert res["0"][2] == 5 and res["1"][2] == 0
    for a, b in zip(res["0"][0], res["1"][0]):
        assert torch.equal(a, b)
    ga, gb = res["0"][1], res["1"][1]
    rel = ((ga - gb).norm() / ga.norm()).item()
    cos = F.cosine_similarity(ga.reshape(1, -1), gb.reshape(1, -1)).item()
    import _margins
    _margins.NOTES.append(f"detector image gradient, zero-insert form vs 2x2 depth-to-space form at 8x512x512: rel {rel:.5f} cos {cos:.7f}")
    # each form is ~1.25 % from fp32 autograd (test_backward_to_image_vs_fp32_autograd) with its own bf16 roundings: measured
    # 0.9 % apart, cosine 0.99996
    assert rel < 2e-2 and cos > 0.9998, (rel, cos)


def test_config4_shape_tuned_engine_vs_fp32_autograd():
    """the training engine at BASELINE config 4's per-rank shape (8 x 512 x 512) with the TUNED table — the
    split-K (S = 2..6), stride-2-gradient and SiLU'-in-epilogue variants the autotuner picks only at this shape are all active
    — against fp32 autograd of the module tree THROUGH THE WHOLE NETWORK: raw head maps of the training forward and the image
    gradient of every image (relative L2 / cosine per image, recorded in the parity-margins file). train.py:262-271,341-342."""
    import os
    import _margins
    from _synth import synth_yolo_state_dict, test_image
    from adaptiveisp_amd.yolo import YoloTrainEngine, yolov3
    cache = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "adaptiveisp_amd", "yolo", "tuning", "mi355x.json")
    det = yolov3()
    det.load_state_dict(synth_yolo_state_dict(det, seed=2))
    det = det.to(DEV).train()
    for m in det.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()
    for p in det.parameters():
        p.requires_grad_(False)
    B, H, W = 8, 512, 512
    eng = YoloTrainEngine(det, B, H, W, device=DEV)
    tuned = eng.autotune(cache=cache, write=False)
    bwd = eng._backward_plan()
    kinds = {e[0] for e in bwd}
    assert any(v >= 100 for v in tuned.values()), "config-4 shape: the table routes no layer to split-K"
    assert "zins" not in kinds, "config-4 shape: a stride-2 layer still goes through zero insertion"
    x = torch.from_numpy(test_image(B, H, W, seed=23, special=False)).to(DEV)
    g = torch.Generator().manual_seed(9)
    xr = x.clone().requires_grad_(True)
    raws_ref = det(xr)                                    # fp32 module tree (512 is a multiple of 32: no letterbox rows)
    R = [torch.randn(r.shape, generator=g).to(DEV) for r in raws_ref]
    sum((r * w).sum() for r, w in zip(raws_ref, R)).backward()
    xh = x.clone().requires_grad_(True)
    raws = eng(xh)
    for a, b in zip(raws, raws_ref):
        _margins.close_scaled("yolo.train_engine_raw_maps_8x512x512", a, b.detach(), 5e-2)
    sum((r * w).sum() for r, w in zip(raws, R)).backward()
    torch.cuda.synchronize()
    assert torch.isfinite(xh.grad).all()
    for i in range(B):                                    # every image of the batch on its own: a wrong tile of one image cannot hide
        _margins.vector_close("yolo.train_engine_image_grad_8x512x512", xh.grad[i], xr.grad[i], max_rel=0.025, min_cos=0.9995)
    _margins.vector_close("yolo.train_engine_image_grad_8x512x512.batch", xh.grad, xr.grad, max_rel=0.025, min_cos=0.9995)


@pytest.mark.parametrize("B,H,W", [(2, 64, 96), (8, 512, 512)])
def test_pair_engine_equals_two_single_passes(B, H, W):
    """YoloTrainPairEngine — ONE forward over [input batch; retouched batch], the backward over the retouched half on the
    buffers that forward filled — against two YoloTrainEngine passes: where both run the same kernels (small shape, no tuning
    table) the per-image losses and the image gradient are bit-identical (a conv output depends on its own image only); with
    other variants / reduction splits for the 2B shapes they agree to bf16 rounding."""
    from _margins import close_scaled, vector_close
    from _synth import synth_yolo_state_dict, test_image
    from adaptiveisp_amd.yolo import YoloTrainEngine, YoloTrainPairEngine, yolov3
    from adaptiveisp_amd.yolo.loss import DetectionLoss, assign_labels_packed, default_hyp
    det = yolov3()
    det.load_state_dict(synth_yolo_state_dict(det, seed=2))
    det = det.to(DEV).train()
    for p in det.parameters():
        p.requires_grad_(False)
    loss_fn = DetectionLoss(det.model[-1].anchors, nc=80, hyp=default_hyp(80, max(H, W)), device=DEV)
    g = torch.Generator().manual_seed(B * 1000 + H)
    labels = _labels(B, g, True)
    imgs = torch.from_numpy(test_image(B, H, W, seed=17, special=False)).to(DEV)
    retouch = torch.from_numpy(test_image(B, H, W, seed=23, special=False)).to(DEV)
    wgt = (torch.rand(B, 1, generator=g) + 0.5).to(DEV)
    single = YoloTrainEngine(det, B, H, W, device=DEV)
    pair = YoloTrainPairEngine(det, B, H, W, device=DEV)
    for tuned in (False, True):
        if tuned:
            import os
            table = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "adaptiveisp_amd", "yolo", "tuning", "mi355x.json")
            single.autotune(cache=table, write=False)
            pair.autotune(cache=table, write=False)
        packed, packed_pair = assign_labels_packed(loss_fn, pair.head_shapes(), labels, DEV, pair=True)
        for (i1, b1), (i2, b2) in zip(packed, packed_pair):
            n = i1.shape[0]
            assert i2.shape[0] == 2 * n and torch.equal(i2[:n], i1) and torch.equal(i2[n:, 0], i1[:, 0] + B)
            assert torch.equal(i2[n:, 1:], i1[:, 1:]) and torch.equal(b2[:n], b1) and torch.equal(b2[n:], b1)
        with torch.no_grad():
            l_in_ref = single.per_sample_loss(loss_fn, imgs, packed)
        xr = retouch.clone().requires_grad_(True)
        l_re_ref = single.per_sample_loss(loss_fn, xr, packed)
        (l_re_ref * wgt).sum().backward()
        xp = retouch.clone().requires_grad_(True)
        l_in, l_re = pair.per_sample_loss_pair(loss_fn, imgs, xp, packed, packed_pair)
        assert not l_in.requires_grad and l_re.requires_grad
        (l_re * wgt).sum().backward()
        torch.cuda.synchronize()
        if not tuned and B == 2:                         # (at 8 x 512 x 512 the default kernel's split of the deep layers' reduction follows the batch)
            assert torch.equal(l_in, l_in_ref) and torch.equal(l_re.detach(), l_re_ref.detach())
            assert torch.equal(xp.grad, xr.grad)
        else:
            close_scaled("yolo.pair_engine.loss", torch.cat([l_in, l_re.detach()]), torch.cat([l_in_ref, l_re_ref.detach()]), 2e-2)
            vector_close("yolo.pair_engine.image_grad", xp.grad, xr.grad, max_rel=5e-2, min_cos=0.998)
        with torch.no_grad():                                              # the no-grad form: the same numbers
            a, b = pair.per_sample_loss_pair(loss_fn, imgs, retouch, packed, packed_pair)
        assert torch.equal(a, l_in) and torch.equal(b, l_re.detach())
    # a backward after a newer forward fails loudly
    xs = retouch.clone().requires_grad_(True)
    _, stale = pair.per_sample_loss_pair(loss_fn, imgs, xs, packed, packed_pair)
    with torch.no_grad():
        pair.per_sample_loss_pair(loss_fn, imgs, retouch, packed, packed_pair)
    with pytest.raises(RuntimeError):
        stale.sum().backward()


def test_detloss_kernels_against_the_reference_fixture_directly(golden):
    """adayolo_detloss_fwd / _bwd through the C-ABI on head maps that ARE the fixture's: detloss.npz `q*` — bf16-exact maps
    scored by the reference's ComputeLossBatch one sample at a time (train.py:175-197), with autograd's gradient of
    sum_b w_b loss_b — laid out as the detector's NHWC bf16 buffers. One hop from the reference to the kernels (the other
    test of this file goes through yolo/loss.py)."""
    import ctypes
    import numpy as np
    from adaptiveisp_amd.yolo import _lib
    from adaptiveisp_amd.yolo.loss import DetectionLoss, assign_labels, pack_assigned
    g = golden("detloss")
    L = _lib.load()
    hyp = dict(box=0.05, cls=0.5, obj=1.0 * (96 / 640) ** 2, anchor_t=4.0, cls_pw=1.0, obj_pw=1.0, fl_gamma=0.0,
               label_smoothing=0.0)
    loss_fn = DetectionLoss(torch.from_numpy(g["anchors"]), nc=80, hyp=hyp, device=DEV)
    qs = [torch.from_numpy(g[f"q{i}"]) for i in range(3)]                      # [B, na, ny, nx, no]
    B, na, no = qs[0].shape[0], qs[0].shape[1], qs[0].shape[4]
    tg = torch.from_numpy(g["targets"])
    labels = [tg[tg[:, 0] == b] for b in range(B)]
    packed = pack_assigned(assign_labels(loss_fn, [q.to(DEV) for q in qs], labels))
    cs = 256
    raws, grads, keep = [], [], []
    a = _lib.LossArgs()
    for i, (q, (idx, box)) in enumerate(zip(qs, packed)):
        ny, nx = q.shape[2], q.shape[3]
        raw = torch.zeros((B, ny, nx, cs), dtype=torch.bfloat16, device=DEV)
        raw[..., : na * no] = q.permute(0, 2, 3, 1, 4).reshape(B, ny, nx, na * no).to(torch.bfloat16).to(DEV)
        assert torch.equal(raw[..., : na * no].float().cpu().view(B, ny, nx, na, no).permute(0, 3, 1, 2, 4), q)   # bf16-exact
        grad = torch.full((B, ny, nx, cs), 7.0, dtype=torch.bfloat16, device=DEV)
        ws = [torch.empty((B, 3), dtype=torch.float32, device=DEV), torch.empty((B, na, ny, nx), dtype=torch.float32, device=DEV),
              torch.empty((B,), dtype=torch.float32, device=DEV)]
        Ly = a.layer[i]
        Ly.raw, Ly.cs, Ly.ny, Ly.nx, Ly.balance = raw.data_ptr(), cs, ny, nx, float(loss_fn.balance[i])
        n = int(idx.shape[0])
        Ly.idx, Ly.box, Ly.n = (idx.data_ptr() if n else None), (box.data_ptr() if n else None), n
        Ly.part, Ly.tobj, Ly.cnt = ws[0].data_ptr(), ws[1].data_ptr(), ws[2].data_ptr()
        Ly.grad, Ly.grad_cs = grad.data_ptr(), cs
        raws.append(raw); grads.append(grad); keep += ws + [idx, box]
    a.nl, a.B, a.na, a.nc, a.no = 3, B, na, 80, no
    a.hyp_box, a.hyp_obj, a.hyp_cls = hyp["box"], hyp["obj"], hyp["cls"]
    a.cp, a.cn, a.cls_pw, a.obj_pw = float(loss_fn.cp), float(loss_fn.cn), 1.0, 1.0
    loss = torch.empty((B,), dtype=torch.float32, device=DEV)
    ticket = torch.zeros((B,), dtype=torch.int32, device=DEV)
    w = torch.from_numpy(g["qweights"]).to(DEV)
    a.loss, a.ticket, a.grad_loss = loss.data_ptr(), ticket.data_ptr(), w.data_ptr()
    st = _lib.stream_ptr()
    _lib.check(L.adayolo_detloss_fwd(ctypes.byref(a), st), "adayolo_detloss_fwd")
    torch.cuda.synchronize()
    want = np.array([g[f"qsample{b}"].sum() for b in range(B)], np.float32)
    np.testing.assert_allclose(loss.cpu().numpy(), want, rtol=2e-5, atol=1e-6)
    scratch = torch.empty((B,), dtype=torch.float32, device=DEV)
    a.loss = scratch.data_ptr()
    _lib.check(L.adayolo_detloss_bwd(ctypes.byref(a), st), "adayolo_detloss_bwd")
    torch.cuda.synchronize()
    assert (ticket == 0).all()
    for i, (gr, q) in enumerate(zip(grads, qs)):
        ny, nx = q.shape[2], q.shape[3]
        got = gr[..., : na * no].float().cpu().view(B, ny, nx, na, no).permute(0, 3, 1, 2, 4)
        assert (gr[..., na * no:] == 0).all()
        ref = torch.from_numpy(g[f"qgrad{i}"])
        scale = ref.abs().max().item()
        # the kernels store the gradient as bf16: one ulp (2^-8 relative) of each element + fp32 ordering noise
        assert ((got - ref).abs() <= ref.abs() * 2 ** -7 + 2e-6 * scale).all(), (i, (got - ref).abs().max().item(), scale)
