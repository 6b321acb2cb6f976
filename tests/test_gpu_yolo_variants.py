"""Every conv kernel the autotuner may pick (yolo/tuning/mi355x.json) against fp32 `F.conv2d` on the same bf16-rounded
operands, at the layer shapes it is benchmarked on, plus the whole TUNED engine at the BASELINE sizes against the fp32
module tree (yolo/model.py, itself bit-identical to the reference's DetectionModel on tests/golden/yolo.npz).

What each case demands (reference semantics: Conv.forward_fuse = act(conv(x) + folded-BN bias),
Bottleneck.forward = x + cv2(cv1(x)); yolov3/models/common.py:45-59,110-120):
  * the output buffer is pre-filled with NaN, so a tile that is never written fails;
  * 4 launches give bit-identical results (race screen);
  * max |err| <= 2e-2 * max(1, max|ref|) and mean |err| <= 2e-3 * max(1, max|ref|) (two bf16 roundings)."""
import ctypes
import json
import os

import pytest
import torch
import torch.nn.functional as F

from _margins import close, close_scaled

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TUNE = os.path.join(ROOT, "adaptiveisp_amd", "yolo", "tuning", "mi355x.json")


def _table():
    return {tuple(int(x) for x in k.split(",")): int(v) for k, v in json.load(open(TUNE)).items()}


SPLITK_BASE = 100


def _splitk_bytes(v, B, H, W, cin, cout, k, s):
    from adaptiveisp_amd.yolo import _lib
    return int(_lib.load().adayolo_conv_splitk_workspace_bytes(B, H, W, cin, cout, k, s, v))


def _serves(v, cin, cout, k, s, bhw=None):
    """Shapes a variant is specialised for (adayolo.h: other shapes fall through to the default kernel; a split-K variant
    serves what adayolo_conv_splitk_workspace_bytes says it serves and nothing else)."""
    if v >= SPLITK_BASE:
        return bhw is not None and _splitk_bytes(v, *bhw, cin, cout, k, s) > 0
    if 40 <= v < 50:
        return k == 3 and cin in (32, 64)
    if 50 <= v < 60:
        return cin % 64 == 0 and cout % 256 == 0
    if 60 <= v < 80:
        return cin % 64 == 0 and cout % 128 == 0
    if 80 <= v < 90:                                   # 80: 256-pixel tile, 85: 128-pixel tile
        return cin % 32 == 0 and cout % 128 == 0 and k * k * cin >= 96
    if v >= 90:
        return k == 3 and s == 1 and cin in (32, 64) and cout % 64 == 0
    return True


def _run_variant(x, w, b, res, k, s, act, v, reps=4, pre=None):
    from adaptiveisp_amd.yolo import _lib
    L = _lib.load()
    B, H, W, cin = x.shape
    cout = w.shape[0]
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    first = None
    ws = None
    if v >= SPLITK_BASE:                                # ONE zeroed workspace for all launches: each must leave its tickets zero
        ws = torch.zeros(_splitk_bytes(v, B, H, W, cin, cout, k, s), dtype=torch.uint8, device=DEV)
        assert ws.numel() > 0, f"split variant {v} does not serve {(B, H, W, cin, cout, k, s)}"
    for _ in range(reps):
        out = torch.full((B, Ho, Wo, cout), float("nan"), dtype=torch.bfloat16, device=DEV)
        common = (ctypes.c_void_p(x.data_ptr()), cin, ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(b.data_ptr()),
                  ctypes.c_void_p(res.data_ptr()) if res is not None else None, cout if res is not None else 0,
                  ctypes.c_void_p(out.data_ptr()), cout)
        if ws is not None:
            rc = L.adayolo_conv_splitk_fwd(*common, ctypes.c_void_p(pre.data_ptr()) if pre is not None else None,
                                           cout if pre is not None else 0, B, H, W, cin, cout, k, s, act, v,
                                           ctypes.c_void_p(ws.data_ptr()), ws.numel(), _lib.stream_ptr())
        elif pre is not None:
            rc = L.adayolo_conv_keep_fwd(*common, ctypes.c_void_p(pre.data_ptr()), cout, B, H, W, cin, cout, k, s, act, v,
                                         _lib.stream_ptr())
        else:
            rc = L.adayolo_conv_fwd_variant(*common, B, H, W, cin, cout, k, s, act, v, _lib.stream_ptr())
        _lib.check(rc, "conv")
        torch.cuda.synchronize()
        if first is None:
            first = out
        else:
            assert torch.equal(first.view(torch.int16), out.view(torch.int16)), "run-to-run difference"
    return first


def _reference(x, w, b, res, k, s, act):
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b, stride=s, padding=k // 2)
    if act:
        ref = F.silu(ref)
    ref = ref.permute(0, 2, 3, 1)
    if res is not None:
        ref = ref.to(torch.bfloat16).float() + res.float()
    return ref


def _check(shape, v, use_res, acts=(1,)):
    B, H, W, cin, cout, k, s = shape
    g = torch.Generator(device="cpu").manual_seed((H * 131 + cin * 7 + cout + v) % 2 ** 31)
    x = torch.randn(B, H, W, cin, generator=g).to(torch.bfloat16).to(DEV)
    w = (torch.randn(cout, k, k, cin, generator=g) / (k * k * cin) ** 0.5).to(torch.bfloat16).to(DEV)
    b = torch.randn(cout, generator=g).to(DEV)
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    res = torch.randn(B, Ho, Wo, cout, generator=g).to(torch.bfloat16).to(DEV) if use_res else None
    for act in acts:
        out = _run_variant(x, w, b, res, k, s, act, v)
        ref = _reference(x, w, b, res, k, s, act)
        assert torch.isfinite(out.float()).all(), "unwritten (NaN) outputs"
        scale = max(1.0, ref.abs().max().item())
        close_scaled(f"yolo.variant_{v}_vs_fp32", out.float(), ref, 2e-2, err_msg=f"{shape} v{v} act{act}")
        d = (out.float() - ref).abs()
        assert d.mean().item() <= 2e-3 * scale, f"{shape} v{v} act{act}: mean err {d.mean().item()}"
        del out, ref, d


# the layer shapes of the BASELINE network (8 x 736 x 1280) with the kernel the tuning table picks for each
def _c2_cases():
    t = _table()
    return sorted((k[:7], v) for k, v in t.items() if k[1] in (736, 368, 184, 92, 46, 23) and k[7] == 1)


@pytest.mark.parametrize("shape,v", _c2_cases(), ids=lambda p: "x".join(map(str, p)) if isinstance(p, tuple) else f"v{p}")
def test_tuned_kernel_at_its_baseline_layer(shape, v):
    """Each (layer shape, chosen variant) pair of the benchmarked network, full size, residual as in a Bottleneck."""
    k, s = shape[5], shape[6]
    _check(shape, v, use_res=(k == 3 and s == 1))


def _used_variants():
    return sorted(set(_table().values()))


RAGGED = [  # B, H, W, Cin, Cout, k, s, residual — partial tiles in M and N, tiny maps, stride 2 on odd sizes
    (2, 19, 33, 128, 256, 3, 1, True), (1, 5, 6, 512, 1024, 3, 1, True), (3, 9, 11, 64, 256, 1, 1, False),
    (1, 7, 9, 64, 256, 3, 2, False), (2, 37, 53, 64, 128, 3, 2, False), (1, 9, 7, 192, 384, 3, 1, True),
    (2, 16, 24, 32, 64, 3, 2, False), (1, 23, 40, 64, 32, 1, 1, False), (2, 12, 20, 32, 64, 3, 1, True),
    (3, 9, 11, 768, 256, 1, 1, False), (2, 7, 5, 8, 32, 3, 1, False), (1, 30, 30, 64, 128, 3, 2, False),
    (1, 1, 1, 64, 256, 3, 1, False), (1, 2, 3, 128, 128, 1, 1, True), (5, 13, 17, 256, 768, 1, 1, False),
    (2, 31, 29, 32, 64, 3, 1, False), (1, 3, 200, 64, 64, 3, 1, True),
]


SPLIT_RAGGED = [  # long reductions on few pixels (what the split-K variants are for), ragged in M
    (1, 5, 6, 1024, 512, 3, 1, True), (2, 9, 7, 512, 256, 3, 1, False), (3, 5, 5, 1024, 128, 1, 1, False),
    (2, 13, 17, 256, 128, 3, 1, True), (1, 16, 16, 512, 1024, 3, 2, False), (8, 16, 16, 1024, 512, 3, 1, True),
]


@pytest.mark.parametrize("v", _used_variants())
def test_every_tuned_variant_on_ragged_shapes(v):
    """Every variant the table can select, on small / odd / partial-tile shapes it serves, act on and off."""
    n = 0
    for (B, H, W, cin, cout, k, s, use_res) in RAGGED + (SPLIT_RAGGED if v >= SPLITK_BASE else []):
        if _serves(v, cin, cout, k, s, (B, H, W)):
            _check((B, H, W, cin, cout, k, s), v, use_res, acts=(0, 1))
            n += 1
    assert n >= 3, f"variant {v}: too few shapes exercised"


@pytest.mark.parametrize("v", _used_variants())
def test_variant_channel_slices(v):
    """Concat is free because convs read / write channel slices of wider tensors: strides larger than C, every variant."""
    from adaptiveisp_amd.yolo import _lib
    L = _lib.load()
    if v >= SPLITK_BASE:
        pytest.skip("split-K variants: slices are covered by test_splitk_conv (their own entry point)")
    cin, cout, k, s = (64, 256, 3, 1)
    g = torch.Generator(device="cpu").manual_seed(50 + v)
    wide_in = torch.randn(2, 21, 35, cin + 64, generator=g).to(torch.bfloat16).to(DEV)
    wide_out = torch.full((2, 21, 35, cout + 128), 7.0, dtype=torch.bfloat16, device=DEV)
    wide_res = torch.randn(2, 21, 35, cout + 8, generator=g).to(torch.bfloat16).to(DEV)
    w = (torch.randn(cout, k, k, cin, generator=g) / 24).to(torch.bfloat16).to(DEV)
    b = torch.randn(cout, generator=g).to(DEV)
    xin, xout, xres = wide_in[..., 32:32 + cin], wide_out[..., 64:64 + cout], wide_res[..., 8:]
    rc = L.adayolo_conv_fwd_variant(ctypes.c_void_p(xin.data_ptr()), wide_in.shape[3], ctypes.c_void_p(w.data_ptr()),
                                    ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(xres.data_ptr()), wide_res.shape[3],
                                    ctypes.c_void_p(xout.data_ptr()), wide_out.shape[3], 2, 21, 35, cin, cout, k, s, 1, v,
                                    _lib.stream_ptr())
    _lib.check(rc, "conv")
    torch.cuda.synchronize()
    ref = _reference(xin.contiguous(), w, b, xres.contiguous(), k, s, 1)
    assert (xout.float() - ref).abs().max().item() <= 2e-2 * max(1.0, ref.abs().max().item())
    assert (wide_out[..., :64] == 7).all() and (wide_out[..., 64 + cout:] == 7).all()       # neighbours untouched


def _tuned_engine_vs_module_tree(B, H, W, images):
    from _synth import synth_yolo_state_dict, test_image
    from adaptiveisp_amd.yolo import YoloEngine, yolov3
    m = yolov3().eval()
    m.load_state_dict(synth_yolo_state_dict(m))
    x = torch.from_numpy(test_image(B, H, W, seed=91, special=False)).to(DEV)
    eng = YoloEngine(m, B, H, W, device=DEV)
    tuned = eng.autotune(cache=TUNE)
    assert set(tuned.values()) - {2}, "the tuning table must route layers to the specialised kernels"
    if (B, H, W) == (8, 720, 1280):
        assert set(tuned.values()) >= {50, 60}, "BASELINE shape: table lost the ping-pong kernels"
        assert eng.fused_pairs >= 8, f"BASELINE shape: only {eng.fused_pairs} 3x3 + 1x1 pairs run fused"
    pred = eng(x).clone()
    raws = [r.clone() for r in eng.raw_maps()]
    torch.cuda.synchronize()
    ref_model = m.to(DEV)
    for i in images:
        boxed = torch.full((1, 3, eng.Hp, W), 114 / 255, device=DEV)
        boxed[:, :, eng.pad_top:eng.pad_top + H] = x[i:i + 1]
        with torch.no_grad():
            ref_pred, ref_raw = ref_model(boxed)
        for r, rr in zip(raws, ref_raw):
            close_scaled(f"yolo.tuned_engine_raw_maps_{H}x{W}", r[i:i + 1], rr, 3e-2, floor=0.0, err_msg=f"image {i}")
        close(f"yolo.tuned_engine_pred_{H}x{W}", pred[i:i + 1], ref_pred, rtol=2e-2, atol=2e-2, err_msg=f"image {i}")
        del ref_pred, ref_raw
    # the same engine with every layer launched separately: the fused pairs must not move the result beyond rounding
    if eng.fused_pairs:
        os.environ["ADAYOLO_FUSE_1X1"] = "0"
        try:
            eng2 = YoloEngine(m.to("cpu"), B, H, W, device=DEV)
            eng2.autotune(cache=TUNE)
            assert eng2.fused_pairs == 0
            pred2 = eng2(x)
            torch.cuda.synchronize()
            rel = ((pred - pred2).abs() / (pred2.abs() + 1.0)).max().item()
            assert rel < 2e-2, f"fused vs separate launches: rel err {rel}"
        finally:
            del os.environ["ADAYOLO_FUSE_1X1"]
    m.to("cpu")
    return tuned


def test_tuned_engine_baseline_size_vs_fp32_module_tree():
    """YoloEngine(8 x 720 x 1280) WITH the tuning table the bench uses (variants 50/60/27/5/22/40) vs the fp32 module
    tree on images 0 and 7: three raw maps <= 3e-2 of scale, decoded prediction <= 2e-2 relative."""
    _tuned_engine_vs_module_tree(8, 720, 1280, images=(0, 7))


def test_tuned_engine_4k_vs_fp32_module_tree():
    """Config 5 detector (4 x 2160 x 3840 -> letterboxed 2176): one image against the fp32 module tree."""
    _tuned_engine_vs_module_tree(4, 2160, 3840, images=(3,))


FUSED_CASES = [  # B, H, W, Cin, k, s, residual — the first layer always has 256 output channels, the second is 1x1 256 -> 128
    (8, 92, 160, 128, 3, 1, True), (8, 184, 320, 128, 3, 2, False), (2, 19, 33, 128, 3, 1, True), (1, 7, 9, 64, 3, 2, False),
    (3, 9, 11, 64, 1, 1, False), (1, 5, 6, 512, 3, 1, True),
]


@pytest.mark.parametrize("case", FUSED_CASES, ids=lambda c: "x".join(map(str, c)))
def test_fused_3x3_plus_1x1(case):
    """adayolo_conv_fused1x1_fwd = Bottleneck.cv2 of one block + Bottleneck.cv1 of the next in one launch (reference:
    yolov3/models/common.py:110-120). `out` must be BIT-IDENTICAL to the unfused kernel's, `out2` must equal the 1x1 conv of
    that bf16 output (fp32 reference, <= 2e-2 of scale; and within bf16 rounding of the separate 1x1 launch), four
    launches bit-identical, NaN-prefilled outputs fully written."""
    from adaptiveisp_amd.yolo import _lib
    L = _lib.load()
    B, H, W, cin, k, s, use_res = case
    g = torch.Generator(device="cpu").manual_seed(H * 31 + cin)
    x = torch.randn(B, H, W, cin, generator=g).to(torch.bfloat16).to(DEV)
    w = (torch.randn(256, k, k, cin, generator=g) / (k * k * cin) ** 0.5).to(torch.bfloat16).to(DEV)
    b = torch.randn(256, generator=g).to(DEV)
    w2 = (torch.randn(128, 1, 1, 256, generator=g) / 16.0).to(torch.bfloat16).to(DEV)
    b2 = torch.randn(128, generator=g).to(DEV)
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    res = torch.randn(B, Ho, Wo, 256, generator=g).to(torch.bfloat16).to(DEV) if use_res else None
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
    w2p = w2.reshape(4, 32, 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous()      # fragment-major (include/adayolo.h)
    first = None
    for _ in range(4):
        out = torch.full((B, Ho, Wo, 256), float("nan"), dtype=torch.bfloat16, device=DEV)
        out2 = torch.full((B, Ho, Wo, 128), float("nan"), dtype=torch.bfloat16, device=DEV)
        rc = L.adayolo_conv_fused1x1_fwd(P(x), cin, P(w), P(b), P(res), 256 if use_res else 0, P(out), 256, B, H, W, cin, 256,
                                         k, s, 1, P(w2p), P(b2), P(out2), 128, 128, _lib.stream_ptr())
        _lib.check(rc, "fused conv")
        torch.cuda.synchronize()
        if first is None:
            first = (out, out2)
        else:
            assert torch.equal(first[0].view(torch.int16), out.view(torch.int16)), "run-to-run difference (out)"
            assert torch.equal(first[1].view(torch.int16), out2.view(torch.int16)), "run-to-run difference (out2)"
    out, out2 = first
    assert torch.isfinite(out.float()).all() and torch.isfinite(out2.float()).all(), "unwritten (NaN) outputs"
    sep = _run_variant(x, w, b, res, k, s, 1, 50, reps=1)
    assert torch.equal(sep.view(torch.int16), out.view(torch.int16)), "first layer differs from the unfused kernel"
    ref2 = _reference(out, w2, b2, None, 1, 1, 1)                       # the second layer reads the bf16 output
    scale = max(1.0, ref2.abs().max().item())
    d = (out2.float() - ref2).abs()
    assert d.max().item() <= 2e-2 * scale and d.mean().item() <= 2e-3 * scale, (d.max().item(), d.mean().item(), scale)
    sep2 = _run_variant(out, w2, b2, None, 1, 1, 1, 80, reps=1)
    assert (out2.float() - sep2.float()).abs().max().item() <= 2.0 ** -6 * scale     # two bf16 roundings of the same sum


def test_fused_entry_rejects_other_shapes():
    from adaptiveisp_amd.yolo import _lib
    L = _lib.load()
    t = torch.zeros(1, 8, 8, 256, dtype=torch.bfloat16, device=DEV)
    f = torch.zeros(256, device=DEV)
    P = lambda v: ctypes.c_void_p(v.data_ptr())  # noqa: E731
    # first layer with 128 output channels: its tile does not hold what the second layer needs
    rc = L.adayolo_conv_fused1x1_fwd(P(t), 256, P(t), P(f), None, 0, P(t), 128, 1, 8, 8, 256, 128, 1, 1, 1, P(t), P(f), P(t), 128, 128,
                                     _lib.stream_ptr())
    assert rc != 0
    rc = L.adayolo_conv_fused1x1_fwd(P(t), 256, P(t), P(f), None, 0, P(t), 256, 1, 8, 8, 256, 256, 1, 1, 1, None, P(f), P(t), 128, 128,
                                     _lib.stream_ptr())
    assert rc != 0


SPLITK_CASES = [  # B, H, W, Cin, Cout, k, s, residual: the deep layers of the 8 x 512 x 512 training step, forward and data gradient
    (8, 16, 16, 512, 1024, 3, 1, True), (8, 16, 16, 1024, 512, 3, 1, True), (8, 32, 32, 512, 256, 3, 1, False),
    (8, 32, 32, 256, 512, 3, 1, True), (8, 32, 32, 512, 1024, 3, 2, False), (8, 16, 16, 1024, 512, 1, 1, False),
    (2, 13, 17, 256, 128, 3, 1, True), (1, 5, 6, 1024, 512, 3, 1, False),
]


@pytest.mark.parametrize("case", SPLITK_CASES, ids=lambda c: "x".join(map(str, c)))
def test_splitk_conv(case):
    """adayolo_conv_splitk_fwd, every split S the library serves for the shape: fp32 conv reference within the variants'
    tolerance, four launches on ONE workspace bit-identical (deterministic range-order sum, tickets left zero), NaN-prefilled
    outputs fully written; with `pre`: pre is the bf16 pre-activation and out == silu(pre) (+ residual) computed from
    that rounded value (the contract of adayolo_conv_keep_fwd / adayolo_silu_fwd); channel-slice strides honoured."""
    from adaptiveisp_amd.yolo import _lib
    L = _lib.load()
    B, H, W, cin, cout, k, s, use_res = case
    served = [v for v in range(SPLITK_BASE + 2, SPLITK_BASE + 17) if _splitk_bytes(v, B, H, W, cin, cout, k, s) > 0]
    assert served, f"no split serves {case}"
    assert _splitk_bytes(SPLITK_BASE + 1, B, H, W, cin, cout, k, s) == 0 and _splitk_bytes(SPLITK_BASE + 17, B, H, W, cin, cout, k, s) == 0
    g = torch.Generator(device="cpu").manual_seed(H * 131 + cin * 7 + cout)
    x = torch.randn(B, H, W, cin, generator=g).to(torch.bfloat16).to(DEV)
    w = (torch.randn(cout, k, k, cin, generator=g) / (k * k * cin) ** 0.5).to(torch.bfloat16).to(DEV)
    b = torch.randn(cout, generator=g).to(DEV)
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    res = torch.randn(B, Ho, Wo, cout, generator=g).to(torch.bfloat16).to(DEV) if use_res else None
    for v in served:
        for act in (0, 1):
            out = _run_variant(x, w, b, res, k, s, act, v)
            ref = _reference(x, w, b, res, k, s, act)
            assert torch.isfinite(out.float()).all(), f"v{v}: unwritten (NaN) outputs"
            scale = max(1.0, ref.abs().max().item())
            d = (out.float() - ref).abs()
            assert d.max().item() <= 2e-2 * scale and d.mean().item() <= 2e-3 * scale, (v, act, d.max().item(), d.mean().item())
        # training forward: second output
        pre = torch.full((B, Ho, Wo, cout), float("nan"), dtype=torch.bfloat16, device=DEV)
        out = _run_variant(x, w, b, res, k, s, 1, v, reps=2, pre=pre)
        lin = _run_variant(x, w, b, None, k, s, 0, v, reps=1)
        assert torch.equal(pre.view(torch.int16), lin.view(torch.int16)), f"v{v}: pre is not the conv + bias output"
        two = torch.empty_like(out)
        rc = L.adayolo_silu_fwd(ctypes.c_void_p(pre.data_ptr()), cout, ctypes.c_void_p(res.data_ptr()) if use_res else None,
                                cout if use_res else 0, ctypes.c_void_p(two.data_ptr()), cout, B * Ho * Wo, cout, _lib.stream_ptr())
        _lib.check(rc, "silu_fwd")
        torch.cuda.synchronize()
        assert torch.equal(two.view(torch.int16), out.view(torch.int16)), f"v{v}: out differs from silu_fwd(pre, residual)"
    # channel slices of wider tensors, one split
    v = served[len(served) // 2]
    wide_in = torch.randn(B, H, W, cin + 64, generator=g).to(torch.bfloat16).to(DEV)
    wide_out = torch.full((B, Ho, Wo, cout + 128), 7.0, dtype=torch.bfloat16, device=DEV)
    xin, xout = wide_in[..., 32:32 + cin], wide_out[..., 64:64 + cout]
    ws = torch.zeros(_splitk_bytes(v, B, H, W, cin, cout, k, s), dtype=torch.uint8, device=DEV)
    rc = L.adayolo_conv_splitk_fwd(ctypes.c_void_p(xin.data_ptr()), cin + 64, ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(b.data_ptr()),
                                   None, 0, ctypes.c_void_p(xout.data_ptr()), cout + 128, None, 0, B, H, W, cin, cout, k, s, 1, v,
                                   ctypes.c_void_p(ws.data_ptr()), ws.numel(), _lib.stream_ptr())
    _lib.check(rc, "splitk conv")
    torch.cuda.synchronize()
    ref = _reference(xin.contiguous(), w, b, None, k, s, 1)
    assert (xout.float() - ref).abs().max().item() <= 2e-2 * max(1.0, ref.abs().max().item())
    assert (wide_out[..., :64] == 7).all() and (wide_out[..., 64 + cout:] == 7).all()
    assert int(ws[: 1024].view(torch.int32).abs().sum()) == 0, "tickets not left zero"
    # argument checks of the entry point
    small = torch.zeros(1024, dtype=torch.uint8, device=DEV)
    args = (ctypes.c_void_p(x.data_ptr()), cin, ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(b.data_ptr()), None, 0,
            ctypes.c_void_p(wide_out.data_ptr()), cout + 128, None, 0, B, H, W, cin, cout, k, s, 1)
    assert L.adayolo_conv_splitk_fwd(*args, v, ctypes.c_void_p(small.data_ptr()), small.numel(), _lib.stream_ptr()) == -1   # EINVAL: workspace too small
    assert L.adayolo_conv_splitk_fwd(*args, v, None, 0, _lib.stream_ptr()) == -1
    assert L.adayolo_conv_splitk_fwd(*args, 60, ctypes.c_void_p(ws.data_ptr()), ws.numel(), _lib.stream_ptr()) == -1        # not a split variant


def test_splitk_conv_repeated_launches_are_bit_stable():
    """Race screen for the split-K reduce (partial tiles cross XCDs at device scope, last-arriver ticket): 300 launches each
    of two training-step shapes on ONE workspace, other work interleaved on a second stream, every result bit-identical to
    the first and the tickets zero at the end."""
    from adaptiveisp_amd.yolo import _lib
    L = _lib.load()
    other = torch.cuda.Stream()
    junk = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    for (B, H, W, cin, cout, k, s, v) in [(8, 16, 16, 1024, 512, 3, 1, 106), (8, 32, 32, 512, 256, 3, 1, 104)]:
        nws = _splitk_bytes(v, B, H, W, cin, cout, k, s)
        assert nws > 0
        ws = torch.zeros(nws, dtype=torch.uint8, device=DEV)
        g = torch.Generator(device="cpu").manual_seed(cin + v)
        x = torch.randn(B, H, W, cin, generator=g).to(torch.bfloat16).to(DEV)
        w = (torch.randn(cout, k, k, cin, generator=g) / (k * k * cin) ** 0.5).to(torch.bfloat16).to(DEV)
        b = torch.randn(cout, generator=g).to(DEV)
        outs = [torch.empty(B, H, W, cout, dtype=torch.bfloat16, device=DEV) for _ in range(2)]
        first = None
        for it in range(300):
            out = outs[it & 1]
            if it % 7 == 0:
                with torch.cuda.stream(other):
                    junk.add_(1)                       # traffic through the L2s / Infinity Cache beside the launch
            rc = L.adayolo_conv_splitk_fwd(ctypes.c_void_p(x.data_ptr()), cin, ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(b.data_ptr()),
                                           None, 0, ctypes.c_void_p(out.data_ptr()), cout, None, 0, B, H, W, cin, cout, k, s, 1, v,
                                           ctypes.c_void_p(ws.data_ptr()), ws.numel(), _lib.stream_ptr())
            assert rc == 0
            if first is None:
                torch.cuda.synchronize()
                first = out.clone()
            elif it % 10 == 9:
                torch.cuda.synchronize()
                assert torch.equal(first.view(torch.int16), out.view(torch.int16)), f"launch {it} differs"
        torch.cuda.synchronize()
        assert int(ws[:1024].view(torch.int32).abs().sum()) == 0          # the ticket block (<= 256 tiles)


@pytest.mark.parametrize("shape", [(1, 16, 16), (1, 5, 7), (2, 23, 37), (1, 40, 33), (8, 92, 160)])
def test_bottleneck256_kernel(shape):
    """adayolo_bottleneck256_fwd (csrc/yolo_bneck.hip: a whole Bottleneck of the C = 256 stage in one launch, hidden tensor in
    LDS) against fp32 `x + cv2(cv1(x))` on the same bf16 operands with the hidden tensor rounded to bf16 as the stand-alone
    layers store it (yolov3/models/common.py:110-120), and against those two layers launched separately: ragged tiles, image
    borders (the 3x3 pads the HIDDEN tensor with zeros, not x), NaN-prefilled output, run-to-run bit identity."""
    from adaptiveisp_amd.yolo import _lib
    L = _lib.load()
    B, H, W = shape
    g = torch.Generator(device="cpu").manual_seed(H * 7 + W)
    x = torch.randn(B, H, W, 256, generator=g).to(torch.bfloat16).to(DEV)
    w1 = (torch.randn(128, 1, 1, 256, generator=g) / 16).to(torch.bfloat16).to(DEV)
    b1 = (torch.randn(128, generator=g) * 0.5).to(DEV)
    w2 = (torch.randn(256, 3, 3, 128, generator=g) / (9 * 128) ** 0.5).to(torch.bfloat16).to(DEV)
    b2 = (torch.randn(256, generator=g) * 0.5).to(DEV)
    P = lambda t: ctypes.c_void_p(t.data_ptr())                                # noqa: E731
    st = _lib.stream_ptr()
    outs = []
    for _ in range(3):
        out = torch.full((B, H, W, 256), float("nan"), dtype=torch.bfloat16, device=DEV)
        assert L.adayolo_bottleneck256_fwd(P(x), 256, P(w1), P(b1), P(w2), P(b2), P(out), 256, B, H, W, st) == 0
        outs.append(out)
    torch.cuda.synchronize()
    out = outs[0]
    assert torch.isfinite(out.float()).all(), "unwritten (NaN) outputs"
    assert all(torch.equal(out.view(torch.int16), o.view(torch.int16)) for o in outs[1:]), "run-to-run difference"
    xf = x.float().permute(0, 3, 1, 2)
    h = F.silu(F.conv2d(xf, w1.float().permute(0, 3, 1, 2), b1)).to(torch.bfloat16).float()
    ref = (F.silu(F.conv2d(h, w2.float().permute(0, 3, 1, 2), b2, padding=1)).to(torch.bfloat16).float() + xf).permute(0, 2, 3, 1)
    close_scaled("yolo.bottleneck256_vs_fp32", out.float(), ref, 2e-2)
    # the two stand-alone layers (1x1 on the 128-px two-workgroup kernel, 3x3 + residual on the 256 x 256 kernel)
    hid = torch.empty(B, H, W, 128, dtype=torch.bfloat16, device=DEV)
    two = torch.empty_like(out)
    assert L.adayolo_conv_fwd_variant(P(x), 256, P(w1), P(b1), None, 0, P(hid), 128, B, H, W, 256, 128, 1, 1, 1, 85, st) == 0
    assert L.adayolo_conv_fwd_variant(P(hid), 128, P(w2), P(b2), P(x), 256, P(two), 256, B, H, W, 128, 256, 3, 1, 1, 50, st) == 0
    torch.cuda.synchronize()
    close_scaled("yolo.bottleneck256_vs_two_layers", out.float(), two.float(), 2.0 ** -6)     # bf16 roundings of differently ordered sums
    # argument checks of the C-ABI
    assert L.adayolo_bottleneck256_fwd(P(x), 256, P(w1), P(b1), P(w2), P(b2), P(x), 256, B, H, W, st) == -1      # in place
    assert L.adayolo_bottleneck256_fwd(P(x), 250, P(w1), P(b1), P(w2), P(b2), P(out), 256, B, H, W, st) == -2


def test_engine_with_bottleneck_launches_matches_the_default_plan():
    """ADAYOLO_BNECK=1: the eight blocks of the C = 256 stage as one launch each. Same predictions as the default plan
    (pairs [3x3 | next 1x1]) up to bf16 rounding, at the BASELINE shape."""
    from _synth import synth_yolo_state_dict, test_image
    from adaptiveisp_amd.yolo import YoloEngine, yolov3
    m = yolov3().eval()
    m.load_state_dict(synth_yolo_state_dict(m))
    B, H, W = 8, 720, 1280
    x = torch.from_numpy(test_image(B, H, W, seed=5, special=False)).to(DEV)
    base = YoloEngine(m, B, H, W, device=DEV)
    base.autotune(cache=TUNE, write=False)
    ref = base(x).clone()
    os.environ["ADAYOLO_BNECK"] = "1"
    try:
        eng = YoloEngine(m, B, H, W, device=DEV)
        eng.autotune(cache=TUNE, write=False)
    finally:
        os.environ.pop("ADAYOLO_BNECK", None)
    assert eng.fused_blocks == 8 and sum(1 for e in eng.plan if e[0] == "bneck") == 8
    out = eng(x)
    torch.cuda.synchronize()
    close("yolo.engine_bneck_vs_default_plan", out, ref, rtol=2e-2, atol=2e-2)
