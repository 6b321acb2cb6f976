"""adayolo_bottleneck_ws_fwd (csrc/yolo_bneck_ws.hip: a whole Bottleneck of the C = 64 / C = 128 stages in one launch, hidden
tensor in LDS) against fp32 `x + cv2(cv1(x))` on the same bf16 operands with the hidden tensor rounded to bf16 as the stand-alone
layers store it (yolov3/models/common.py:110-120; yolov3.yaml:13-27), and against those two layers launched separately: ragged
tiles, image borders (the 3x3 pads the HIDDEN tensor with zeros, not x), channel-slice strides, NaN-prefilled output, run-to-run
bit identity, the engine with and without the fused blocks at the BASELINE shape."""
import ctypes
import os

import pytest
import torch
import torch.nn.functional as F

from _margins import close_scaled

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TUNE = os.path.join(ROOT, "adaptiveisp_amd", "yolo", "tuning", "mi355x.json")


def _operands(B, H, W, C, seed, x_cs=None):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(B, H, W, x_cs or C, generator=g).to(torch.bfloat16).to(DEV)
    w1 = (torch.randn(C // 2, 1, 1, C, generator=g) / C ** 0.5).to(torch.bfloat16).to(DEV)
    b1 = (torch.randn(C // 2, generator=g) * 0.5).to(DEV)
    w2 = (torch.randn(C, 3, 3, C // 2, generator=g) / (9 * C // 2) ** 0.5).to(torch.bfloat16).to(DEV)
    b2 = (torch.randn(C, generator=g) * 0.5).to(DEV)
    return x, w1, b1, w2, b2


def _ref(xs, w1, b1, w2, b2):
    xf = xs.float().permute(0, 3, 1, 2)
    h = F.silu(F.conv2d(xf, w1.float().permute(0, 3, 1, 2), b1)).to(torch.bfloat16).float()
    return (F.silu(F.conv2d(h, w2.float().permute(0, 3, 1, 2), b2, padding=1)).to(torch.bfloat16).float() + xf).permute(0, 2, 3, 1)


@pytest.mark.parametrize("C", [128, 64])
@pytest.mark.parametrize("shape", [(1, 16, 16), (1, 8, 16), (1, 5, 7), (2, 23, 37), (1, 40, 33), (3, 17, 130)],
                         ids=lambda s: "x".join(map(str, s)))
def test_bottleneck_ws_kernel(shape, C):
    from adaptiveisp_amd.yolo import _lib
    L = _lib.load()
    B, H, W = shape
    x, w1, b1, w2, b2 = _operands(B, H, W, C, seed=H * 7 + W + C)
    P = lambda t: ctypes.c_void_p(t.data_ptr())                                # noqa: E731
    st = _lib.stream_ptr()
    outs = []
    for _ in range(3):
        out = torch.full((B, H, W, C), float("nan"), dtype=torch.bfloat16, device=DEV)
        assert L.adayolo_bottleneck_ws_fwd(P(x), C, P(w1), P(b1), P(w2), P(b2), P(out), C, B, H, W, C, st) == 0
        outs.append(out)
    torch.cuda.synchronize()
    out = outs[0]
    assert torch.isfinite(out.float()).all(), "unwritten (NaN) outputs"
    assert all(torch.equal(out.view(torch.int16), o.view(torch.int16)) for o in outs[1:]), "run-to-run difference"
    close_scaled("yolo.bottleneck_ws_vs_fp32", out.float(), _ref(x, w1, b1, w2, b2), 2e-2, err_msg=f"{shape} C{C}")
    # the two stand-alone layers: 1x1 on the default ring kernel, 3x3 + residual on the weights-in-registers kernel
    hid = torch.empty(B, H, W, C // 2, dtype=torch.bfloat16, device=DEV)
    two = torch.empty_like(out)
    assert L.adayolo_conv_fwd_variant(P(x), C, P(w1), P(b1), None, 0, P(hid), C // 2, B, H, W, C, C // 2, 1, 1, 1, 22, st) == 0
    assert L.adayolo_conv_fwd_variant(P(hid), C // 2, P(w2), P(b2), P(x), C, P(two), C, B, H, W, C // 2, C, 3, 1, 1, 90, st) == 0
    torch.cuda.synchronize()
    close_scaled("yolo.bottleneck_ws_vs_two_layers", out.float(), two.float(), 2.0 ** -6)      # bf16 roundings of differently ordered sums
    # argument checks of the C-ABI
    assert L.adayolo_bottleneck_ws_fwd(P(x), C, P(w1), P(b1), P(w2), P(b2), P(x), C, B, H, W, C, st) == -1        # in place
    assert L.adayolo_bottleneck_ws_fwd(P(x), C - 6, P(w1), P(b1), P(w2), P(b2), P(out), C, B, H, W, C, st) == -2
    assert L.adayolo_bottleneck_ws_fwd(P(x), C, P(w1), P(b1), P(w2), P(b2), P(out), C, B, H, W, 256, st) == -2     # C = 256: adayolo_bottleneck256_fwd


@pytest.mark.parametrize("C,shape", [(128, (8, 184, 320)), (64, (8, 368, 640))], ids=["C128@184x320", "C64@368x640"])
def test_bottleneck_ws_at_the_baseline_layers(C, shape):
    """The blocks' shapes inside the benchmarked network (8 x 736 x 1280 input): many tiles per persistent workgroup, the
    one-tile-ahead patch prefetch, a ragged last tile column (320 and 640 are multiples of 16; 184 = 23 x 8, 368 = 23 x 16)."""
    from adaptiveisp_amd.yolo import _lib
    L = _lib.load()
    B, H, W = shape
    x, w1, b1, w2, b2 = _operands(B, H, W, C, seed=C)
    P = lambda t: ctypes.c_void_p(t.data_ptr())                                # noqa: E731
    out = torch.full((B, H, W, C), float("nan"), dtype=torch.bfloat16, device=DEV)
    assert L.adayolo_bottleneck_ws_fwd(P(x), C, P(w1), P(b1), P(w2), P(b2), P(out), C, B, H, W, C, _lib.stream_ptr()) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    ref = _ref(x, w1, b1, w2, b2)
    close_scaled("yolo.bottleneck_ws_vs_fp32", out.float(), ref, 2e-2, err_msg=f"{shape} C{C}")
    assert (out.float() - ref).abs().mean().item() <= 2e-3 * max(1.0, ref.abs().max().item())


def test_bottleneck_ws_channel_slices():
    """x is a 128-channel slice of a 192-wide tensor, out a slice of a 160-wide one; what lies outside the output slice keeps its NaN."""
    from adaptiveisp_amd.yolo import _lib
    L = _lib.load()
    B, H, W, C = 2, 19, 35, 128
    x, w1, b1, w2, b2 = _operands(B, H, W, C, seed=3, x_cs=192)
    out = torch.full((B, H, W, 160), float("nan"), dtype=torch.bfloat16, device=DEV)
    P = lambda t: ctypes.c_void_p(t.data_ptr())                                # noqa: E731
    assert L.adayolo_bottleneck_ws_fwd(ctypes.c_void_p(x.data_ptr() + 2 * 64), 192, P(w1), P(b1), P(w2), P(b2),
                                       ctypes.c_void_p(out.data_ptr() + 2 * 32), 160, B, H, W, C, _lib.stream_ptr()) == 0
    torch.cuda.synchronize()
    got = out[..., 32:160].float()
    assert torch.isfinite(got).all()
    close_scaled("yolo.bottleneck_ws_vs_fp32", got, _ref(x[..., 64:192], w1, b1, w2, b2), 2e-2)
    assert torch.isnan(out[..., :32].float()).all()


def test_engine_with_ws_bottlenecks_matches_the_two_launch_plan(monkeypatch):
    """ADAYOLO_BNECK_WS=1 (the three blocks of the C = 64 / C = 128 stages as one launch each; k_stem_down no longer computes the
    first block's cv1) against ADAYOLO_BNECK_WS=0 at the BASELINE shape: the same predictions up to bf16 rounding."""
    from _synth import synth_yolo_state_dict, test_image
    from adaptiveisp_amd.yolo import YoloEngine, yolov3
    m = yolov3().eval()
    m.load_state_dict(synth_yolo_state_dict(m))
    B, H, W = 8, 720, 1280
    x = torch.from_numpy(test_image(B, H, W, seed=5, special=False)).to(DEV)
    monkeypatch.setenv("ADAYOLO_BNECK_WS", "0")
    base = YoloEngine(m, B, H, W, device=DEV)
    base.autotune(cache=TUNE, write=False)
    ref = base(x).clone()
    assert not any(k == "bneckws" for k, _, _ in base.plan)
    monkeypatch.setenv("ADAYOLO_BNECK_WS", "1")
    eng = YoloEngine(m, B, H, W, device=DEV)
    eng.autotune(cache=TUNE, write=False)
    assert sum(k == "bneckws" for k, _, _ in eng.plan) == 3 and eng.fused_ws_blocks == 3
    got = eng(x)
    torch.cuda.synchronize()
    close_scaled("yolo.engine_bneck_ws_vs_two_launch_plan", got, ref, 2e-2)
    # and as a replayed graph
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        eng(x)
    eng.pred.fill_(float("nan"))
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(eng.pred, got)
