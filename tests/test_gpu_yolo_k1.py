"""The "whole K at once" 1x1 conv kernel (adayolo_conv1x1_stream_fwd, csrc/yolo_conv_k1.hip) against fp32 `F.conv2d` on the same
bf16-rounded operands and against the ring kernel the tuned engine used before it (Conv.forward_fuse of a k = 1 layer:
yolov3/models/common.py:45-59; Bottleneck.cv1: :110-120). NaN-prefilled outputs, four launches bit-identical, ragged M,
channel-slice strides, and the shapes the entry point refuses."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from _margins import close_scaled

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def pack_fragments(w):
    """[Cout][Cin] -> [Cout/32][Cin/16][2][32][8] (include/adayolo.h)."""
    cout, cin = w.shape
    return w.reshape(cout // 32, 32, cin // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous()


def _operands(B, H, W, cin, cout, seed, in_cs=None, out_cs=None):
    g = torch.Generator(device="cpu").manual_seed(seed)
    in_cs, out_cs = in_cs or cin, out_cs or cout
    xb = torch.randn(B, H, W, in_cs, generator=g).to(torch.bfloat16).to(DEV)
    w = (torch.randn(cout, cin, generator=g) / cin ** 0.5).to(torch.bfloat16).to(DEV)
    b = torch.randn(cout, generator=g).to(DEV)
    return xb, w, b


def _k1(xb, cin, w, b, cout, act, out_cs=None, coff_in=0, coff_out=0, reps=4):
    from adaptiveisp_amd.yolo import _lib
    L = _lib.load()
    B, H, W, in_cs = xb.shape
    out_cs = out_cs or cout
    wp = pack_fragments(w)
    first = None
    for _ in range(reps):
        out = torch.full((B, H, W, out_cs), float("nan"), dtype=torch.bfloat16, device=DEV)
        rc = L.adayolo_conv1x1_stream_fwd(ctypes.c_void_p(xb.data_ptr() + 2 * coff_in), in_cs, ctypes.c_void_p(wp.data_ptr()),
                                          ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(out.data_ptr() + 2 * coff_out), out_cs,
                                          B, H, W, cin, cout, act, _lib.stream_ptr())
        _lib.check(rc, "conv1x1_stream")
        torch.cuda.synchronize()
        if first is None:
            first = out
        else:
            assert torch.equal(first.view(torch.int16), out.view(torch.int16)), "run-to-run difference"
    return first


def _ref(x, w, b, act):
    r = F.conv2d(x.float().permute(0, 3, 1, 2), w.float()[:, :, None, None], b)
    if act:
        r = F.silu(r)
    return r.permute(0, 2, 3, 1)


@pytest.mark.parametrize("shape", [(8, 46, 80, 512, 256), (8, 92, 160, 256, 256), (8, 23, 40, 512, 256), (1, 7, 9, 512, 512),
                                   (2, 5, 13, 256, 512), (4, 270, 480, 512, 256)],
                         ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("act", [1, 0])
def test_k1_vs_fp32(shape, act):
    B, H, W, cin, cout = shape
    xb, w, b = _operands(B, H, W, cin, cout, seed=H * 131 + cin + cout + act)
    out = _k1(xb, cin, w, b, cout, act)
    ref = _ref(xb, w, b, act)
    assert torch.isfinite(out.float()).all(), "unwritten (NaN) outputs"
    scale = max(1.0, ref.abs().max().item())
    close_scaled("yolo.k1_vs_fp32", out.float(), ref, 2e-2, err_msg=f"{shape} act{act}")
    assert (out.float() - ref).abs().mean().item() <= 2e-3 * scale


def test_k1_equals_the_ring_kernel_bit_for_bit_or_within_one_rounding():
    """Same operands through adayolo_conv_fwd_variant(60) (the 256 x 128 ring kernel the tuning table names for 512 -> 256 on
    46 x 80 maps): both accumulate k in ascending order in fp32 MFMA chains of 16 — the results agree to the last bf16 bit
    except where the accumulation grouping (BK = 64 tiles vs one chain) lands on a rounding boundary."""
    from adaptiveisp_amd.yolo import _lib
    L = _lib.load()
    B, H, W, cin, cout = 8, 46, 80, 512, 256
    xb, w, b = _operands(B, H, W, cin, cout, seed=77)
    out = _k1(xb, cin, w, b, cout, 1)
    ring = torch.full((B, H, W, cout), float("nan"), dtype=torch.bfloat16, device=DEV)
    w4 = w.reshape(cout, 1, 1, cin).contiguous()
    _lib.check(L.adayolo_conv_fwd_variant(ctypes.c_void_p(xb.data_ptr()), cin, ctypes.c_void_p(w4.data_ptr()), ctypes.c_void_p(b.data_ptr()),
                                          None, 0, ctypes.c_void_p(ring.data_ptr()), cout, B, H, W, cin, cout, 1, 1, 1, 60,
                                          _lib.stream_ptr()), "ring")
    torch.cuda.synchronize()
    d = (out.float() - ring.float()).abs()
    same = (out.view(torch.int16) == ring.view(torch.int16)).float().mean().item()
    assert same > 0.99, same
    assert d.max().item() <= 2.0 ** -7 * max(1.0, ring.float().abs().max().item())       # one bf16 ulp of the largest value


def test_k1_channel_slices_and_ragged_tiles():
    """Reads a 512-channel slice of a 768-wide tensor (Concat input), writes a 256-channel slice of a 384-wide one; M = 3 x 11 x 17
    = 561 pixels (4 full tiles + 49 pixels); channels outside the output slice keep their NaN."""
    B, H, W, cin, cout = 3, 11, 17, 512, 256
    xb, w, b = _operands(B, H, W, cin, cout, seed=5, in_cs=768)
    out = _k1(xb, cin, w, b, cout, 1, out_cs=384, coff_in=256, coff_out=128)
    ref = _ref(xb[..., 256:768], w, b, 1)
    got = out[..., 128:384].float()
    assert torch.isfinite(got).all()
    close_scaled("yolo.k1_vs_fp32", got, ref, 2e-2, err_msg="slices")
    assert torch.isnan(out[..., :128].float()).all()


def test_k1_entry_refuses_other_shapes():
    from adaptiveisp_amd.yolo import _lib
    L = _lib.load()
    xb, w, b = _operands(1, 8, 8, 512, 256, seed=1)
    out = torch.zeros(1, 8, 8, 256, dtype=torch.bfloat16, device=DEV)
    vp = ctypes.c_void_p

    def call(cin=512, cout=256, in_cs=512, out_cs=256, x=xb, o=out):
        return L.adayolo_conv1x1_stream_fwd(vp(x.data_ptr()), in_cs, vp(w.data_ptr()), vp(b.data_ptr()), vp(o.data_ptr()), out_cs,
                                            1, 8, 8, cin, cout, 1, _lib.stream_ptr())
    assert call() == 0
    assert call(cin=384, in_cs=384) == -2            # ADAYOLO_ESHAPE: Cin not in {256, 512}
    assert call(cout=128, out_cs=128) == -2          # Cout % 256
    assert call(in_cs=500) == -2                     # strides are multiples of 8 and >= the channel count
    assert L.adayolo_conv1x1_stream_fwd(None, 512, vp(w.data_ptr()), vp(b.data_ptr()), vp(out.data_ptr()), 256, 1, 8, 8, 512, 256, 1,
                                        _lib.stream_ptr()) == -1
    torch.cuda.synchronize()
