#!/usr/bin/env python3
"""adayolo_bottleneck256_fwd against the two stand-alone layers (1x1 256 -> 128, then 3x3 128 -> 256 + residual) and against an
fp32 reference; timing next to the round-3 arrangement (fused pair [3x3 + res | next 1x1]) at the BASELINE shape 8 x 92 x 160."""
import ctypes, os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptiveisp_amd.yolo import _lib
L = _lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = _lib.stream_ptr()


def make(B, H, W, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(B, H, W, 256, generator=g).to(torch.bfloat16).cuda()
    w1 = (torch.randn(128, 1, 1, 256, generator=g) / 16).to(torch.bfloat16).cuda()
    b1 = torch.randn(128, generator=g).cuda() * 0.5
    w2 = (torch.randn(256, 3, 3, 128, generator=g) / (9 * 128) ** 0.5).to(torch.bfloat16).cuda()
    b2 = torch.randn(256, generator=g).cuda() * 0.5
    return x, w1, b1, w2, b2


def bneck(x, w1, b1, w2, b2, out):
    B, H, W, _ = x.shape
    rc = L.adayolo_bottleneck256_fwd(P(x), 256, P(w1), P(b1), P(w2), P(b2), P(out), 256, B, H, W, st)
    assert rc == 0, rc


def two_layers(x, w1, b1, w2, b2, h, out, v1=85, v2=50):
    B, H, W, _ = x.shape
    rc = L.adayolo_conv_fwd_variant(P(x), 256, P(w1), P(b1), None, 0, P(h), 128, B, H, W, 256, 128, 1, 1, 1, v1, st)
    assert rc == 0, rc
    rc = L.adayolo_conv_fwd_variant(P(h), 128, P(w2), P(b2), P(x), 256, P(out), 256, B, H, W, 128, 256, 3, 1, 1, v2, st)
    assert rc == 0, rc


def reference(x, w1, b1, w2, b2):
    xf = x.float().permute(0, 3, 1, 2)
    h = F.silu(F.conv2d(xf, w1.float().permute(0, 3, 1, 2), b1)).to(torch.bfloat16).float()
    y = F.silu(F.conv2d(h, w2.float().permute(0, 3, 1, 2), b2, padding=1)).to(torch.bfloat16).float()
    return (y + xf).to(torch.bfloat16).permute(0, 2, 3, 1).contiguous()


bad = 0
for (B, H, W) in [(1, 16, 16), (1, 5, 7), (2, 23, 37), (1, 40, 33), (8, 92, 160)]:
    x, w1, b1, w2, b2 = make(B, H, W, seed=H)
    out = torch.full((B, H, W, 256), float("nan"), dtype=torch.bfloat16, device="cuda")
    h = torch.empty(B, H, W, 128, dtype=torch.bfloat16, device="cuda")
    out2 = torch.empty_like(out)
    bneck(x, w1, b1, w2, b2, out)
    two_layers(x, w1, b1, w2, b2, h, out2)
    torch.cuda.synchronize()
    ref = reference(x, w1, b1, w2, b2)
    same = torch.equal(out.view(torch.int16), out2.view(torch.int16))
    d_seq = (out.float() - out2.float()).abs().max().item()
    d_ref = (out.float() - ref.float()).abs().max().item()
    d_ref2 = (out2.float() - ref.float()).abs().max().item()
    scale = ref.float().abs().max().item()
    nan = int(torch.isnan(out.float()).sum())
    print(f"{B}x{H}x{W}: identical to the two layers {same} (max diff {d_seq:.4g}); vs fp32 reference {d_ref:.4g} (two layers: {d_ref2:.4g}), scale {scale:.3g}, NaN {nan}")
    if nan or d_ref > 3e-2 * max(1.0, scale):
        bad += 1
    outs = []
    for _ in range(3):
        o = torch.empty_like(out); bneck(x, w1, b1, w2, b2, o); outs.append(o)
    torch.cuda.synchronize()
    assert all(torch.equal(outs[0].view(torch.int16), o.view(torch.int16)) for o in outs), "run-to-run difference"
print("FAILED" if bad else "ok")


def timed(fn, reps=30):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


B, H, W = 8, 92, 160
x, w1, b1, w2, b2 = make(B, H, W)
xs = [x] + [x.clone() for _ in range(3)]            # rotate: block i+1 reads what block i wrote, never the same tensor twice
out = torch.empty_like(x); h = torch.empty(B, H, W, 128, dtype=torch.bfloat16, device="cuda"); h2 = torch.empty_like(h)
w1p = w1.reshape(128, 256).reshape(4, 32, 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous()
state = {"i": 0}


def chain_bneck():
    i = state["i"]; state["i"] = (i + 1) % 4
    bneck(xs[i], w1, b1, w2, b2, xs[(i + 1) % 4])


def chain_pair():      # round 3: [3x3 128->256 + res | next block's 1x1 256->128] per launch
    i = state["i"]; state["i"] = (i + 1) % 4
    rc = L.adayolo_conv_fused1x1_fwd(P(h), 128, P(w2), P(b2), P(xs[i]), 256, P(xs[(i + 1) % 4]), 256, B, H, W, 128, 256, 3, 1, 1, P(w1p), P(b1), P(h2), 128, 128, st)
    assert rc == 0


def chain_two():
    i = state["i"]; state["i"] = (i + 1) % 4
    two_layers(xs[i], w1, b1, w2, b2, h, xs[(i + 1) % 4])


fl = 2.0 * B * H * W * (256 * 128 + 9 * 128 * 256)
for name, fn in (("bottleneck kernel", chain_bneck), ("fused pair (round 3)", chain_pair), ("two launches (v85 + v50)", chain_two)):
    ts = sorted(timed(fn) for _ in range(5))
    print(f"{name:28s} {ts[2]:7.1f} us (min {ts[0]:.1f})  {fl / ts[2] / 1e6:7.1f} TFLOP/s per Bottleneck")
if hasattr(L, "adayolo_debug_bneck_stamps"):
    import numpy as np
    n = B * 6 * 10
    buf = np.zeros(n * 8, np.uint64)
    chain_bneck(); torch.cuda.synchronize()
    L.adayolo_debug_bneck_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert L.adayolo_debug_bneck_stamps(buf.ctypes.data, n * 8) == 0
    t = buf.reshape(n, 8).astype(np.float64)[:, :6]
    d = np.diff(t, axis=1)
    for i, nm in enumerate(["set-up + first DMA", "stage A k-loop", "W2 issue + h epilogue", "stage B k-loop", "epilogue"]):
        print(f"   {nm:26s} median {np.median(d[:, i]):8.0f} p10 {np.percentile(d[:, i], 10):8.0f} p90 {np.percentile(d[:, i], 90):8.0f}")
    print("   whole workgroup median", np.median(t[:, 5] - t[:, 0]))
