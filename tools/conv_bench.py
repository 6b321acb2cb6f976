#!/usr/bin/env python3
"""A/B of the implicit-GEMM conv variants on the YOLOv3 layer shapes (bs8, 1280x736). TFLOP/s per shape."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptiveisp_amd.yolo import _lib  # noqa: E402

DEV = "cuda:0"
# (H, W, Cin, Cout, k, s, count in the network)
SHAPES = [
    (736, 1280, 32, 64, 3, 2, 1), (368, 640, 64, 32, 1, 1, 1), (368, 640, 32, 64, 3, 1, 1),
    (368, 640, 64, 128, 3, 2, 1), (184, 320, 128, 64, 1, 1, 2), (184, 320, 64, 128, 3, 1, 2),
    (184, 320, 128, 256, 3, 2, 1), (92, 160, 256, 128, 1, 1, 10), (92, 160, 128, 256, 3, 1, 11),
    (92, 160, 256, 512, 3, 2, 1), (46, 80, 512, 256, 1, 1, 10), (46, 80, 256, 512, 3, 1, 11),
    (46, 80, 512, 1024, 3, 2, 1), (23, 40, 1024, 512, 1, 1, 7), (23, 40, 512, 1024, 3, 1, 7),
    (46, 80, 768, 256, 1, 1, 1), (92, 160, 384, 128, 1, 1, 1), (92, 160, 256, 256, 1, 1, 1),
]


def run(L, x, w, b, out, B, H, W, cin, cout, k, s, variant, iters):
    args = (ctypes.c_void_p(x.data_ptr()), cin, ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(b.data_ptr()), None, 0,
            ctypes.c_void_p(out.data_ptr()), cout, B, H, W, cin, cout, k, s, 1, variant)
    st = _lib.stream_ptr()
    for _ in range(2):
        _lib.check(L.adayolo_conv_fwd_variant(*args, st), "conv")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        L.adayolo_conv_fwd_variant(*args, st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    global SHAPES
    variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "2,5,27,50,60").split(",")]
    B = 8
    L = _lib.load()
    if len(sys.argv) > 2 and sys.argv[2] == "small":        # the shallow 3x3 layers (Cin 32 / 64)
        SHAPES[:] = [sh for sh in SHAPES if sh[4] == 3 and sh[2] <= 64]
    if len(sys.argv) > 2 and sys.argv[2] == "big":          # the MFMA-bound 3x3 layers only
        SHAPES[:] = [sh for sh in SHAPES if sh[4] == 3 and sh[2] >= 128]
    tot = {v: 0.0 for v in variants}
    totfl = 0.0
    print(f"{'shape':38s} " + " ".join(f"v{v:>2d} TF/s   ms " for v in variants))
    for (H, W, cin, cout, k, s, cnt) in SHAPES:
        g = torch.Generator(device="cpu").manual_seed(H + cin)
        x = torch.randn(B, H, W, cin, generator=g).to(torch.bfloat16).to(DEV)
        w = (torch.randn(cout, k, k, cin, generator=g) / (k * k * cin) ** 0.5).to(torch.bfloat16).to(DEV)
        b = torch.randn(cout, generator=g).to(DEV)
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        fl = 2.0 * B * Ho * Wo * cout * k * k * cin
        outs, row = {}, []
        for v in variants:
            out = torch.zeros(B, Ho, Wo, cout, dtype=torch.bfloat16, device=DEV)
            ms = run(L, x, w, b, out, B, H, W, cin, cout, k, s, v, 10)
            outs[v] = out
            tot[v] += ms * cnt
            row.append(f"{fl / ms / 1e9:7.1f} {ms:6.3f}")
        totfl += fl * cnt
        dmax = max([(outs[variants[0]].float() - outs[v].float()).abs().max().item() for v in variants[1:]] + [0.0])
        print(f"{H}x{W} {cin:4d}->{cout:4d} k{k} s{s} x{cnt:<2d}          " + "  ".join(row) + f"  maxdiff {dmax:.3g}")
        if (90 in variants or 91 in variants) and k == 3 and s == 1 and cin in (32, 64):
            try:
                d = (ctypes.c_ulonglong * 16)()
                torch.cuda.synchronize()
                if L.adayolo_debug_ws(d) == 0 and d[6]:
                    names = ["wait patch", "barrier A", "MFMA steps", "residual issue + SiLU + output tile", "barrier B + patch issue", "rows + stores"]
                    print("      ws wg0, cycles per tile: " + ", ".join(f"{n} {d[i] / d[6]:.0f}" for i, n in enumerate(names)) + f"  ({d[6]} tiles)")
            except AttributeError:
                pass
    if hasattr(L, "adayolo_debug_ws") and 90 in variants:
        pass
    print("network conv total (ms), TF/s: " + "  ".join(f"v{v}: {tot[v]:.3f} ms {totfl / tot[v] / 1e9:.1f}" for v in variants))


if __name__ == "__main__":
    main()
