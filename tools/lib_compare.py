#!/usr/bin/env python3
"""Where the hand-written kernels stand against the vendor libraries on the detector's layer shapes (measurement only,
nothing here is on the product path): bf16 GEMM through torch (hipBLASLt) for the 1x1 layers, MIOpen bf16 conv
(channels_last) for the 3x3 layers."""
import torch
import torch.nn.functional as F

dev = "cuda"


def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print("1x1 layers as GEMM [M,K]x[N,K]^T (+bias), bf16, torch/hipBLASLt")
for (H, W, cin, cout) in [(92, 160, 256, 128), (46, 80, 512, 256), (23, 40, 1024, 512), (184, 320, 128, 64),
                          (368, 640, 64, 32), (92, 160, 384, 128), (46, 80, 768, 256), (92, 160, 256, 256)]:
    M = 8 * H * W
    a = torch.randn(M, cin, device=dev, dtype=torch.bfloat16)
    w = torch.randn(cout, cin, device=dev, dtype=torch.bfloat16)
    b = torch.randn(cout, device=dev, dtype=torch.bfloat16)
    t = bench(lambda: F.linear(a, w, b))
    t2 = bench(lambda: F.silu(F.linear(a, w, b)))
    fl = 2.0 * M * cin * cout
    by = (M * cin + M * cout) * 2
    print(f"  {H}x{W} {cin}->{cout}: linear {t:7.1f} us {fl / t / 1e6:7.1f} TF {by / t / 1e3:6.0f} GB/s | +silu (2 kernels) {t2:7.1f} us")

print("3x3 layers, MIOpen bf16 channels_last")
torch.backends.cudnn.benchmark = True
for (H, W, cin, cout, s) in [(92, 160, 128, 256, 1), (46, 80, 256, 512, 1), (23, 40, 512, 1024, 1), (184, 320, 64, 128, 1),
                             (368, 640, 32, 64, 1), (184, 320, 128, 256, 2)]:
    x = torch.randn(8, cin, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 3, 3, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    try:
        t = bench(lambda: F.conv2d(x, w, None, stride=s, padding=1), n=10)
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        fl = 2.0 * 8 * Ho * Wo * cout * 9 * cin
        print(f"  {H}x{W} {cin}->{cout} s{s}: {t:7.1f} us {fl / t / 1e6:7.1f} TF")
    except Exception as e:
        print(f"  {H}x{W} {cin}->{cout} s{s}: failed {type(e).__name__}: {e}")
