#!/usr/bin/env python3
"""Race screen + accuracy check of a conv variant against the fp32 torch conv and against repeated runs of itself.
usage: conv_pp_check.py [variant=50] [repeats=20]"""
import ctypes, os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptiveisp_amd.yolo import _lib
L = _lib.load()
V = int(sys.argv[1]) if len(sys.argv) > 1 else 50
REP = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = 0
# B, H, W, cin, cout, k, s, res
for (B, H, W, cin, cout, k, s, use_res) in [(8, 92, 160, 128, 256, 3, 1, True), (8, 46, 80, 256, 512, 3, 1, True),
                                            (8, 23, 40, 512, 1024, 3, 1, False), (8, 92, 160, 256, 512, 3, 2, False),
                                            (8, 23, 40, 1024, 512, 1, 1, False), (8, 46, 80, 768, 256, 1, 1, False),
                                            (2, 19, 33, 128, 256, 3, 1, True), (1, 5, 6, 512, 1024, 3, 1, True),
                                            (3, 9, 11, 64, 256, 1, 1, False), (1, 7, 9, 64, 256, 3, 2, False),
                                            (8, 184, 320, 128, 256, 3, 2, False), (8, 92, 160, 256, 128, 1, 1, False),
                                            (4, 184, 320, 64, 128, 3, 1, True), (2, 37, 53, 64, 128, 3, 2, False),
                                            (1, 9, 7, 192, 384, 3, 1, True), (8, 368, 640, 32, 64, 3, 1, True), (2, 33, 47, 64, 128, 3, 1, True),
                                            (1, 16, 16, 32, 64, 3, 1, False), (3, 17, 5, 64, 64, 3, 1, False)]:
    g = torch.Generator(device="cpu").manual_seed(H * 131 + cin)
    x = torch.randn(B, H, W, cin, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(cout, k, k, cin, generator=g) / (k * k * cin) ** 0.5).to(torch.bfloat16).cuda()
    b = torch.randn(cout, generator=g).cuda()
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    res = torch.randn(B, Ho, Wo, cout, generator=g).to(torch.bfloat16).cuda() if use_res else None
    ref = F.silu(F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b, stride=s, padding=k // 2)).permute(0, 2, 3, 1)
    if use_res:
        ref = ref.to(torch.bfloat16).float() + res.float()
    first, worst, nondet = None, 0.0, 0
    for it in range(REP):
        out = torch.full((B, Ho, Wo, cout), float("nan"), dtype=torch.bfloat16, device="cuda")
        rc = L.adayolo_conv_fwd_variant(ctypes.c_void_p(x.data_ptr()), cin, ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(b.data_ptr()),
                                        ctypes.c_void_p(res.data_ptr()) if use_res else None, cout if use_res else 0,
                                        ctypes.c_void_p(out.data_ptr()), cout, B, H, W, cin, cout, k, s, 1, V, _lib.stream_ptr())
        assert rc == 0
        torch.cuda.synchronize()
        err = (out.float() - ref).abs().max().item()
        worst = max(worst, err if err == err else float("inf"))
        if first is None:
            first = out.clone()
        elif not torch.equal(first.view(torch.int16), out.view(torch.int16)):
            nondet += 1
    tol = 2e-2 * max(1.0, ref.abs().max().item())
    ok = worst <= tol and nondet == 0
    bad += not ok
    print(f"{B}x{H}x{W} {cin}->{cout} k{k}s{s} res={use_res}: max err {worst:.4g} (tol {tol:.3g}) nondeterministic runs {nondet}/{REP-1} {'OK' if ok else 'FAIL'}")
print("ALL OK" if bad == 0 else f"{bad} FAILED")
sys.exit(1 if bad else 0)
