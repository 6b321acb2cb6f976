#!/usr/bin/env python3
"""Split-K sweep: one conv shape per argument (B,H,W,Cin,Cout,k,s), the unsplit kernels (variants 5, 60, 85) beside every
split S the library serves; median of 5 rounds of 20 launches, residual + SiLU as in a Bottleneck."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptiveisp_amd.yolo import _lib
L = _lib.load()
st = _lib.stream_ptr()
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [
    (8, 16, 16, 1024, 512, 3, 1), (8, 16, 16, 512, 1024, 3, 1), (8, 32, 32, 512, 256, 3, 1), (8, 32, 32, 256, 512, 3, 1),
    (8, 16, 16, 1024, 512, 1, 1), (8, 32, 32, 512, 256, 1, 1)]
P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
for (B, H, W, cin, cout, k, s) in shapes:
    g = torch.Generator(device="cpu").manual_seed(H + cin)
    x = torch.randn(B, H, W, cin, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(cout, k, k, cin, generator=g) / (k * k * cin) ** 0.5).to(torch.bfloat16).cuda()
    b = torch.randn(cout, generator=g).cuda()
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    res = torch.randn(B, Ho, Wo, cout, generator=g).to(torch.bfloat16).cuda()
    out = torch.zeros(B, Ho, Wo, cout, dtype=torch.bfloat16, device="cuda")
    fl = 2.0 * B * Ho * Wo * cout * k * k * cin
    cands = [5, 60, 85] + [v for v in range(102, 117) if L.adayolo_conv_splitk_workspace_bytes(B, H, W, cin, cout, k, s, v) > 0]
    ws = torch.zeros(max([16] + [L.adayolo_conv_splitk_workspace_bytes(B, H, W, cin, cout, k, s, v) for v in cands if v >= 100]), dtype=torch.uint8, device="cuda")
    line = []
    for v in cands:
        def run(n):
            for _ in range(n):
                if v >= 100:
                    rc = L.adayolo_conv_splitk_fwd(P(x), cin, P(w), P(b), P(res), cout, P(out), cout, None, 0, B, H, W, cin, cout, k, s, 1, v, P(ws), ws.numel(), st)
                else:
                    rc = L.adayolo_conv_fwd_variant(P(x), cin, P(w), P(b), P(res), cout, P(out), cout, B, H, W, cin, cout, k, s, 1, v, st)
                assert rc == 0, rc
        ts = []
        for _ in range(5):
            run(3); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(20); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        t = sorted(ts)[2]
        line.append(f"v{v} {t:.1f}us ({fl / t / 1e6:.0f} TF/s)")
        if v >= 100 and os.environ.get("ADAYOLO_SPLITK_PROBE"):
            ws.zero_()
    print(f"{B}x{H}x{W} {cin}->{cout} k{k}s{s}: " + "  ".join(line), flush=True)
