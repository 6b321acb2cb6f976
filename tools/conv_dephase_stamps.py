"""Stamps of the 256x256 ping-pong conv (unfused with residual, and the fused pair) on 128->256 k3 @8x92x160, per group of
workgroups (late starters of a -DPP_DEPHASE build vs the others). Needs a -DADAYOLO_MEASURE build (ADAYOLO_LIB=...)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from adaptiveisp_amd.yolo import _lib
L = _lib.load()
L.adayolo_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
B, H, W, cin, k, s = 8, 92, 160, 128, 3, 1
g = torch.Generator(device="cpu").manual_seed(0)
x = torch.randn(B, H, W, cin, generator=g).to(torch.bfloat16).cuda()
w = (torch.randn(256, k, k, cin, generator=g) / (k * k * cin) ** 0.5).to(torch.bfloat16).cuda()
b = torch.randn(256, generator=g).cuda()
res = torch.randn(B, H, W, 256, generator=g).to(torch.bfloat16).cuda()
w2 = (torch.randn(128, 256, generator=g) / 16).to(torch.bfloat16).cuda()
w2p = w2.reshape(4, 32, 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous()
b2 = torch.randn(128, generator=g).cuda()
out = torch.zeros(B, H, W, 256, dtype=torch.bfloat16, device="cuda"); out2 = torch.zeros(B, H, W, 128, dtype=torch.bfloat16, device="cuda")
P = lambda t: ctypes.c_void_p(t.data_ptr())
n = 460


def report(t, names, cols):
    tt = t[:, cols]
    d = np.diff(tt, axis=1)
    late = ((np.arange(n) >> 3) & 1).astype(bool)
    start = tt[:, 0] - tt[:, 0].min()
    for grp, sel in (("all", np.ones(n, bool)), ("round 1", start < 20000), ("round 2", start >= 20000)):
        if not sel.any():
            continue
        print(f"  [{grp}: {sel.sum()} workgroups] whole wg median {np.median(tt[sel, -1] - tt[sel, 0]):.0f}; start spread p10/p50/p90 "
              f"{np.percentile(start[sel], 10):.0f}/{np.percentile(start[sel], 50):.0f}/{np.percentile(start[sel], 90):.0f}")
        for i, nm in enumerate(names):
            print(f"    {nm:36s} median {np.median(d[sel, i]):8.0f} p10 {np.percentile(d[sel, i], 10):8.0f} p90 {np.percentile(d[sel, i], 90):8.0f}")
    print(f"  kernel span (first stamp -> last stamp) {tt[:, -1].max() - tt[:, 0].min():.0f} cycles")


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


buf = np.zeros(n * 8, np.uint64)
os.environ["ADAYOLO_PP_STAMPS"] = "1"
fused = lambda: L.adayolo_conv_fused1x1_fwd(P(x), cin, P(w), P(b), P(res), 256, P(out), 256, B, H, W, cin, 256, k, s, 1, P(w2p), P(b2), P(out2), 128, 128, _lib.stream_ptr())
for _ in range(3):
    assert fused() == 0
torch.cuda.synchronize()
assert L.adayolo_debug_stamps(buf.ctypes.data, n * 8) == 0
print(f"fused pair (stamped build): {timed(fused):.1f} us per launch")
report(buf.reshape(n, 8).astype(np.float64),
       ["args", "prologue", "k-loop", "tail+epilogue+writeback (3->6)", "wait+barrier1 (6->4)", "GEMM+barrier2 (4->5)", "out2 epilogue+drain (5->7)"],
       [0, 1, 2, 3, 6, 4, 5, 7])
args = (P(x), cin, P(w), P(b), P(res), 256, P(out), 256, B, H, W, cin, 256, k, s, 1, 57)
unf = lambda: L.adayolo_conv_fwd_variant(*args, _lib.stream_ptr())
for _ in range(3):
    assert unf() == 0
torch.cuda.synchronize()
assert L.adayolo_debug_stamps(buf.ctypes.data, n * 8) == 0
print(f"unfused, with residual (stamped build): {timed(unf):.1f} us per launch")
report(buf.reshape(n, 8).astype(np.float64), ["args", "prologue", "k-loop", "tail+epilogue (3->6)", "store drain (6->7)"], [0, 1, 2, 3, 6, 7])
del os.environ["ADAYOLO_PP_STAMPS"]
args50 = args[:-1] + (50,)
print(f"unfused v50 (no stamps): {timed(lambda: L.adayolo_conv_fwd_variant(*args50, _lib.stream_ptr())):.1f} us;  fused (no stamps): {timed(fused):.1f} us")
