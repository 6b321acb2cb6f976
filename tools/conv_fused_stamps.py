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
for _ in range(3):
    rc = L.adayolo_conv_fused1x1_fwd(P(x), cin, P(w), P(b), P(res), 256, P(out), 256, B, H, W, cin, 256, k, s, 1, P(w2p), P(b2), P(out2), 128, 128, _lib.stream_ptr())
    assert rc == 0
torch.cuda.synchronize()
n = 460
buf = np.zeros(n * 8, np.uint64)
assert L.adayolo_debug_stamps(buf.ctypes.data, n * 8) == 0
t = buf.reshape(n, 8).astype(np.float64)
names = ["args", "prologue", "k-loop", "tail+epilogue+writeback (3->6)", "wait+barrier1 (6->4)", "GEMM+barrier2 (4->5)", "out2 epilogue+drain (5->7)"]
order = [0, 1, 2, 3, 6, 4, 5, 7]
tt = t[:, order]
d = np.diff(tt, axis=1)
for i, nm in enumerate(names):
    print(f"{nm:36s} median {np.median(d[:, i]):8.0f} p90 {np.percentile(d[:, i], 90):8.0f}")
print("whole wg median", np.median(tt[:, -1] - tt[:, 0]))
# the unfused kernel on the same layer (variant 57 = stamped build), with the residual
args = (P(x), cin, P(w), P(b), P(res), 256, P(out), 256, B, H, W, cin, 256, k, s, 1, 57)
for _ in range(3):
    assert L.adayolo_conv_fwd_variant(*args, _lib.stream_ptr()) == 0
torch.cuda.synchronize()
assert L.adayolo_debug_stamps(buf.ctypes.data, n * 8) == 0
t = buf.reshape(n, 8).astype(np.float64)[:, [0, 1, 2, 3, 6, 7]]
d = np.diff(t, axis=1)
print("unfused, with residual:")
for i, nm in enumerate(["args", "prologue", "k-loop", "tail+epilogue (3->6)", "store drain (6->7)"]):
    print(f"{nm:36s} median {np.median(d[:, i]):8.0f} p90 {np.percentile(d[:, i], 90):8.0f}")
print("whole wg median", np.median(t[:, -1] - t[:, 0]))
