#!/usr/bin/env python3
"""Experiment: run the first N launches of the detector depth-first over batch chunks, so that the big early
activations (stem output: 482 MB for 8 images) are consumed while still in the 256 MB Infinity Cache.
usage: chunk_experiment.py"""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptiveisp_amd.yolo import YoloEngine, yolov3, _lib
from adaptiveisp_amd.yolo.engine import LETTERBOX_VALUE
torch.manual_seed(1)
eng = YoloEngine(yolov3().eval(), 8, 720, 1280)
eng.autotune(cache=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "adaptiveisp_amd", "yolo", "tuning", "mi355x.json"))
x = torch.rand(8, 3, 720, 1280, device="cuda")
ref = eng(x).clone(); torch.cuda.synchronize()
L = eng.L

def run(nchunks, nlayers):
    st = _lib.stream_ptr()
    B = eng.B
    Bc = B // nchunks
    head = eng.plan[:nlayers]
    for c in range(nchunks):
        for kind, fn, args in head:
            if kind == "stem":
                w, b, out = eng._stem
                ip = x.data_ptr() + c * Bc * 3 * eng.H * eng.W * 4
                op = out.ptr + c * Bc * eng.Hp * eng.W * out.cs * 2
                L.adayolo_stem_fwd(ctypes.c_void_p(ip), ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(op), out.cs,
                                   Bc, eng.H, eng.W, eng.Hp, eng.pad_top, LETTERBOX_VALUE, 32, st)
            elif kind == "conv":
                a = list(args)
                H, W, k, s = a[9], a[10], a[13], a[14]
                Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
                a[0] = ctypes.c_void_p(a[0].value + c * Bc * H * W * a[1] * 2)
                if a[4] is not None and a[4].value:
                    a[4] = ctypes.c_void_p(a[4].value + c * Bc * Ho * Wo * a[5] * 2)
                a[6] = ctypes.c_void_p(a[6].value + c * Bc * Ho * Wo * a[7] * 2)
                a[8] = Bc
                fn(*a, st)
            else:
                raise SystemExit("only stem/conv entries can be chunked: " + kind)
    for kind, fn, args in eng.plan[nlayers:]:
        if kind == "stem":
            w, b, out = eng._stem
            L.adayolo_stem_fwd(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(out.ptr), out.cs,
                               B, eng.H, eng.W, eng.Hp, eng.pad_top, LETTERBOX_VALUE, 32, st)
        else:
            fn(*args, st)

for nlayers in (0, 2, 4, 6, 9, 12):
    for nchunks in ((1,) if nlayers == 0 else (2, 4, 8)):
        run(nchunks, nlayers); torch.cuda.synchronize()
        ok = torch.equal(eng.pred, ref)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run(nchunks, nlayers)
        e1.record(); torch.cuda.synchronize()
        print(f"first {nlayers:2d} launches in {nchunks} chunk(s): {e0.elapsed_time(e1) / 10:.3f} ms per forward, identical output: {ok}")
print([k for k, _, _ in eng.plan[:14]])
