#!/usr/bin/env python3
"""Time of the fused stem + down-sampling kernel vs the two separate launches (bs 8, 1280x720)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptiveisp_amd.yolo import YoloEngine, yolov3, _lib
from adaptiveisp_amd.yolo.engine import LETTERBOX_VALUE
torch.manual_seed(1)
eng = YoloEngine(yolov3().eval(), 8, 720, 1280)
eng.autotune(cache=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "adaptiveisp_amd", "yolo", "tuning", "mi355x.json"))
x = torch.rand(8, 3, 720, 1280, device="cuda")
L, st = eng.L, _lib.stream_ptr()
w, b, out = eng._stem
d = eng._head_down
def fused():
    L.adayolo_stem_down_fwd(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(d["w"].data_ptr()),
                            ctypes.c_void_p(d["b"].data_ptr()), ctypes.c_void_p(d["dst"].ptr), d["dst"].cs, eng.B, eng.H, eng.W, eng.Hp, eng.pad_top, LETTERBOX_VALUE, None, None, None, 0, st)
n = eng._head_next
def fused3():
    L.adayolo_stem_down_fwd(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(d["w"].data_ptr()),
                            ctypes.c_void_p(d["b"].data_ptr()), ctypes.c_void_p(d["dst"].ptr), d["dst"].cs, eng.B, eng.H, eng.W, eng.Hp, eng.pad_top, LETTERBOX_VALUE,
                            ctypes.c_void_p(n["w"].data_ptr()), ctypes.c_void_p(n["b"].data_ptr()), ctypes.c_void_p(n["dst"].ptr), n["dst"].cs, st)
def conv2():
    _, fn, args = eng.plan[2]
    fn(*args, st)
def stem():
    L.adayolo_stem_fwd(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(out.ptr), out.cs,
                       eng.B, eng.H, eng.W, eng.Hp, eng.pad_top, LETTERBOX_VALUE, 32, st)
def conv1():
    _, fn, args = eng.plan[1]
    fn(*args, st)
for name, fn in (("fused stem+down", fused), ("fused stem+down+1x1", fused3), ("stem", stem), ("down conv", conv1), ("1x1 conv", conv2)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:20s} {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us")
