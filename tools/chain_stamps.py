#!/usr/bin/env python3
"""Where a tile's cycles go INSIDE the persistent chain (measurement build: tools/build_variant.py yolo_chainstamps yolo
-DADAYOLO_CHAIN_STAMPS; run with ADAYOLO_LIB=build/variants/yolo_chainstamps/libadayolo.so): thread 0 of every workgroup adds
the cycles between consecutive stamps to per-phase accumulators; this prints them per tile, for the detector's chains at
8 x 720 x 1280 (chain 0 = the backbone's C = 256 stage), detector alone and beside a loaded second stream."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _synth import synth_yolo_state_dict, test_image  # noqa: E402
from adaptiveisp_amd.yolo import YoloEngine, _lib, yolov3  # noqa: E402

L = _lib.load()
L.adayolo_debug_chain_stamps.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
det = yolov3()
det.load_state_dict(synth_yolo_state_dict(det, seed=2))
eng = YoloEngine(det.eval(), 8, 720, 1280, device=dev)
eng.autotune(cache=os.path.join(ROOT, "adaptiveisp_amd", "yolo", "tuning", "mi355x.json"), write=False)
x = torch.from_numpy(test_image(8, 720, 1280, seed=3, special=False)).to(dev)
NAMES = {0: "-> tile start", 1: "arguments, row decode, W/A DMA issue (0->1)", 2: "prologue landed + publish of the previous tile (1->2)",
         3: "k-loop (2->3)", 6: "tail + epilogue (3->6)", 4: "fused: wait + barrier (6->4)", 5: "fused: 1x1 GEMM (4->5)",
         8: "fused: out2 epilogue / last stores issued (5|6->8)", 11: "look-ahead stage 3 (8->11)", 9: "end barrier: waiting for the slowest wave (11->9)",
         12: "loop top, slow path taken: publish, poll, acquire (9->12)", 10: "loop top: hand-over through LDS, layer arguments (9|12->10)"}
st = _lib.stream_ptr()
buf = np.zeros(16, np.uint64)
for ci, c in enumerate(eng.chains):
    kind, fn, args = next(p for p in eng.plan if p[0] == "chain" and p[2][2].value == c["ws"].data_ptr())
    for _ in range(3):
        fn(*args, st)
    torch.cuda.synchronize()
    assert L.adayolo_debug_chain_stamps(buf.ctypes.data) == 0
    reps = 10
    for _ in range(reps):
        fn(*args, st)
    torch.cuda.synchronize()
    assert L.adayolo_debug_chain_stamps(buf.ctypes.data) == 0
    tiles = float(buf[14])
    print(f"chain {ci}: {c['layers']} layers, {tiles / reps:.0f} tiles per launch; cycles per tile (thread 0 of each workgroup):")
    tot = 0.0
    print(f"   slow path taken for {float(buf[13]) / max(tiles, 1) * 100:.1f} % of the tiles")
    for k in (12, 10, 0, 1, 2, 3, 6, 4, 5, 8, 11, 9):
        v = float(buf[k]) / max(tiles, 1)
        tot += v
        print(f"   {NAMES[k]:72s} {v:9.0f}")
    print(f"   {'sum':72s} {tot:9.0f}")
