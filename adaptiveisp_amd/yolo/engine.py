"""HIP/MFMA execution engine for the YOLOv3 reward-model forward.

Built once per (model, batch, resolution): folds every BatchNorm into its conv, lays the weights out as
bf16 [Cout][KH][KW][Cin], allocates every activation as an NHWC bf16 tensor and records a flat launch
plan of C-ABI calls (include/adayolo.h). What the reference does with ~250 ATen launches per forward
(conv, BN, SiLU, add, upsample, cat, permute, sigmoid...) becomes ~80 launches, each conv carrying its
bias + SiLU + residual epilogue:

  * stem: reads the ISP output directly (planar fp32), letterboxes to a multiple of 32 rows and applies
    Conv(3->32)+SiLU in one kernel — no separate pad / layout / dtype pass;
  * Concat is free: the two producers write straight into channel slices of the concatenated tensor
    (the convs take channel strides), Upsample is a scatter into its slice;
  * Detect: 1x1 conv to 255(+1 pad) channels, then one decode kernel writes the [B, N, 85] fp32 prediction.

There is no eager fallback: without libadayolo.so / a HIP device this raises.
"""
import ctypes
import os

import torch

from . import _lib
from .model import Bottleneck, Concat, Conv, Detect, DetectionModel

LETTERBOX_VALUE = 114.0 / 255.0     # yolov3/utils/augmentations.py:111 (color=(114,114,114)) on a [0,1] image


class _View:
    """A channel slice [coff, coff+C) of an NHWC bf16 buffer [B,H,W,CS]."""

    def __init__(self, buf, coff, C):
        self.buf, self.coff, self.C = buf, coff, C
        self.H, self.W, self.cs = buf.shape[1], buf.shape[2], buf.shape[3]

    @property
    def ptr(self):
        return self.buf.data_ptr() + 2 * self.coff

    def tensor(self):
        return self.buf[..., self.coff:self.coff + self.C]


def _pack_conv(w, b, pad_cout_to=None):
    """[Cout,Cin,k,k] fp32 -> bf16 [Cout,k,k,Cin] contiguous (+ fp32 bias); optional zero rows up to pad_cout_to."""
    w = w.detach().float().permute(0, 2, 3, 1).contiguous()
    b = b.detach().float().contiguous()
    if pad_cout_to is not None and w.shape[0] < pad_cout_to:
        extra = pad_cout_to - w.shape[0]
        w = torch.cat([w, w.new_zeros((extra,) + tuple(w.shape[1:]))], 0)
        b = torch.cat([b, b.new_zeros(extra)], 0)
    return w.to(torch.bfloat16).contiguous(), b


BNECK_WS_DEFAULT = "1"       # whole-Bottleneck launches for the C = 64 / 128 stages (YoloEngine.fuse_bottlenecks_ws)


class YoloEngine:
    def __init__(self, model: DetectionModel, batch, height, width, device="cuda:0", head_chunks=None):
        """`head_chunks`: run the stem and the first down-sampling conv depth-first over this many batch chunks, so the
        stem's output (the largest tensor of the network: 482 MB for 8 x 736 x 1280 x 32 bf16) is consumed while it is
        still in the 256 MB Infinity Cache. Measured: -0.1 ms per forward with 2 chunks when launched eagerly, no gain
        under hipGraph replay (4.09 vs 4.12 ms) — default 1 (off); ADAYOLO_HEAD_CHUNKS overrides."""
        if width % 32:
            raise ValueError(f"image width {width} must be a multiple of 32 (reference: check_img_size)")
        self.L = _lib.load()
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise _lib.AdayoloError("YoloEngine needs a HIP device; there is no CPU path")
        self.B, self.H, self.W = int(batch), int(height), int(width)
        self.Hp = (self.H + 31) // 32 * 32
        self.pad_top = (self.Hp - self.H) // 2
        det = model.model[-1]
        self.na, self.no, self.nc = det.na, det.no, det.nc
        self._keep = []          # tensors referenced by raw pointers in the plan
        self.plan = []
        self.fused_pairs = 0
        self._build(model)
        if head_chunks is None and os.environ.get("ADAYOLO_HEAD_CHUNKS"):
            head_chunks = int(os.environ["ADAYOLO_HEAD_CHUNKS"])
        if head_chunks is None:
            head_chunks = 1
        # fused stem + first down-sampling conv (yolo_stem_down.hip): available when layer 1 is Conv(32->64, k3, s2) + SiLU
        self._head_down, self._head_next, self.fuse_head = None, None, False
        if len(self.ops) > 1 and self.ops[0]["kind"] == "stem" and self.ops[1]["kind"] == "conv":
            o = self.ops[1]
            if (o["k"], o["s"], o["act"], o["cout"], o["src"].C, o["res"]) == (3, 2, _lib.ACT_SILU, 64, 32, None) and \
                    o["src"] is self._stem[2]:
                self._head_down = o
                self.fuse_head = os.environ.get("ADAYOLO_FUSE_HEAD", "1") == "1"
                n = self.ops[2] if len(self.ops) > 2 else None       # Bottleneck.cv1 of layer 2: 1x1, 64 -> 32
                if n is not None and n["kind"] == "conv" and n["src"] is o["dst"] and n["res"] is None and \
                        (n["k"], n["s"], n["act"], n["cout"], n["src"].C) == (1, 1, _lib.ACT_SILU, 32, 64) and \
                        self.plan[2][0] == "conv" and os.environ.get("ADAYOLO_FUSE_HEAD_NEXT", "1") == "1" and \
                        os.environ.get("ADAYOLO_BNECK_WS", BNECK_WS_DEFAULT) != "1":
                    # (with the whole-Bottleneck kernel for the shallow stages the block's own launch computes cv1: k_stem_down
                    # neither computes nor writes the hidden tensor)
                    self._head_next = n
        if self.B % head_chunks:
            raise ValueError(f"head_chunks={head_chunks} does not divide the batch {self.B}")
        self.head_chunks = int(head_chunks)

    # ------------------------------------------------------------------------------------------
    def _new(self, H, W, C):
        """An NHWC bf16 buffer [B,H,W,C]. With `self._shared = (parent, first)` — an engine built over the same model with a
        larger batch — the buffer IS images [first, first + B) of the parent's buffer of the same allocation index (both
        constructors walk the model the same way): what the parent's forward leaves there (activations, kept
        pre-activations) this engine's launches read (YoloTrainPairEngine)."""
        allocs = self.__dict__.setdefault("_allocs", [])
        shared = getattr(self, "_shared", None)
        if shared is not None:
            parent, first = shared
            t = parent._allocs[len(allocs)][first:first + self.B]
            if tuple(t.shape) != (self.B, H, W, C):
                raise _lib.AdayoloError("shared buffers: the two engines do not allocate alike")
        else:
            t = torch.empty((self.B, H, W, C), dtype=torch.bfloat16, device=self.dev)
        allocs.append(t)
        self._keep.append(t)
        return t

    def _conv_op(self, src, conv_w, conv_b, dst, k, s, act, res=None, cout=None):
        w, b = conv_w.to(self.dev), conv_b.to(self.dev)
        self._keep += [w, b]
        cout = cout if cout is not None else w.shape[0]
        args = [ctypes.c_void_p(src.ptr), src.cs, ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(b.data_ptr()),
                ctypes.c_void_p(res.ptr) if res is not None else None, res.cs if res is not None else 0,
                ctypes.c_void_p(dst.ptr), dst.cs, self.B, src.H, src.W, src.C, cout, k, s, act, 0]   # last: variant
        self.plan.append(("conv", self._conv_launch, args))
        self.ops.append(dict(kind="conv", src=src, dst=dst, res=res, w=w, b=b, k=k, s=s, act=act, cout=cout))
        flops = 2.0 * self.B * dst.H * dst.W * cout * k * k * src.C
        self.flops += flops

    @staticmethod
    def _fused_stem_ok(m):
        c = m.conv
        return (c.in_channels, c.out_channels, c.kernel_size[0], c.stride[0]) == (3, 32, 3, 1)

    def _build(self, model):
        self.flops = 0.0
        self.ops = []            # the same launches as `plan`, with their views (used by the training engine)
        layers = list(model.model)
        # pass 1: shapes (C,H,W) of every layer output
        shp = []
        for i, m in enumerate(layers):
            prev = (3, self.Hp, self.W) if i == 0 else (shp[_src(i, m.f)] if isinstance(m.f, int) else None)
            if isinstance(m, Conv):
                s = m.conv.stride[0]
                shp.append((m.conv.out_channels, (prev[1] - 1) // s + 1, (prev[2] - 1) // s + 1))
            elif isinstance(m, (Bottleneck, torch.nn.Sequential)):
                last = m if isinstance(m, Bottleneck) else m[-1]
                shp.append((last.cv2.conv.out_channels, prev[1], prev[2]))
            elif isinstance(m, torch.nn.Upsample):
                shp.append((prev[0], prev[1] * 2, prev[2] * 2))
            elif isinstance(m, Concat):
                srcs = [shp[_src(i, j)] for j in m.f]
                shp.append((sum(s[0] for s in srcs), srcs[0][1], srcs[0][2]))
            elif isinstance(m, Detect):
                shp.append(None)
        # pass 2: output views; concat sources write into slices of the concat tensor
        view = [None] * len(layers)
        for i, m in enumerate(layers):
            if isinstance(m, Concat):
                C, H, W = shp[i]
                buf = self._new(H, W, C)
                view[i] = _View(buf, 0, C)
                off = 0
                for j in m.f:
                    src = _src(i, j)
                    view[src] = _View(buf, off, shp[src][0])
                    off += shp[src][0]
        for i, m in enumerate(layers):
            if view[i] is None and shp[i] is not None:
                C, H, W = shp[i]
                view[i] = _View(self._new(H, W, C), 0, C)
        # pass 3: launch plan
        for i, m in enumerate(layers):
            src = view[_src(i, m.f)] if (i > 0 and isinstance(m.f, int)) else None
            if i == 0 and not self._fused_stem_ok(m):
                # width-scaled detector (first conv is not 3->32 k3 s1): letterbox + NHWC pack, then the generic conv
                # on weights zero-padded to 8 input channels
                packed = _View(self._new(self.Hp, self.W, 8), 0, 8)
                args = (None, ctypes.c_void_p(packed.ptr), packed.cs, self.B, self.H, self.W, self.Hp, self.pad_top,
                        LETTERBOX_VALUE)
                self._stem = None
                self._pack_dst = packed
                self.plan.append(("pack", self.L.adayolo_letterbox_pack, args))
                self.ops.append(dict(kind="pack", dst=packed))
                w, b = m.folded()
                w = torch.nn.functional.pad(w.detach().float(), (0, 0, 0, 0, 0, 8 - w.shape[1]))
                w, b = _pack_conv(w, b)
                self._conv_op(packed, w, b, view[0], m.conv.kernel_size[0], m.conv.stride[0], _lib.ACT_SILU)
            elif i == 0:
                w, b = m.folded()
                w = w.detach().float().permute(0, 2, 3, 1).contiguous().to(self.dev)      # [32][3][3][3] fp32
                b = b.detach().float().contiguous().to(self.dev)
                self._keep += [w, b]
                self._stem = (w, b, view[0])
                self.plan.append(("stem", None, None))
                self.ops.append(dict(kind="stem", dst=view[0], w=w, b=b))
                self.flops += 2.0 * self.B * self.Hp * self.W * 32 * 27
            elif isinstance(m, Conv):
                w, b = _pack_conv(*m.folded())
                self._conv_op(src, w, b, view[i], m.conv.kernel_size[0], m.conv.stride[0], _lib.ACT_SILU)
            elif isinstance(m, (Bottleneck, torch.nn.Sequential)):
                blocks = [m] if isinstance(m, Bottleneck) else list(m)
                cur = src
                for bi, blk in enumerate(blocks):
                    # the hidden width is c2/2: 4 in the narrowest width-scaled checkpoints. Pad it to the kernels' 8-channel
                    # granule with zero filters (SiLU(0) = 0 feeds zero weights: the result is unchanged)
                    ch = (blk.cv1.conv.out_channels + 7) // 8 * 8
                    hidden = _View(self._new(cur.H, cur.W, ch), 0, ch)
                    out = view[i] if bi == len(blocks) - 1 else \
                        _View(self._new(cur.H, cur.W, blk.cv2.conv.out_channels), 0, blk.cv2.conv.out_channels)
                    w1, b1 = _pack_conv(*blk.cv1.folded(), pad_cout_to=ch)
                    wf2, bf2 = blk.cv2.folded()
                    w2, b2 = _pack_conv(torch.nn.functional.pad(wf2.detach().float(), (0, 0, 0, 0, 0, ch - wf2.shape[1])), bf2)
                    self._conv_op(cur, w1, b1, hidden, 1, 1, _lib.ACT_SILU)
                    self._conv_op(hidden, w2, b2, out, 3, 1, _lib.ACT_SILU, res=cur if blk.add else None)
                    cur = out
            elif isinstance(m, torch.nn.Upsample):
                args = (ctypes.c_void_p(src.ptr), src.cs, ctypes.c_void_p(view[i].ptr), view[i].cs, self.B, src.H,
                        src.W, src.C)
                self.plan.append(("up", self.L.adayolo_upsample2x, args))
                self.ops.append(dict(kind="up", src=src, dst=view[i]))
            elif isinstance(m, Detect):
                ins = [view[j] for j in m.f]
                self.rows = sum(self.na * v.H * v.W for v in ins)
                self.pred = torch.empty((self.B, self.rows, self.no), dtype=torch.float32, device=self.dev)
                self.raw = []
                row = 0
                cpad = (self.na * self.no + 7) // 8 * 8
                for li, v in enumerate(ins):
                    raw = _View(self._new(v.H, v.W, cpad), 0, cpad)
                    w, b = _pack_conv(m.m[li].weight, m.m[li].bias, pad_cout_to=cpad)
                    self._conv_op(v, w, b, raw, 1, 1, _lib.ACT_NONE, cout=cpad)
                    anc = (m.anchors[li].float() * float(m.stride[li])).contiguous().to(self.dev)
                    self._keep.append(anc)
                    args = (ctypes.c_void_p(raw.ptr), raw.cs, ctypes.c_void_p(self.pred.data_ptr()), self.rows, row,
                            ctypes.c_void_p(anc.data_ptr()), float(m.stride[li]), self.B, v.H, v.W, self.na, self.no)
                    self.plan.append(("decode", self.L.adayolo_detect_decode, args))
                    self.raw.append(raw)
                    row += self.na * v.H * v.W
        self.views = view

    # ------------------------------------------------------------------------------------------
    SPLITK_BASE = 100                                    # include/adayolo.h: variant 100 + S = variant 60 with S ranges of k-tiles
    SPLITK_CANDIDATES = (102, 103, 104, 106, 108, 112, 116)
    TUNE_CANDIDATES = (2, 5, 22, 26, 27, 40, 50, 60, 80, 85, 90) + SPLITK_CANDIDATES

    def _plans(self):
        return [self.plan]

    # ---- split-K launches take a workspace (fp32 partial tiles + tickets): one per engine, sized for every conv of its
    #      plans and every split the library serves for it; zeroed once, the kernels leave the tickets zero
    def _splitk_bytes(self, args, variant):
        return int(self.L.adayolo_conv_splitk_workspace_bytes(*args[8:15], variant))

    def _splitk_workspace(self):
        ws = getattr(self, "_splitk_ws", None)
        if ws is None:
            need = 0
            for plan in self._plans():
                for kind, _, args in plan:
                    if kind == "conv":
                        need = max([need] + [self._splitk_bytes(args, v) for v in self.SPLITK_CANDIDATES])
            if torch.cuda.is_current_stream_capturing():
                raise _lib.AdayoloError("split-K workspace requested inside a graph capture: run the engine once before capturing")
            t = torch.zeros(max(need, 16), dtype=torch.uint8, device=self.dev)
            ws = self._splitk_ws = (t, ctypes.c_void_p(t.data_ptr()), ctypes.c_size_t(t.numel()))
        return ws

    def _tune_penalty(self, key, variant):
        """ms added to a candidate's measured time (what choosing it costs elsewhere; the training engine's forward)."""
        return 0.0

    S2GRAD_VARIANTS = (5, 22, 26, 27, 60) + SPLITK_CANDIDATES     # kernels that serve adayolo_conv_s2grad_fwd (listed with k = 2)

    def _conv_launch(self, *a):
        """adayolo_conv_fwd_variant's argument list (17 + stream); the split-K variants go to their own entry point, and
        k = 2 is the training engine's stride-2 data gradient (adayolo_conv_s2grad_fwd: H x W its Ho x Wo grid, Cin the
        layer's output channels, Cout 4 x its input channels)."""
        if a[13] == 2:
            ptr, nbytes = (self._splitk_workspace()[1:]) if a[16] >= self.SPLITK_BASE else (None, 0)
            return self.L.adayolo_conv_s2grad_fwd(*a[:8], None, 0, None, 0, a[8], a[9], a[10], a[11], a[12] // 4, a[16], ptr, nbytes, a[17])
        if a[16] >= self.SPLITK_BASE:
            _, ptr, nbytes = self._splitk_workspace()
            return self.L.adayolo_conv_splitk_fwd(*a[:8], None, 0, *a[8:17], ptr, nbytes, a[17])
        return self.L.adayolo_conv_fwd_variant(*a)

    def autotune(self, reps=5, cache=None, retune=False, write=True):
        """Pick the fastest conv kernel variant per layer by timing it on this engine's own buffers (all variants
        compute the same result; see include/adayolo.h). Like a vendor library's 'find' step. With `cache` (a JSON
        path) the choice of every layer shape the table holds is loaded, the others are measured and written back
        (atomically; `write=False` for ranks other than 0 of a multi-process job); `retune` measures all of them."""
        import json
        import os
        st = _lib.stream_ptr()
        chosen = {}
        entries = [e for plan in self._plans() for e in plan]
        keys = {tuple(args[8:16]) for kind, _, args in entries if kind == "conv"}
        if cache and os.path.exists(cache) and not retune:
            try:
                table = {tuple(int(x) for x in k.split(",")): int(v) for k, v in json.load(open(cache)).items()}
            except Exception:
                table = {}
            # a table written by an older build may name variants this library no longer has: treat them as missing
            table = {k: v for k, v in table.items() if v in self.TUNE_CANDIDATES}
            chosen = {k: table[k] for k in keys if k in table}       # only the layer shapes the table lacks are measured
            if len(chosen) == len(keys):
                for kind, fn, args in entries:
                    if kind == "conv":
                        args[16] = chosen[tuple(args[8:16])]
                self.tuned = chosen
                self.fuse_pairs()
                return self.tuned
        with torch.cuda.device(self.dev):
            for kind, fn, args in entries:
                if kind != "conv":
                    continue
                key = tuple(args[8:16])                      # B,H,W,Cin,Cout,k,s,act
                if key not in chosen:
                    best = (None, float("inf"))
                    for v in self.TUNE_CANDIDATES:
                        if 40 <= v < 50 and not (args[13] == 3 and args[11] in (32, 64)):
                            continue                             # whole-K-resident kernels: 3x3 with Cin 32 / 64
                        if 50 <= v < 60 and (args[11] % 64 or args[12] % 256):
                            continue                             # ping-pong kernel: Cin % 64 == 0, Cout % 256 == 0
                        if 60 <= v < 80 and (args[11] % 64 or args[12] % 128):
                            continue                             # 256x128 ping-pong kernel: Cin % 64 == 0, Cout % 128 == 0
                        if 80 <= v < 90 and (args[11] % 32 or args[12] % 128):
                            continue                             # 256x128, two workgroups per CU: Cin % 32 == 0, Cout % 128 == 0
                        if 90 <= v < 100 and not (args[13] == 3 and args[14] == 1 and args[11] in (32, 64) and args[12] % 64 == 0 and
                                                  args[15] == _lib.ACT_SILU):
                            continue                             # weights-in-registers kernel: 3x3 s1, Cin 32 / 64
                        if v >= self.SPLITK_BASE and self._splitk_bytes(args, v) == 0:
                            continue                             # this split does not serve the shape
                        if args[13] == 2 and v not in self.S2GRAD_VARIANTS:
                            continue                             # stride-2 data gradient: the kernels with that epilogue
                        args[16] = v
                        fn(*args, st)
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        t = float("inf")
                        for _ in range(reps):                     # best of `reps` single launches
                            e0.record()
                            fn(*args, st)
                            e1.record()
                            e1.synchronize()
                            t = min(t, e0.elapsed_time(e1))
                        t += self._tune_penalty(key, v)
                        if t < best[1]:
                            best = (v, t)
                    chosen[key] = best[0]
                args[16] = chosen[key]
        self.tuned = chosen
        self.fuse_pairs()
        if cache and write:
            try:
                try:
                    old = json.load(open(cache)) if os.path.exists(cache) else {}
                except ValueError:
                    old = {}
                old.update({",".join(str(int(x)) for x in k): int(v) for k, v in chosen.items()})
                os.makedirs(os.path.dirname(cache), exist_ok=True)
                tmp = f"{cache}.{os.getpid()}.tmp"                # whole-file replace: a concurrent reader never sees a
                with open(tmp, "w") as f:                         # partial table (one process per GPU under torchrun)
                    json.dump(old, f, indent=0, sort_keys=True)
                os.replace(tmp, cache)
            except OSError:
                pass                                            # read-only checkout: keep the in-memory choice
        return chosen

    # ------------------------------------------------------------------------------------------
    _pair_fusion = True                                  # (the eval plan only: the training engine's tfwd / tbwd lists are separate)

    def fuse_pairs(self):
        """Bottleneck.cv2 of one block + Bottleneck.cv1 of the next in ONE launch (adayolo_conv_fused1x1_fwd) wherever the
        first conv runs on the 256 x 256 kernel (variant 50) with all its 256 output channels in one tile and the second is
        the 1x1 256 -> 128 + SiLU that reads exactly that output: the 1x1 layers of the C = 256 stage are HBM-bound on their
        own. Runs after the variants are known (autotune); ADAYOLO_FUSE_1X1=0 keeps the layers separate. Returns the
        number of fused pairs."""
        self.fused_pairs = getattr(self, "fused_pairs", 0)
        if self._pair_fusion:
            self.fuse_bottlenecks_ws()
            self.fuse_bottlenecks()
        if not self._pair_fusion or os.environ.get("ADAYOLO_FUSE_1X1", "1") != "1":
            return 0
        out, i, n, P = [], 0, 0, self.plan
        first_free = 3 if self._head_next is not None else 2      # the head's launches are addressed by plan index
        while i < len(P):
            kind, fn, a = P[i]
            if i >= first_free and kind == "conv" and i + 1 < len(P) and P[i + 1][0] == "conv":
                b = P[i + 1][2]
                Ho, Wo = (a[9] - 1) // a[14] + 1, (a[10] - 1) // a[14] + 1
                if (a[12] == 256 and a[16] == 50 and a[11] % 64 == 0 and b[11] == 256 and b[12] == 128 and b[13] == 1 and
                        b[14] == 1 and b[15] == _lib.ACT_SILU and b[4] is None and b[0].value == a[6].value and
                        b[1] == a[7] and (b[8], b[9], b[10]) == (a[8], Ho, Wo)):
                    # the second layer's weights in the fragment-major order the fused epilogue loads (include/adayolo.h)
                    w2 = next(o["w"] for o in self.ops if o["kind"] == "conv" and o["w"].data_ptr() == b[2].value)
                    w2p = w2.reshape(4, 32, 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous()
                    self._keep.append(w2p)
                    out.append(("conv2", self.L.adayolo_conv_fused1x1_fwd,
                                list(a[:16]) + [ctypes.c_void_p(w2p.data_ptr()), b[3], b[6], b[7], 128]))
                    i, n = i + 2, n + 1
                    continue
            out.append(P[i])
            i += 1
        self.plan = out
        self.fused_pairs += n
        self.fuse_chains()
        self.fuse_k1()
        return n

    def fuse_k1(self):
        """1x1 layers with Cin in {256, 512} and Cout % 256 == 0 that are still launches of their own (Bottleneck.cv1 of the
        C = 512 stage and the head's 1x1 convs: yolov3/models/common.py:45-59,110-120) on the whole-K kernel
        (adayolo_conv1x1_stream_fwd, csrc/yolo_conv_k1.hip) instead of a ring kernel. OFF by default (ADAYOLO_K1=1 turns it on):
        measured in round 6 (profiles/round6_k1_ab.txt) the whole-K kernel is no faster than the 256 x 128 ring kernel on these
        layers (512 -> 256 @46x80: 16.2-17.4 vs 13.5-16.0 us back to back, detector 3.356 vs 3.339 ms) — a CU's operand ingest
        (128 KB of activations + 256 KB of weights per 128 px x 256 ch of output) bounds both, not the k-loop's structure.
        Returns the number of launches moved."""
        self.k1_layers = getattr(self, "k1_layers", 0)
        if os.environ.get("ADAYOLO_K1", "0") != "1" or not self._pair_fusion:
            return 0
        first_free = 3 if self._head_next is not None else 2
        n = 0
        for i, (kind, fn, a) in enumerate(self.plan):
            if i < first_free or kind != "conv":
                continue
            if not (a[13] == 1 and a[14] == 1 and a[4] is None and a[11] in (256, 512) and a[12] % 256 == 0):
                continue
            w = next(o["w"] for o in self.ops if o["kind"] == "conv" and o["w"].data_ptr() == a[2].value)
            cout, cin = a[12], a[11]
            if w.shape[0] != cout or w.numel() != cout * cin:
                continue
            wp = w.reshape(cout // 32, 32, cin // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous()   # fragment-major (include/adayolo.h)
            self._keep.append(wp)
            self.plan[i] = ("k1", self.L.adayolo_conv1x1_stream_fwd,
                            [a[0], a[1], ctypes.c_void_p(wp.data_ptr()), a[3], a[6], a[7], a[8], a[9], a[10], cin, cout, a[15]])
            n += 1
        self.k1_layers += n
        return n

    CHAIN_MIN = 4                                        # launches a run must replace to become a chain

    def fuse_chains(self):
        """Runs of consecutive launches of the 256 x 256 kernel (variant 50, alone or with the next block's 1x1 fused) and the
        256 x 128 kernel (variant 60) — the backbone from its C = 256 stage through the C = 512 stage, the head's blocks — as ONE
        persistent launch each (adayolo_conv_chain_fwd,
        csrc/yolo_conv_pp.hip: k_conv_chain): tiles of all the run's layers drawn from one work counter, a tile waiting only for
        the producer tiles its input window / residual rows lie in. Bit-identical to the separate launches (same tile code).
        ADAYOLO_CHAIN=0 keeps the launches separate. Returns the number of chains."""
        # chains of an earlier plan that are still IN the plan stay registered; the others are dropped (a re-plan must not leave
        # workspaces behind that no launch uses)
        live = {a[2].value for kind, _, a in self.plan if kind == "chain"}
        self.chains = [c for c in getattr(self, "chains", []) if c["ws"].data_ptr() in live]
        if os.environ.get("ADAYOLO_CHAIN", "1") != "1":
            return 0
        P, out, i, made = self.plan, [], 0, 0
        first_free = 3 if self._head_next is not None else 2

        def eligible(j):
            kind, _, a = P[j]
            if j < first_free:
                return False
            if kind == "conv2":
                return True
            if kind != "conv" or a[11] % 64:
                return False
            return (a[16] == 50 and a[12] % 256 == 0) or (a[16] == 60 and a[12] % 128 == 0)

        # What a chain can win is the partly empty LAST round of every layer (460 tiles on 256 CUs: the next layer's tiles fill it)
        # and the launch boundaries; a layer with FEWER tiles than CUs has no second round to fill, and its successor cannot start
        # before (nearly) all of it is done — measured (round 5, profiles/round5_chain_ab.txt): the C = 512 stage's 16 layers of 230
        # tiles inside a chain run exactly as fast as their 16 launches, the head's short runs slower. So: a run is made of
        # layers with more tiles than CUs, plus at most ONE trailing layer below that (it fills the run's own tail).
        cus = torch.cuda.get_device_properties(self.dev).multi_processor_count

        def tiles(j):
            kind, _, a = P[j]
            Ho, Wo = (a[9] - 1) // a[14] + 1, (a[10] - 1) // a[14] + 1
            bn = 128 if (kind == "conv" and a[16] == 60) else 256
            return ((a[8] * Ho * Wo + 255) // 256) * (a[12] // bn)

        if os.environ.get("ADAYOLO_CHAIN_ALL", "0") == "1":      # measurement: every eligible run, whatever its tile counts
            cus = 0
        runs = dict(chain_runs([(eligible(j), tiles(j) if eligible(j) else 0) for j in range(len(P))], cus, self.CHAIN_MIN))
        while i < len(P):
            j = runs.get(i, i)
            if j > i:
                n = j - i
                layers = (_lib.ChainLayer * n)()
                flops = 0.0
                for k in range(n):
                    kind, _, a = P[i + k]
                    ly = layers[k]
                    ly.in_, ly.in_cstride, ly.weight, ly.bias = a[0], a[1], a[2], a[3]
                    ly.residual, ly.res_cstride, ly.out, ly.out_cstride = a[4], a[5], a[6], a[7]
                    ly.B, ly.H, ly.W, ly.Cin, ly.Cout, ly.ksize, ly.stride, ly.act = a[8:16]
                    ly.tile = 1 if (kind == "conv" and a[16] == 60) else 0
                    Ho, Wo = (a[9] - 1) // a[14] + 1, (a[10] - 1) // a[14] + 1
                    flops += 2.0 * a[8] * Ho * Wo * a[12] * a[13] * a[13] * a[11]
                    if kind == "conv2":
                        ly.weight2, ly.bias2, ly.out2, ly.out2_cstride, ly.Cout2 = a[16], a[17], a[18], a[19], a[20]
                        flops += 2.0 * a[8] * Ho * Wo * a[12] * a[20]
                nbytes = int(self.L.adayolo_conv_chain_workspace_bytes(layers, n))
                if nbytes:
                    ws = torch.empty((nbytes,), dtype=torch.uint8, device=self.dev)
                    _lib.check(self.L.adayolo_conv_chain_prepare(layers, n, ctypes.c_void_p(ws.data_ptr()), nbytes), "adayolo_conv_chain_prepare")
                    self._keep += [ws, layers]
                    self.chains.append(dict(layers=n, flops=flops, ws=ws, first=i, entries=P[i:j]))
                    out.append(("chain", self.L.adayolo_conv_chain_fwd, [layers, n, ctypes.c_void_p(ws.data_ptr()), nbytes]))
                    made += 1
                    i = j
                    continue
                # the library does not serve this run as a chain (a dependency window wider than 32 m-tiles, an output >= 2 GB,
                # aliased tensors): the launches stay separate — said once, not silently
                import warnings
                warnings.warn(f"YoloEngine.fuse_chains: the run of {n} launches at plan index {i} stays unchained "
                              f"(adayolo_conv_chain_workspace_bytes == 0: shape not served)", RuntimeWarning, stacklevel=2)
            out.extend(P[i:max(j, i + 1)])
            i = max(j, i + 1)
        self.plan = out
        return made

    def chain_status(self):
        """0 when every dependency wait of every chain's forwards saw its counters arrive. BLOCKS (device synchronise + copy):
        tests and end-of-run checks; the forward path uses chain_poll."""
        return max([int(self.L.adayolo_conv_chain_status(ctypes.c_void_p(c["ws"].data_ptr()))) for c in getattr(self, "chains", [])] + [0])

    def chain_poll(self):
        """The same WITHOUT touching the device: the pinned host word a launch's last workgroup mirrors a give-up code to
        (adayolo_conv_chain_poll). 0 = nothing recorded so far (a forward still in flight has not reported yet)."""
        return max([int(self.L.adayolo_conv_chain_poll(ctypes.c_void_p(c["ws"].data_ptr()))) for c in getattr(self, "chains", [])] + [0])

    def check_chains(self, sync=False):
        """Raise if a dependency wait of a persistent chain gave up (the forward that contained it ran on incomplete inputs:
        its detections are WRONG although every launch returned ADAYOLO_OK). `sync=False`: the host-word poll — called at the top
        of every forward, so a bad forward is reported by the next one at the latest; `sync=True`: the blocking form, for the
        places that synchronise anyway (end of an eval batch, end of bench.py's timed region)."""
        code = self.chain_status() if sync else self.chain_poll()
        if code:
            raise _lib.AdayoloError(f"persistent conv chain: the dependency wait of work item {code - 1} gave up after 1 s — the "
                                    f"forward that contained it ran on incomplete inputs (ADAYOLO_CHAIN=0 runs the layers as "
                                    f"separate launches)")

    def fuse_bottlenecks_ws(self):
        """A whole Bottleneck of the shallow stages — cv1 (1x1 C -> C/2 + SiLU) and cv2 (3x3 C/2 -> C + SiLU, + the block's
        input), C = 64 on 368 x 640 maps and C = 128 on 184 x 320 at the benchmark's size (yolov3/models/common.py:110-120,
        yolov3.yaml:13-27) — as ONE launch with the hidden tensor in LDS (adayolo_bottleneck_ws_fwd, csrc/yolo_bneck_ws.hip)
        wherever the plan holds exactly that pair. ADAYOLO_BNECK_WS=0 keeps the two launches. Returns the number of fused blocks."""
        self.fused_ws_blocks = getattr(self, "fused_ws_blocks", 0)
        if os.environ.get("ADAYOLO_BNECK_WS", BNECK_WS_DEFAULT) != "1":
            return 0
        out, i, n, P = [], 0, 0, self.plan
        first_free = 3 if self._head_next is not None else 2
        while i < len(P):
            kind, fn, a = P[i]
            if i >= first_free and kind == "conv" and i + 1 < len(P) and P[i + 1][0] == "conv":
                b = P[i + 1][2]
                C = a[11]
                if (C in (64, 128) and (a[12], a[13], a[14], a[15]) == (C // 2, 1, 1, _lib.ACT_SILU) and a[4] is None and
                        (b[11], b[12], b[13], b[14], b[15]) == (C // 2, C, 3, 1, _lib.ACT_SILU) and b[4] is not None and
                        b[4].value == a[0].value and b[5] == a[1] and b[0].value == a[6].value and b[1] == a[7] and
                        (b[8], b[9], b[10]) == (a[8], a[9], a[10]) and b[6].value != a[0].value and
                        2 * a[8] * a[9] * a[10] * max(a[1], b[7]) + 256 <= 0xFFFFFF00):
                    out.append(("bneckws", self.L.adayolo_bottleneck_ws_fwd,
                                [a[0], a[1], a[2], a[3], b[2], b[3], b[6], b[7], a[8], a[9], a[10], C]))
                    i, n = i + 2, n + 1
                    continue
            out.append(P[i])
            i += 1
        self.plan = out
        self.fused_ws_blocks += n
        return n

    def fuse_bottlenecks(self):
        """A whole Bottleneck of the C = 256 stage — cv1 (1x1 256 -> 128 + SiLU) and cv2 (3x3 128 -> 256 + SiLU, + the block's
        input) — as ONE launch with the hidden tensor in LDS (adayolo_bottleneck256_fwd, csrc/yolo_bneck.hip), wherever the plan
        holds exactly that pair. Off by default (ADAYOLO_BNECK=1 turns it on): measured equal to the [3x3 | next 1x1] pairs
        it replaces. Returns the number of fused blocks."""
        self.fused_blocks = getattr(self, "fused_blocks", 0)
        if os.environ.get("ADAYOLO_BNECK", "0") != "1":
            return 0
        out, i, n, P = [], 0, 0, self.plan
        first_free = 3 if self._head_next is not None else 2
        while i < len(P):
            kind, fn, a = P[i]
            if i >= first_free and kind == "conv" and i + 1 < len(P) and P[i + 1][0] == "conv":
                b = P[i + 1][2]
                if ((a[11], a[12], a[13], a[14], a[15]) == (256, 128, 1, 1, _lib.ACT_SILU) and a[4] is None and
                        (b[11], b[12], b[13], b[14], b[15]) == (128, 256, 3, 1, _lib.ACT_SILU) and b[4] is not None and
                        b[4].value == a[0].value and b[5] == a[1] and b[0].value == a[6].value and b[1] == a[7] and a[7] == 128 and
                        (b[8], b[9], b[10]) == (a[8], a[9], a[10]) and b[6].value != a[0].value):
                    out.append(("bneck", self.L.adayolo_bottleneck256_fwd,
                                [a[0], a[1], a[2], a[3], b[2], b[3], b[6], b[7], a[8], a[9], a[10]]))
                    i, n = i + 2, n + 1
                    continue
            out.append(P[i])
            i += 1
        self.plan = out
        self.fused_blocks += n
        return n

    # ONE set of activation buffers and ONE split-K workspace (tickets + fp32 partial tiles) serve every pass of an engine —
    # the inference plan, the training forward, the backward and their hipGraphs — so two passes must never overlap. A pass that
    # starts on another stream than the previous one therefore waits for it (one event per pass); inside a graph capture the
    # capture's own dependencies do that. (rl.py runs the input batch's forward on a side stream and the retouched batch's on
    # the current one: its explicit wait_stream is now belt and braces.)
    def _pass_begin(self):
        if torch.cuda.is_current_stream_capturing():
            return
        last = getattr(self, "_last_pass", None)
        if last is not None:
            cur = torch.cuda.current_stream(self.dev)
            if last[0] != cur:
                cur.wait_event(last[1])

    def _pass_end(self):
        if torch.cuda.is_current_stream_capturing():
            return
        cur = torch.cuda.current_stream(self.dev)
        ev = getattr(self, "_pass_event", None)
        if ev is None:
            ev = self._pass_event = torch.cuda.Event()
        ev.record(cur)
        self._last_pass = (cur, ev)

    # `hook(i)`, when set, is called after the i-th launch of a forward (0 = the stem / fused head) on the launch stream: a
    # caller that software-pipelines OTHER work between the detector's layers (bench.py: the next batch's ISP filters) puts
    # it there instead of on a second stream whose workgroups would time-slice the CUs with the conv kernels'.
    hook = None

    def num_launches(self):
        """Launches one forward issues (the hook's index range)."""
        if self._stem is None:
            return len(self.plan)
        if self.fuse_head:
            return 1 + len(self.plan) - (3 if self._head_next is not None else 2)
        return len(self.plan)

    def forward(self, img):
        """img: planar fp32 [B,3,H,W] in [0,1] on the engine's device -> pred fp32 [B, N, 85] (eval decode)."""
        if img.shape != (self.B, 3, self.H, self.W) or img.dtype != torch.float32 or img.device != self.dev:
            raise ValueError(f"expected fp32 {(self.B, 3, self.H, self.W)} on {self.dev}, got {img.dtype} "
                             f"{tuple(img.shape)} on {img.device}")
        img = img.contiguous()
        if getattr(self, "chains", None):
            self.check_chains()                              # (a host memory read per chain: no device call)
        with torch.cuda.device(self.dev):
            self._pass_begin()
            try:
                return self._forward_launches(img)
            finally:
                self._pass_end()

    def _forward_launches(self, img):
        """The launch sequence of one forward on the current stream (see forward)."""
        st = _lib.stream_ptr()
        if self._stem is None:                               # generic first conv: pack launch takes the image pointer
            for kind, fn, args in self.plan:
                rc = fn(ctypes.c_void_p(img.data_ptr()), *args[1:], st) if kind == "pack" else fn(*args, st)
                if rc != 0:
                    _lib.check(rc, f"adayolo {kind}")
            return self.pred
        w, b, out = self._stem
        if self.fuse_head:
            d, n = self._head_down, self._head_next
            rc = self.L.adayolo_stem_down_fwd(ctypes.c_void_p(img.data_ptr()), ctypes.c_void_p(w.data_ptr()),
                                              ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(d["w"].data_ptr()),
                                              ctypes.c_void_p(d["b"].data_ptr()), ctypes.c_void_p(d["dst"].ptr), d["dst"].cs,
                                              self.B, self.H, self.W, self.Hp, self.pad_top, LETTERBOX_VALUE,
                                              ctypes.c_void_p(n["w"].data_ptr()) if n else None,
                                              ctypes.c_void_p(n["b"].data_ptr()) if n else None,
                                              ctypes.c_void_p(n["dst"].ptr) if n else None, n["dst"].cs if n else 0, st)
            if rc != 0:
                _lib.check(rc, "adayolo stem_down")
            hook = self.hook
            if hook is not None:
                hook(0)
            for li, (kind, fn, args) in enumerate(self.plan[(3 if n else 2):]):
                rc = fn(*args, st)
                if rc != 0:
                    _lib.check(rc, f"adayolo {kind}")
                if hook is not None:
                    hook(li + 1)
            return self.pred
        nc = self.head_chunks if (len(self.plan) > 1 and self.plan[0][0] == "stem" and self.plan[1][0] == "conv") else 1
        Bc = self.B // nc
        for c in range(nc):                                   # [stem, first conv] per batch chunk
            rc = self.L.adayolo_stem_fwd(ctypes.c_void_p(img.data_ptr() + c * Bc * 3 * self.H * self.W * 4),
                                         ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(b.data_ptr()),
                                         ctypes.c_void_p(out.ptr + c * Bc * self.Hp * self.W * out.cs * 2), out.cs,
                                         Bc, self.H, self.W, self.Hp, self.pad_top, LETTERBOX_VALUE, 32, st)
            if rc != 0:
                _lib.check(rc, "adayolo stem")
            if nc > 1:
                _, fn, args = self.plan[1]
                a = list(args)
                H, W, s = a[9], a[10], a[14]
                Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
                a[0] = ctypes.c_void_p(a[0].value + c * Bc * H * W * a[1] * 2)
                a[6] = ctypes.c_void_p(a[6].value + c * Bc * Ho * Wo * a[7] * 2)
                a[8] = Bc
                rc = fn(*a, st)
                if rc != 0:
                    _lib.check(rc, "adayolo conv")
        for kind, fn, args in self.plan[(2 if nc > 1 else 1):]:
            rc = fn(*args, st)
            if rc != 0:
                _lib.check(rc, f"adayolo {kind}")
        return self.pred

    __call__ = forward

    def raw_maps(self):
        """Training-style outputs [B, na, ny, nx, no] (fp32 copies of the raw head maps)."""
        outs = []
        for v in self.raw:
            t = v.buf[..., : self.na * self.no].float()
            outs.append(t.view(self.B, v.H, v.W, self.na, self.no).permute(0, 3, 1, 2, 4).contiguous())
        return outs


def chain_runs(entries, cus, chain_min):
    """Which runs of a launch plan become persistent chains (YoloEngine.fuse_chains): `entries` = (eligible, tiles) per launch;
    a run = consecutive eligible launches with MORE tiles than `cus`, plus at most one trailing eligible launch below that (it fills
    the run's own tail); runs shorter than `chain_min` launches stay as they are. Returns [(first, end)), ...]."""
    out, i, n = [], 0, len(entries)
    while i < n:
        j = i
        while j < n and entries[j][0] and entries[j][1] > cus:
            j += 1
        if i < j < n and entries[j][0]:
            j += 1
        if j - i >= chain_min:
            out.append((i, j))
        i = max(j, i + 1)
    return out


def _src(i, f):
    """Absolute index of the layer that feeds layer i through the (possibly relative) reference `from` field."""
    return i + f if f < 0 else f
