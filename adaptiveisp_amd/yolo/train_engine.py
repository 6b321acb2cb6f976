"""Training-mode detector on the HIP path: raw head maps with a gradient back to the input image.

The reward model is frozen (train.py:239-243) and its BatchNorms stay in eval mode, so the forward is the same folded
conv stack as YoloEngine; what training adds is (a) the three raw head maps instead of the decoded prediction
(Detect.forward in training mode, yolo.py:56-60) and (b) the data gradient through every layer down to the retouched
image (train.py:267-271,341-342) — no weight gradients.

Forward (train): every Conv keeps its pre-activation P (bf16): conv kernel without activation -> P, then
`adayolo_silu_fwd` writes silu(P) (+ shortcut) into the activation. Backward, layer by layer in reverse:
    dP = dY * silu'(P)  (adayolo_silu_bwd; the shortcut gradient is added into the block input's gradient there)
    dX (+)= conv(dP, W^T flipped)   — the SAME implicit-GEMM MFMA kernels as the forward (adayolo_conv_fwd_variant);
                                      stride-2 layers convolve the zero-inserted dP (adayolo_zero_insert2x)
Upsample backward sums 2x2 blocks, Concat is free (gradients are read from channel slices), the stem's gradient is
converted from NHWC bf16 to the planar fp32 image layout without the letterbox rows.
Gradients are bf16 tensors (fp32 accumulation inside every kernel), like the activations.
"""
import ctypes
import os

import torch

from . import _lib
from .engine import LETTERBOX_VALUE, YoloEngine, _View


class YoloTrainEngine(YoloEngine):
    def __init__(self, model, batch, height, width, device="cuda:0", share_with=None, first_image=0, capture=("fwd", "bwd")):
        """`share_with` / `first_image`: this engine's activation buffers are images [first_image, first_image + batch) of
        that (larger-batch) engine's — its backward then runs on what THAT engine's forward kept (YoloTrainPairEngine).
        `capture`: which launch sequences are replayed from hipGraphs."""
        if share_with is not None:
            self._shared = (share_with, int(first_image))
        self._capture = tuple(capture)
        super().__init__(model, batch, height, width, device)
        self._gen = 0
        self._build_train()

    # ------------------------------------------------------------------------------------------
    def _dense(self, H, W, C):
        return _View(self._new(H, W, C), 0, C)

    def _conv_entry(self, src, w, b, dst, k, s, act, res, cout, variant=0):
        args = [ctypes.c_void_p(src.ptr), src.cs, ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(b.data_ptr()),
                ctypes.c_void_p(res.ptr) if res is not None else None, res.cs if res is not None else 0,
                ctypes.c_void_p(dst.ptr), dst.cs, self.B, src.H, src.W, src.C, cout, k, s, act, variant]
        return ("conv", self._conv_launch, args)

    def _build_train(self):
        L, B = self.L, self.B
        self.tfwd, self.tbwd = [], []
        gbuf = {}                                   # id(activation buffer) -> gradient buffer of the same shape

        def G(v):
            if id(v.buf) not in gbuf:
                gbuf[id(v.buf)] = torch.zeros_like(v.buf)
                self._keep.append(gbuf[id(v.buf)])
            return _View(gbuf[id(v.buf)], v.coff, v.C)

        written = []                                # (id(grad buffer), coff, C) ranges that already hold a gradient

        def is_written(v):
            return any(i == id(v.buf) and c0 <= v.coff and v.coff + v.C <= c0 + C for i, c0, C in written)

        def mark(v):
            written.append((id(v.buf), v.coff, v.C))

        zeros_bias = {}

        def zb(n):
            if n not in zeros_bias:
                zeros_bias[n] = torch.zeros(n, dtype=torch.float32, device=self.dev)
                self._keep.append(zeros_bias[n])
            return zeros_bias[n]

        bwd = []                                    # built in forward order, reversed at the end
        for op in self.ops:
            if op["kind"] == "pack":
                raise _lib.AdayoloError("YoloTrainEngine serves the yolov3.yaml stem (Conv 3->32 k3 s1); width-scaled "
                                        "detectors run forward-only (YoloEngine)")
            if op["kind"] == "stem":
                dst = op["dst"]
                P = self._dense(dst.H, dst.W, 32)
                self._stem_pre = P
                self.tfwd.append(("stem", None, None))
                self.tfwd.append(("silu", L.adayolo_silu_fwd, (ctypes.c_void_p(P.ptr), P.cs, None, 0, ctypes.c_void_p(dst.ptr),
                                                                dst.cs, B * dst.H * dst.W, 32)))
                # backward: dP0 = dY0 * silu'(P0); 3x3 data gradient to 3 (padded to 8) channels; unpack to planar fp32
                w = op["w"]                                                         # fp32 [32][3][3][3] (co,kh,kw,ci)
                wt = torch.zeros(8, 3, 3, 32, dtype=torch.float32, device=self.dev)
                wt[:3] = w.flip(1, 2).permute(3, 1, 2, 0)
                wt = wt.to(torch.bfloat16).contiguous()
                dP, gimg = self._dense(dst.H, dst.W, 32), self._dense(dst.H, dst.W, 8)
                self._keep.append(wt)
                self._gimg, self._stem_dst, self._stem_dP = gimg, dst, dP
                bwd.append([("dsilu", L.adayolo_silu_bwd, lambda dst=dst, P=P, dP=dP: (
                                ctypes.c_void_p(G(dst).ptr), dst.cs, ctypes.c_void_p(P.ptr), P.cs, ctypes.c_void_p(dP.ptr),
                                dP.cs, None, 0, 0, B * dst.H * dst.W, 32)),
                            self._conv_entry(dP, wt, zb(8), gimg, 3, 1, _lib.ACT_NONE, None, 8),
                            ("imggrad", None, None)])
            elif op["kind"] == "up":
                src, dst = op["src"], op["dst"]
                self.tfwd.append(("up", L.adayolo_upsample2x, (ctypes.c_void_p(src.ptr), src.cs, ctypes.c_void_p(dst.ptr),
                                                                dst.cs, B, src.H, src.W, src.C)))
                bwd.append([("upbwd", L.adayolo_upsample2x_bwd, ("up", src, dst))])
            else:
                src, dst, res, k, s, act, cout = (op[n] for n in ("src", "dst", "res", "k", "s", "act", "cout"))
                w, b = op["w"], op["b"]
                wt = w.flip(1, 2).permute(3, 1, 2, 0).contiguous()                    # [Cin][k][k][Cout], taps flipped
                self._keep.append(wt)
                if act == _lib.ACT_SILU:
                    P = self._dense(dst.H, dst.W, cout)
                    self.tfwd.append(self._conv_entry(src, w, b, P, k, s, _lib.ACT_NONE, None, cout))
                    self.tfwd.append(("silu", L.adayolo_silu_fwd, (ctypes.c_void_p(P.ptr), P.cs,
                                                                    ctypes.c_void_p(res.ptr) if res is not None else None,
                                                                    res.cs if res is not None else 0, ctypes.c_void_p(dst.ptr),
                                                                    dst.cs, B * dst.H * dst.W, cout)))
                    dP = self._dense(dst.H, dst.W, cout)
                else:
                    P, dP = None, None
                    self.tfwd.append(self._conv_entry(src, w, b, dst, k, s, act, None, cout))
                # stride 2: the data gradient as a 2x2 conv over the OUTPUT grid with depth-to-space stores
                # (adayolo_conv_s2grad_fwd, include/adayolo.h) where the sizes are even; zero insertion + 3x3 otherwise
                w4 = None
                if s == 2 and k == 3 and src.H == 2 * dst.H and src.W == 2 * dst.W and os.environ.get("ADAYOLO_TRAIN_S2GRAD", "1") == "1":
                    cin = src.C
                    w4 = torch.zeros(4 * cin, 2, 2, cout, dtype=w.dtype, device=self.dev)
                    KH = {(0, 0): 1, (1, 0): 2, (1, 1): 0}
                    for (pa, dh), kh in KH.items():
                        for (pb, dw), kw in KH.items():
                            w4[(2 * pa + pb) * cin:(2 * pa + pb + 1) * cin, dh, dw] = w[:, kh, kw, :cin].t()
                    self._keep.append(w4)
                U = self._dense(src.H, src.W, cout) if (s == 2 and w4 is None) else None
                bwd.append([("convbwd", None, dict(src=src, dst=dst, res=res, k=k, s=s, cout=cout, wt=wt, P=P, dP=dP, U=U, w4=w4))])

        # resolve the backward launches in reverse order (accumulate-or-overwrite is decided here, once); `meta` says
        # which views each launch reads and writes (what _backward_plan needs to move a SiLU' into its producer)
        meta = self._tbwd_meta = []
        for group in reversed(bwd):
            for kind, fn, a in group:
                if kind == "convbwd":
                    src, dst, res, k, s, cout, wt, P, dP, U, w4 = (a[n] for n in ("src", "dst", "res", "k", "s", "cout", "wt", "P", "dP", "U", "w4"))
                    gdst = G(dst)
                    if P is not None:
                        gres, acc = (G(res), int(is_written(G(res)))) if res is not None else (None, 0)
                        self.tbwd.append(("dsilu", self.L.adayolo_silu_bwd, (
                            ctypes.c_void_p(gdst.ptr), gdst.cs, ctypes.c_void_p(P.ptr), P.cs, ctypes.c_void_p(dP.ptr), dP.cs,
                            ctypes.c_void_p(gres.ptr) if gres is not None else None, gres.cs if gres is not None else 0, acc,
                            self.B * dst.H * dst.W, cout)))
                        meta.append(dict(reads=[gdst] + ([gres] if acc else []), writes=[dP] + ([gres] if gres is not None else []),
                                         gy=gdst, P=P, dP=dP, gres=gres, acc=acc))
                        if gres is not None:
                            mark(gres)
                        g_in = dP
                    else:
                        g_in = gdst                                                    # no activation: dP is dY itself
                    if w4 is not None:                                                 # stride 2 without zero insertion
                        gsrc = G(src)
                        acc_view = gsrc if is_written(gsrc) else None
                        gin_view = _View(g_in.buf, g_in.coff, cout)
                        gin_view.H, gin_view.W = dst.H, dst.W
                        self.tbwd.append(self._conv_entry(gin_view, w4, zb(4 * src.C), gsrc, 2, 1, _lib.ACT_NONE, acc_view, 4 * src.C, variant=5))
                        meta.append(dict(reads=[gin_view] + ([acc_view] if acc_view is not None else []), writes=[gsrc], out=gsrc,
                                         res=acc_view))
                        mark(gsrc)
                        continue
                    if s == 2:
                        self.tbwd.append(("zins", self.L.adayolo_zero_insert2x, (
                            ctypes.c_void_p(g_in.ptr), g_in.cs, ctypes.c_void_p(U.ptr), U.cs, self.B, dst.H, dst.W, src.H, src.W, cout)))
                        meta.append(dict(reads=[g_in], writes=[U]))
                        g_in = U
                    gsrc = G(src)
                    acc_view = gsrc if is_written(gsrc) else None
                    # the data gradient is a stride-1 conv from `cout` channels to the layer's input channels
                    gin_view = _View(g_in.buf, g_in.coff, cout)
                    gin_view.H, gin_view.W = (src.H, src.W) if s == 2 else (dst.H, dst.W)
                    self.tbwd.append(self._conv_entry(gin_view, wt, zb(src.C), gsrc, k, 1, _lib.ACT_NONE, acc_view, src.C))
                    meta.append(dict(reads=[gin_view] + ([acc_view] if acc_view is not None else []), writes=[gsrc], out=gsrc,
                                     res=acc_view))
                    mark(gsrc)
                elif kind == "upbwd":
                    _, src, dst = a
                    gsrc, gdst = G(src), G(dst)
                    self.tbwd.append(("upbwd", fn, (ctypes.c_void_p(gdst.ptr), gdst.cs, ctypes.c_void_p(gsrc.ptr), gsrc.cs,
                                                    int(is_written(gsrc)), self.B, src.H, src.W, src.C)))
                    meta.append(dict(reads=[gdst] + ([gsrc] if is_written(gsrc) else []), writes=[gsrc]))
                    mark(gsrc)
                elif kind == "dsilu":                                                   # the stem's
                    self.tbwd.append((kind, fn, a()))
                    meta.append(dict(reads=[G(self._stem_dst)], writes=[self._stem_dP], gy=G(self._stem_dst), P=self._stem_pre,
                                     dP=self._stem_dP, gres=None, acc=0))
                elif kind == "conv":                                                    # the stem's data gradient
                    self.tbwd.append((kind, fn, a))
                    meta.append(dict(reads=[self._stem_dP], writes=[self._gimg], out=self._gimg, res=None))
                else:
                    self.tbwd.append((kind, fn, a))
                    meta.append(dict(reads=[self._gimg], writes=[]))
        self._graw = [G(v) for v in self.raw]

    def _plans(self):
        return [self.plan, [e for e in self.tfwd if e[0] == "conv"], [e for e in self.tbwd if e[0] == "conv"]]

    def _tune_penalty(self, key, variant):
        """A forward conv followed by SiLU runs as ONE launch only on the kernels that can store the pre-activation
        (KEEP_VARIANTS); any other choice adds the adayolo_silu_fwd launch of that layer, whose duration is measured here
        (best of 5, once per layer shape) and added to the candidate's time."""
        if variant in self.KEEP_VARIANTS:
            return 0.0
        cost = getattr(self, "_silu_cost", None)
        if cost is None:
            cost = self._silu_cost = {}
            st = _lib.stream_ptr()
            for i, e in enumerate(self.tfwd[:-1]):
                nxt = self.tfwd[i + 1]
                k = tuple(e[2][8:16]) if e[0] == "conv" else None
                if k is not None and k not in cost and nxt[0] == "silu" and e[2][6].value == nxt[2][0].value:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    nxt[1](*nxt[2], st)
                    t = float("inf")
                    for _ in range(5):
                        e0.record()
                        nxt[1](*nxt[2], st)
                        e1.record()
                        e1.synchronize()
                        t = min(t, e0.elapsed_time(e1))
                    cost[k] = t
        return cost.get(key, 0.0)

    # ------------------------------------------------------------------------------------------
    def _run(self, plan, img=None, grad_img=None):
        st = _lib.stream_ptr()
        for kind, fn, args in plan:
            if kind == "stem":
                w, b, _ = self._stem
                P = self._stem_pre
                rc = self.L.adayolo_stem_fwd_act(ctypes.c_void_p(img.data_ptr()), ctypes.c_void_p(w.data_ptr()),
                                                 ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(P.ptr), P.cs, self.B, self.H,
                                                 self.W, self.Hp, self.pad_top, LETTERBOX_VALUE, 32, _lib.ACT_NONE, st)
            elif kind == "stemkeep":                       # stem + its SiLU in one launch (args: the SiLU launch's)
                w, b, _ = self._stem
                P = self._stem_pre
                rc = self.L.adayolo_stem_keep_fwd(ctypes.c_void_p(img.data_ptr()), ctypes.c_void_p(w.data_ptr()),
                                                  ctypes.c_void_p(b.data_ptr()), args[4], args[5], ctypes.c_void_p(P.ptr), P.cs,
                                                  self.B, self.H, self.W, self.Hp, self.pad_top, LETTERBOX_VALUE, 32, st)
            elif kind == "imggrad":
                g = self._gimg
                rc = self.L.adayolo_image_grad(ctypes.c_void_p(g.ptr), g.cs, ctypes.c_void_p(grad_img.data_ptr()), self.B,
                                               self.H, self.W, self.Hp, self.pad_top, st)
            else:
                rc = fn(*args, st)
            if rc != 0:
                _lib.check(rc, f"adayolo {kind}")

    # kernels whose epilogue stores the pre-activation beside the activation
    KEEP_VARIANTS = (5, 22, 26, 27, 60, 80, 85) + YoloEngine.SPLITK_CANDIDATES

    def _conv_keep_launch(self, *a):
        """adayolo_conv_keep_fwd's argument list (19 + stream); the split-K variants go to their own entry point."""
        if a[18] >= self.SPLITK_BASE:
            _, ptr, nbytes = self._splitk_workspace()
            return self.L.adayolo_conv_splitk_fwd(*a[:19], ptr, nbytes, a[19])
        return self.L.adayolo_conv_keep_fwd(*a)

    def _forward_plan(self):
        """tfwd with every [conv -> pre-activation, SiLU(+residual)] pair whose tuned kernel can do both in one launch
        (adayolo_conv_keep_fwd, bit-identical to the pair) replaced by that launch; rebuilt when autotune changed a variant."""
        import os
        sig = tuple(args[16] for kind, _, args in self.tfwd if kind == "conv")
        cached = getattr(self, "_tfwd_fused", None)
        if cached is not None and cached[0] == sig:
            return cached[1]
        plan, i = [], 0
        fuse = os.environ.get("ADAYOLO_TRAIN_KEEP", "1") == "1"
        while i < len(self.tfwd):
            e = self.tfwd[i]
            nxt = self.tfwd[i + 1] if i + 1 < len(self.tfwd) else None
            if (fuse and e[0] == "stem" and nxt is not None and nxt[0] == "silu" and nxt[2][0].value == self._stem_pre.ptr
                    and nxt[2][2] is None):
                plan.append(("stemkeep", None, nxt[2]))
                i += 2
                continue
            if (fuse and e[0] == "conv" and nxt is not None and nxt[0] == "silu" and e[2][16] in self.KEEP_VARIANTS
                    and e[2][6].value == nxt[2][0].value):        # the conv's output IS the SiLU kernel's pre-activation
                a, sl = e[2], nxt[2]
                # conv args: in, in_cs, w, b, res(None), 0, out(=P), out_cs, B, H, W, Cin, Cout, k, s, act(NONE), variant
                # silu args: pre, pre_cs, res, res_cs, out, out_cs, npix, C
                args = (a[0], a[1], a[2], a[3], sl[2], sl[3], sl[4], sl[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13],
                        a[14], _lib.ACT_SILU, a[16])
                rc = self._conv_keep_launch(*args, _lib.stream_ptr())   # probe once: ESHAPE = this kernel does not serve the shape
                if rc == 0:
                    plan.append(("convkeep", self._conv_keep_launch, args))
                    i += 2
                    continue
            plan.append(e)
            i += 1
        self._tfwd_fused = (sig, plan)
        self.keep_fused = sum(1 for e in plan if e[0] == "convkeep")
        return plan

    def _conv_dsilu_launch(self, *a):
        """adayolo_conv_dsilu_fwd's argument list up to `variant` (20) + stream; the workspace is the engine's. k = 2: the
        stride-2 data gradient with the same epilogue (adayolo_conv_s2grad_fwd)."""
        if a[17] == 2:
            ptr, nbytes = (self._splitk_workspace()[1:]) if a[19] >= self.SPLITK_BASE else (None, 0)
            return self.L.adayolo_conv_s2grad_fwd(*a[:12], a[12], a[13], a[14], a[15], a[16] // 4, a[19], ptr, nbytes, a[20])
        if a[19] >= self.SPLITK_BASE:
            _, ptr, nbytes = self._splitk_workspace()
            return self.L.adayolo_conv_dsilu_fwd(*a[:20], ptr, nbytes, a[20])
        return self.L.adayolo_conv_dsilu_fwd(*a[:20], None, 0, a[20])

    def _backward_plan(self):
        """tbwd with every SiLU' launch moved into the data-gradient conv that COMPLETES the gradient it reads
        (adayolo_conv_dsilu_fwd, bit-identical to the pair), where that conv writes exactly the view, runs on a kernel with
        that epilogue (KEEP_VARIANTS) and nothing else reads the gradient:
          * no shortcut: the conv writes only dL/d(pre) — the layer-output gradient is never stored;
          * Bottleneck.cv2 (its SiLU' also copied the output gradient into the block input's): the conv stores the output
            gradient as well, and the conv that accumulated into the copy (cv1's data gradient) takes it from there as
            its residual — the copy is gone. Only when the copy was the first write of that view and that conv the next.
        Rebuilt when autotune changed a variant; ADAYOLO_TRAIN_FUSE_DSILU=0 keeps every launch."""
        import os
        sig = tuple(args[16] for kind, _, args in self.tbwd if kind == "conv")
        cached = getattr(self, "_tbwd_fused", None)
        if cached is not None and cached[0] == sig:
            return cached[1]
        E, M = self.tbwd, self._tbwd_meta
        n = len(E)
        key = lambda v: (id(v.buf), v.coff, v.C)                                        # noqa: E731
        overlap = lambda a, b: a.buf is b.buf and a.coff < b.coff + b.C and b.coff < a.coff + a.C   # noqa: E731
        touches = lambda m, v, what: any(overlap(x, v) for x in m[what])               # noqa: E731
        drop, ds, res_from = set(), {}, {}

        def fused_args(j, m, store):
            a = list(E[j][2])
            if j in res_from:
                a[4], a[5] = ctypes.c_void_p(res_from[j].ptr), res_from[j].cs
            return a[:6] + [a[6] if store else None, a[7] if store else 0, ctypes.c_void_p(m["P"].ptr), m["P"].cs,
                            ctypes.c_void_p(m["dP"].ptr), m["dP"].cs] + a[8:15] + [a[16]]

        def served(j, m, store):                          # one real launch: the named kernel must take the shape in this form
            return self._conv_dsilu_launch(*fused_args(j, m, store), _lib.stream_ptr()) == 0

        if os.environ.get("ADAYOLO_TRAIN_FUSE_DSILU", "1") == "1":
            for i in range(n):
                m = M[i]
                if E[i][0] != "dsilu":
                    continue
                gy = m["gy"]
                j = next((t for t in range(i - 1, -1, -1) if touches(M[t], gy, "writes")), None)
                if j is None or E[j][0] != "conv" or key(M[j]["out"]) != key(gy) or E[j][2][16] not in self.KEEP_VARIANTS:
                    continue
                if any(touches(M[t], gy, "reads") for t in range(j + 1, i)):
                    continue
                if m["gres"] is None:
                    if not any(touches(M[t], gy, "reads") for t in range(i + 1, n)) and served(j, m, False):
                        ds[j] = (m, False)
                        drop.add(i)
                    continue
                gres = m["gres"]
                if m["acc"]:
                    continue
                k = next((t for t in range(i + 1, n) if touches(M[t], gres, "writes") or touches(M[t], gres, "reads")), None)
                if (k is None or E[k][0] != "conv" or M[k]["res"] is None or key(M[k]["res"]) != key(gres) or
                        key(M[k]["out"]) != key(gres)):
                    continue
                if any(touches(M[t], gy, "writes") for t in range(i + 1, k + 1)) or not served(j, m, True):
                    continue
                ds[j] = (m, True)
                res_from[k] = gy
                drop.add(i)
        plan = []
        for i in range(n):
            if i in drop:
                continue
            kind, fn, a = E[i]
            if i in ds:
                plan.append(("convds", self._conv_dsilu_launch, fused_args(i, *ds[i])))
            elif i in res_from:
                a = list(a)
                a[4], a[5] = ctypes.c_void_p(res_from[i].ptr), res_from[i].cs
                plan.append((kind, fn, a))
            else:
                plan.append((kind, fn, a))
        self._tbwd_fused = (sig, plan)
        self.dsilu_fused = len(drop)
        return plan

    # ---- launch sequences as hipGraphs --------------------------------------------------------------------------
    # A training iteration is ~450 detector launches (2 forwards of 75 convs + 72 SiLU kernels, one backward of ~150) next
    # to ~1000 small PyTorch launches, and it is the HOST that is saturated (tools/train_trace.sh: GPU busy 43 % of an
    # iteration): the two fixed launch sequences are captured once each (static image / image-gradient buffers, the
    # engine's own activation and gradient buffers) and replayed. ADAYOLO_TRAIN_GRAPH=0 launches them one by one.
    def _graph(self, which):
        import os
        if os.environ.get("ADAYOLO_TRAIN_GRAPH", "1") != "1" or torch.cuda.is_current_stream_capturing():
            return None
        sig = tuple(args[16] for kind, _, args in self.tfwd + self.tbwd if kind == "conv")    # re-capture after autotune
        st = getattr(self, "_graphs", None)
        if st is None or st["sig"] != sig:
            st = self._graphs = dict(sig=sig, fwd=None, bwd=None,
                                     img=torch.empty((self.B, 3, self.H, self.W), dtype=torch.float32, device=self.dev),
                                     grad_img=torch.empty((self.B, 3, self.H, self.W), dtype=torch.float32, device=self.dev))
        if st["fwd"] is None:
            # both sequences at the first forward: the warm-up launches then run on buffers nobody has filled yet
            with torch.cuda.device(self.dev):
                torch.cuda.synchronize()
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                cap = self._capture
                with torch.cuda.stream(side):             # warm-up outside the capture (lazy kernel attributes)
                    if "fwd" in cap:
                        self._run(self._forward_plan(), img=st["img"])
                    if "bwd" in cap:
                        self._run(self._backward_plan(), grad_img=st["grad_img"])
                torch.cuda.current_stream().wait_stream(side)
                for key, make, kw in (("fwd", self._forward_plan, dict(img=st["img"])), ("bwd", self._backward_plan, dict(grad_img=st["grad_img"]))):
                    if key not in cap:
                        st[key] = False                   # this sequence is launched eagerly (its plan is not even built here:
                        continue                          # building probes launches on the engine's buffers)
                    plan = make()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        self._run(plan, **kw)
                    st[key] = g
        return st

    def _forward_raw(self, img):
        """The forward launch sequence: the raw head maps land in self.raw (bf16, the engine's buffers)."""
        st = self._graph("fwd")
        with torch.cuda.device(self.dev):
            self._pass_begin()
            try:
                if st is None or not st["fwd"]:
                    self._run(self._forward_plan(), img=img)
                else:
                    st["img"].copy_(img)
                    st["fwd"].replay()
            finally:
                self._pass_end()
        self._gen += 1

    def _forward_raw_halves(self, first, second):
        """_forward_raw of the batch [first; second] (two [B/2,3,H,W] tensors) without materialising the concatenation when
        the sequence replays from a graph (the halves are copied into the graph's input buffer)."""
        st = self._graph("fwd")
        if st is None or not st["fwd"]:
            return self._forward_raw(torch.cat([first, second], 0))
        h = self.B // 2
        with torch.cuda.device(self.dev):
            self._pass_begin()
            try:
                st["img"][:h].copy_(first)
                st["img"][h:].copy_(second)
                st["fwd"].replay()
            finally:
                self._pass_end()
        self._gen += 1

    def _backward_raw(self):
        """The backward launch sequence from the head-gradient buffers (self._graw) -> d loss / d img."""
        st = self._graph("bwd")
        with torch.cuda.device(self.dev):
            self._pass_begin()
            try:
                if st is None or not st["bwd"]:
                    grad_img = torch.empty((self.B, 3, self.H, self.W), dtype=torch.float32, device=self.dev)
                    self._run(self._backward_plan(), grad_img=grad_img)
                    return grad_img
                st["bwd"].replay()
                return st["grad_img"].clone()
            finally:
                self._pass_end()

    def forward_train(self, img):
        """img planar fp32 [B,3,H,W] on the device -> the three raw head maps [B,na,ny,nx,no] fp32."""
        if img.shape != (self.B, 3, self.H, self.W) or img.dtype != torch.float32 or img.device != self.dev:
            raise ValueError(f"expected fp32 {(self.B, 3, self.H, self.W)} on {self.dev}")
        self._forward_raw(img.contiguous())
        return self.raw_maps()

    def backward_image(self, grads):
        """grads: three tensors shaped like forward_train's outputs -> d loss / d img, planar fp32 [B,3,H,W]."""
        for g, gv, v in zip(grads, self._graw, self.raw):
            t = gv.buf
            t.zero_()
            if g is not None:
                t[..., : self.na * self.no] = g.permute(0, 2, 3, 1, 4).reshape(self.B, v.H, v.W, self.na * self.no).to(torch.bfloat16)
        return self._backward_raw()

    # ---- fused per-image detection loss on the raw maps (csrc/yolo_loss.hip) -----------------------------------
    def head_shapes(self):
        """Stand-ins for the three head maps where only their SHAPES are needed (loss.assign_labels)."""
        import types
        return [types.SimpleNamespace(shape=(self.B, self.na, v.H, v.W, self.no), device=self.dev) for v in self.raw]

    def _loss_workspace(self):
        """What the loss forward leaves for its backward (objectness targets, match counts) + the reduction scratch."""
        ws = getattr(self, "_loss_ws", None)
        if ws is None:
            dev = self.dev
            ws = self._loss_ws = dict(
                tobj=[torch.empty((self.B, self.na, v.H, v.W), dtype=torch.float32, device=dev) for v in self.raw],
                cnt=[torch.empty((self.B,), dtype=torch.float32, device=dev) for _ in self.raw],
                part=[torch.empty((self.B, 3), dtype=torch.float32, device=dev) for _ in self.raw],
                ticket=torch.zeros((self.B,), dtype=torch.int32, device=dev))
        return ws

    def _loss_args(self, loss_fn, packed):
        """adayolo_loss_args over this engine's raw maps / gradient buffers and the packed target assignment."""
        dev, nl = self.dev, len(self.raw)
        ws = self._loss_workspace()
        a = _lib.LossArgs()
        keep = []
        for i, (v, gv, (idx, box)) in enumerate(zip(self.raw, self._graw, packed)):
            n = int(idx.shape[0])
            L = a.layer[i]
            L.raw, L.cs, L.ny, L.nx, L.balance = v.ptr, v.cs, v.H, v.W, float(loss_fn.balance[i])
            L.idx, L.box, L.n = idx.data_ptr() if n else None, box.data_ptr() if n else None, n
            L.part, L.tobj, L.cnt = ws["part"][i].data_ptr(), ws["tobj"][i].data_ptr(), ws["cnt"][i].data_ptr()
            L.grad, L.grad_cs = gv.ptr, gv.cs
            keep += [idx, box]
        a.nl, a.B, a.na, a.nc, a.no = nl, self.B, self.na, loss_fn.nc, self.no
        a.ticket = ws["ticket"].data_ptr()
        h = loss_fn.hyp
        a.hyp_box, a.hyp_obj, a.hyp_cls = float(h["box"]), float(h["obj"]), float(h["cls"])
        a.cp, a.cn, a.cls_pw, a.obj_pw = float(loss_fn.cp), float(loss_fn.cn), float(h["cls_pw"]), float(h["obj_pw"])
        return a, keep

    def _loss_forward(self, loss_fn, img, packed):
        if img.shape != (self.B, 3, self.H, self.W) or img.dtype != torch.float32 or img.device != self.dev:
            raise ValueError(f"expected fp32 {(self.B, 3, self.H, self.W)} on {self.dev}")
        if loss_fn.nc + 5 != self.no or len(packed) != len(self.raw) or loss_fn.hyp.get("fl_gamma", 0.0) != 0.0:
            raise ValueError("loss / detector mismatch (classes, layers) or focal loss requested: use the PyTorch loss")
        loss = torch.empty((self.B,), dtype=torch.float32, device=self.dev)
        self._forward_raw(img.contiguous())
        with torch.cuda.device(self.dev):
            a, keep = self._loss_args(loss_fn, packed)
            a.loss = loss.data_ptr()
            _lib.check(self.L.adayolo_detloss_fwd(ctypes.byref(a), _lib.stream_ptr()), "adayolo_detloss_fwd")
        return loss.view(self.B, 1)

    def per_sample_loss(self, loss_fn, img, packed):
        """[B,1] per-image detection losses of `img` through the detector (reference train.py:175-197, 262-271) with the
        target assignment `packed` (loss.pack_assigned(loss.assign_labels(...))), on the fused HIP kernels: detector forward,
        one loss launch on the bf16 head maps; with autograd: two launches write the head-map gradients, then the detector's
        backward. The same numbers as `loss.batched_per_sample_loss(loss_fn, engine(img), ...)` (tests/test_gpu_yolo_train.py)."""
        if torch.is_grad_enabled() and img.requires_grad:
            return _DetectorLossFn.apply(img, self, loss_fn, packed)
        return self._loss_forward(loss_fn, img, packed)

    def __call__(self, img):
        """Differentiable detector: raw maps with autograd to `img` (the frozen reward model of the RL loop)."""
        if torch.is_grad_enabled() and img.requires_grad:
            return list(_DetectorFn.apply(img, self))
        return self.forward_train(img)


class _DetectorLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, engine, loss_fn, packed):
        loss = engine._loss_forward(loss_fn, img, packed)
        ctx.engine, ctx.gen, ctx.loss_fn, ctx.packed = engine, engine._gen, loss_fn, packed
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        eng = ctx.engine
        if ctx.gen != eng._gen:
            raise RuntimeError("YoloTrainEngine: backward after a newer forward overwrote the saved pre-activations "
                               "(one engine holds one set of buffers; use a second engine for interleaved graphs)")
        g = grad_loss.reshape(eng.B).float().contiguous()
        with torch.cuda.device(eng.dev):
            a, keep = eng._loss_args(ctx.loss_fn, ctx.packed)
            scratch = torch.empty((eng.B,), dtype=torch.float32, device=eng.dev)
            a.loss, a.grad_loss = scratch.data_ptr(), g.data_ptr()
            _lib.check(eng.L.adayolo_detloss_bwd(ctypes.byref(a), _lib.stream_ptr()), "adayolo_detloss_bwd")
        return eng._backward_raw(), None, None, None


class _DetectorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, engine):
        outs = engine.forward_train(img)
        ctx.engine, ctx.gen = engine, engine._gen
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        eng = ctx.engine
        if ctx.gen != eng._gen:
            raise RuntimeError("YoloTrainEngine: backward after a newer forward overwrote the saved pre-activations "
                               "(one engine holds one set of buffers; use a second engine for interleaved graphs)")
        return eng.backward_image(grads), None


class YoloTrainPairEngine:
    """The detector of one RL iteration (train.py:262-271): the detection loss of the INPUT batch (a constant) and of the
    RETOUCHED batch (differentiable) — two forwards of B images in the reference. Here ONE forward of 2B images (the deep
    layers of a 512 x 512 input are 16 x 16 maps: at B = 8 their launches cannot fill 256 CUs; measured 2.90 ms against
    2 x 2.02 at 8 x 512 x 512) and the backward over the second half only: a B-image engine whose activation buffers ARE
    images [B, 2B) of the 2B engine's reads the pre-activations that forward kept. Same kernels, same per-image arithmetic
    as two YoloTrainEngine passes (a conv output depends on its own image only); what may differ is the kernel variant the
    tuning table holds for the 2B shapes, i.e. bf16 rounding of the forward."""

    def __init__(self, model, batch, height, width, device="cuda:0"):
        import os
        self.B = int(batch)
        self.full = YoloTrainEngine(model, 2 * self.B, height, width, device, capture=("fwd",))
        self.half = YoloTrainEngine(model, self.B, height, width, device, share_with=self.full, first_image=self.B,
                                    capture=("bwd",))
        ws = self.full._loss_workspace()
        B = self.B
        self.half._loss_ws = dict(tobj=[t[B:] for t in ws["tobj"]], cnt=[t[B:] for t in ws["cnt"]],
                                  part=[t[B:] for t in ws["part"]], ticket=ws["ticket"][B:])
        self.dev, self.H, self.W = self.full.dev, self.full.H, self.full.W
        # The SHALLOW layers per half (see begin_input_half): an engine over images [0, B) of the 2B engine's buffers for the
        # input batch; `half` (images [B, 2B)) runs the retouched batch's. ADAYOLO_TRAIN_EARLY = the number of stride-2 convs
        # inside the shallow part (default 2: everything in front of 128 -> 256 s2 — measured best of 0..4, tools/train_graph_ab.py --early), 0 = one 2B-image forward as before.
        self.early_cut = int(os.environ.get("ADAYOLO_TRAIN_EARLY", "2"))
        # (the ordinary loop walks the same three pieces launch by launch — ~75 launches where the one-piece forward is one replay
        # of the engine's own graph, 0.3 ms more host work per iteration: both loops make the same launches, so a replayed
        # iteration stays the ordinary iteration bit for bit)
        self.first = None
        if self.early_cut > 0:
            self.first = YoloTrainEngine(model, self.B, height, width, device, share_with=self.full, first_image=0, capture=())
        self._early, self._split = None, None

    def autotune(self, cache=None, write=True, **kw):
        self.full.autotune(cache=cache, write=write, **kw)
        if self.first is not None:
            self.first.autotune(cache=cache, write=write, **kw)
        self._split = None
        return self.half.autotune(cache=cache, write=write, **kw)

    # ---- the input batch's shallow layers beside the agent ------------------------------------------------------------
    # One 2B-image forward is cheaper than two B-image ones because the DEEP layers (32 x 32 and 16 x 16 maps at 512 x 512) cannot
    # fill 256 CUs with B images. The shallow layers can (8 x 256 x 256 ... 8 x 64 x 64 pixels), and the input batch's need
    # nothing the agent computes: they run on a second stream beside the agent's forward — a latency chain of small launches
    # that leaves the chip idle for ~0.9 ms — so that after the filters only the RETOUCHED half's shallow layers (B images)
    # and the deep layers (2B images) are left on the critical path: at 8 x 512 x 512 the 3.04 ms of the 16-image forward become
    # 0.83 (hidden) + 0.83 + 1.61 (tools/train_det_breakdown.py, PLAN_ORDER=1). Same kernels per image; the B-image shapes
    # take their own rows of the tuning table (bf16 rounding of a layer may differ from the 2B-image kernel's, as between
    # any two tuned shapes).
    @staticmethod
    def _cut_index(eng, ndown):
        """Index into eng._forward_plan() of the launch that runs the (ndown + 1)-th stride-2 conv."""
        plan, seen = eng._forward_plan(), 0
        for j, (kind, _, a) in enumerate(plan):
            if kind in ("conv", "convkeep"):
                stride = a[14] if kind == "conv" else a[16]
                if stride == 2:
                    if seen == ndown:
                        return j
                    seen += 1
        raise _lib.AdayoloError(f"YoloTrainPairEngine: the detector has fewer than {ndown + 1} stride-2 convs")

    def _plans_split(self):
        sig = tuple(id(e._forward_plan()) for e in (self.first, self.half, self.full))      # (a re-tuned engine rebuilds its plan)
        if self._split is None or self._split[0] != sig:
            cut = {id(e): self._cut_index(e, self.early_cut) for e in (self.first, self.half, self.full)}
            self._split = (sig, self.first._forward_plan()[:cut[id(self.first)]], self.half._forward_plan()[:cut[id(self.half)]],
                           self.full._forward_plan()[cut[id(self.full)]:])
        return self._split[1:]

    def _early_now(self):
        return self.first is not None

    def prepare_capture(self):
        """Build every launch plan a captured iteration will walk (building probes kernels: not inside the capture)."""
        if self.first is not None:
            self._plans_split()

    def begin_input_half(self, imgs):
        """Enqueue the shallow layers of the INPUT batch on the current stream (the caller's side stream) and record the event
        the deep layers wait for. Optional: per_sample_loss_pair runs them itself when this was not called for `imgs`."""
        if not self._early_now():
            return
        if imgs.shape != (self.B, 3, self.H, self.W) or imgs.dtype != torch.float32 or imgs.device != self.dev:
            raise ValueError(f"expected fp32 {(self.B, 3, self.H, self.W)} on {self.dev}")
        imgs = imgs.detach().contiguous()
        pre = self._plans_split()[0]
        with torch.cuda.device(self.dev):
            self.first._run(pre, img=imgs)
            ev = torch.cuda.Event()
            ev.record()
        self._early = (imgs.data_ptr(), imgs._version, ev, imgs)

    def head_shapes(self):
        return self.half.head_shapes()

    def per_sample_loss_pair(self, loss_fn, imgs, retouch, packed, packed_pair):
        """(loss of `imgs` [B,1], a constant; loss of `retouch` [B,1] with autograd to `retouch`). `packed`: the target
        assignment of the B label sets, `packed_pair` the same for [labels; labels] (loss.assign_labels_packed(pair=True))."""
        for t in (imgs, retouch):
            if t.shape != (self.B, 3, self.H, self.W) or t.dtype != torch.float32 or t.device != self.dev:
                raise ValueError(f"expected fp32 {(self.B, 3, self.H, self.W)} on {self.dev}")
        if torch.is_grad_enabled() and retouch.requires_grad:
            return _PairLossFn.apply(retouch, imgs, self, loss_fn, packed, packed_pair)
        both = self._forward_losses(loss_fn, imgs, retouch, packed_pair)
        return both[:self.B], both[self.B:]

    def _forward_losses(self, loss_fn, imgs, retouch, packed_pair):
        full = self.full
        if loss_fn.nc + 5 != full.no or len(packed_pair) != len(full.raw) or loss_fn.hyp.get("fl_gamma", 0.0) != 0.0:
            raise ValueError("loss / detector mismatch (classes, layers) or focal loss requested: use the PyTorch loss")
        loss = torch.empty((2 * self.B,), dtype=torch.float32, device=self.dev)
        if not self._early_now():
            full._forward_raw_halves(imgs.detach().contiguous(), retouch.detach().contiguous())
        else:
            imgs_c, ret_c = imgs.detach().contiguous(), retouch.detach().contiguous()
            pre_first, pre_half, deep = self._plans_split()
            early, self._early = self._early, None
            with torch.cuda.device(self.dev):
                full._pass_begin()
                try:
                    if early is not None and early[0] == imgs_c.data_ptr() and early[1] == imgs_c._version:
                        torch.cuda.current_stream().wait_event(early[2])     # the input half's shallow layers, enqueued earlier
                    else:
                        self.first._run(pre_first, img=imgs_c)
                    self.half._run(pre_half, img=ret_c)
                    full._run(deep)
                finally:
                    full._pass_end()
            full._gen += 1
        self.half._gen += 1
        with torch.cuda.device(self.dev):
            a, keep = full._loss_args(loss_fn, packed_pair)
            a.loss = loss.data_ptr()
            _lib.check(full.L.adayolo_detloss_fwd(ctypes.byref(a), _lib.stream_ptr()), "adayolo_detloss_fwd")
        return loss.view(2 * self.B, 1)


class _PairLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, retouch, imgs, pair, loss_fn, packed, packed_pair):
        both = pair._forward_losses(loss_fn, imgs, retouch, packed_pair)
        ctx.pair, ctx.gen, ctx.loss_fn, ctx.packed = pair, pair.half._gen, loss_fn, packed
        l_in, l_re = both[:pair.B], both[pair.B:]
        ctx.mark_non_differentiable(l_in)
        return l_in, l_re

    @staticmethod
    def backward(ctx, _g_in, grad_loss):
        eng = ctx.pair.half
        if ctx.gen != eng._gen:
            raise RuntimeError("YoloTrainPairEngine: backward after a newer forward overwrote the saved pre-activations")
        g = grad_loss.reshape(eng.B).float().contiguous()
        with torch.cuda.device(eng.dev):
            a, keep = eng._loss_args(ctx.loss_fn, ctx.packed)
            scratch = torch.empty((eng.B,), dtype=torch.float32, device=eng.dev)
            a.loss, a.grad_loss = scratch.data_ptr(), g.data_ptr()
            _lib.check(eng.L.adayolo_detloss_bwd(ctypes.byref(a), _lib.stream_ptr()), "adayolo_detloss_bwd")
        return eng._backward_raw(), None, None, None, None, None
