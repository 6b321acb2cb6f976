#!/usr/bin/env python3
"""Golden-vector generator: imports the REFERENCE (OpenImagingLab/AdaptiveISP at /root/reference)
unmodified and records inputs -> outputs of the ISP hot path as small .npz fixtures.

Runs ONLY in the build container (the reference never travels to the GPU box); the fixtures it
writes are committed next to it. Absent third-party modules are stubbed: cv2,
easydict, skimage are used only by visualisation/demo code; torchvision's `torch_pad` IS
torch.nn.functional.pad in the pinned torchvision 0.15.2.

    python tests/golden/gen_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = HERE                                   # --out DIR: write the fixtures somewhere else (tools/regen_check.sh)
sys.path.insert(0, os.path.dirname(HERE))
from _synth import synth_state_dict, synth_yolo_state_dict, test_image  # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference(root="/root/reference"):
    _stub("cv2")
    _stub("easydict", EasyDict=dict)
    sk = _stub("skimage"); sk.io = _stub("skimage.io")
    tv = _stub("torchvision"); tv.transforms = _stub("torchvision.transforms")
    tv.transforms.functional_tensor = _stub("torchvision.transforms.functional_tensor",
                                            torch_pad=torch.nn.functional.pad)
    import matplotlib
    matplotlib.use("Agg")
    sys.path.insert(0, root)
    from isp import filters
    from config import cfg
    import agent
    import value
    return filters, cfg, agent, value


class _AnyObj:
    def __init__(self, *a, **k): pass
    def __call__(self, *a, **k): return _AnyObj()
    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        return _AnyObj()
    def __mro_entries__(self, bases): return (object,)


class _AnyModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _AnyObj()


def import_reference_yolo(root="/root/reference/yolov3"):
    """The detector subtree imports plotting / dataset / export helpers at module scope; none of them is
    on the forward path, so absent packages become permissive stubs."""
    for n in ["cv2", "seaborn", "thop", "ultralytics", "ultralytics.utils", "ultralytics.utils.plotting",
              "ultralytics.utils.checks", "torchvision", "torchvision.ops", "torchvision.datasets",
              "torchvision.transforms", "torchvision.transforms.functional", "IPython", "IPython.display"]:
        m = _AnyModule(n)
        m.__path__ = []
        sys.modules[n] = m
    for n in [k for k in sys.modules if k == "utils" or k.startswith("utils.") or k == "models" or k.startswith("models.")]:
        del sys.modules[n]
    sys.path.insert(0, root)
    from models.yolo import Model
    return Model


def gen_yolo():
    import io, contextlib, json
    Model = import_reference_yolo()
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        m = Model("/root/reference/yolov3/models/yolov3.yaml", ch=3, nc=80)
    m.load_state_dict(synth_yolo_state_dict(m, seed=2))
    m.eval()
    x = test_image(1, 64, 96, seed=51, special=False)
    with torch.no_grad():
        pred, raws = m(torch.from_numpy(x))
    out = {"x": x, "pred": pred.numpy()}
    for i, r in enumerate(raws):
        out[f"raw{i}"] = r.numpy()
    np.savez_compressed(os.path.join(OUT, "yolo.npz"), **out)
    # detection loss (reward signal): the reference's ComputeLossBatch on random head maps + hand-made targets
    from utils.loss import ComputeLossBatch
    m.hyp = dict(box=0.05, cls=0.5, obj=1.0 * (96 / 640) ** 2, anchor_t=4.0, cls_pw=1.0, obj_pw=1.0, fl_gamma=0.0,
                 label_smoothing=0.0)
    m.train()
    crit = ComputeLossBatch(m)
    g = torch.Generator().manual_seed(9)
    preds = [torch.randn(2, 3, 8, 12, 85, generator=g), torch.randn(2, 3, 4, 6, 85, generator=g),
             torch.randn(2, 3, 2, 3, 85, generator=g)]
    targets = torch.tensor([[0, 3, 0.30, 0.40, 0.20, 0.30], [0, 17, 0.70, 0.55, 0.50, 0.60], [1, 0, 0.52, 0.48, 0.10, 0.15],
                            [1, 79, 0.15, 0.85, 0.25, 0.20], [1, 5, 0.9, 0.1, 0.6, 0.9]])
    lbox, lobj, lcls = crit([p.clone() for p in preds], targets.clone())
    lo = {f"p{i}": p.numpy() for i, p in enumerate(preds)}
    lo.update(targets=targets.numpy(), lbox=lbox.numpy(), lobj=lobj.numpy(), lcls=lcls.numpy(),
              anchors=m.model[-1].anchors.numpy())
    for b in range(2):          # per-sample scoring as in train.py:184-196
        tb = targets[targets[:, 0] == b].clone(); tb[:, 0] = 0
        l3 = crit([p[b:b + 1].clone() for p in preds], tb)
        lo[f"sample{b}"] = torch.cat(l3).numpy()
    # the same scoring on head maps that are exactly representable in bf16 (what the HIP loss kernels read from the
    # detector's bf16 buffers), with autograd's gradient of sum_b w_b * loss_b w.r.t. the maps: the direct pin of
    # adayolo_detloss_fwd / _bwd (tests/test_gpu_yolo_train.py)
    qs = [p.to(torch.bfloat16).float().requires_grad_(True) for p in preds]
    wq = torch.tensor([0.7, 1.3])
    total = 0.0
    for b in range(2):
        tb = targets[targets[:, 0] == b].clone(); tb[:, 0] = 0
        l3 = crit([q[b:b + 1] for q in qs], tb)
        lo[f"qsample{b}"] = torch.cat(l3).detach().numpy()
        total = total + wq[b] * torch.cat(l3).sum()
    total.backward()
    lo["qweights"] = wq.numpy()
    for i, q in enumerate(qs):
        lo[f"q{i}"], lo[f"qgrad{i}"] = q.detach().numpy(), q.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "detloss.npz"), **lo)
    m.eval()
    p = os.path.join(OUT, "state_dict_keys.json")
    keys = json.load(open(p))
    keys["yolo"] = {k: list(v.shape) for k, v in m.state_dict().items()}
    json.dump(keys, open(p, "w"), indent=0)


def gen_core():
    """filters / filters_grad / nlm / pool64 / agent / select fixtures + state_dict_keys.json (agent, value)."""
    import warnings
    warnings.filterwarnings("ignore")
    torch.set_num_threads(4)
    filters, cfg, agent_mod, value_mod = import_reference()
    T = torch.from_numpy

    # ---------------------------------------------------------------- A. every filter: process + forward
    out = {}
    img = test_image(2, 24, 40, seed=11)
    out["img"] = img
    classes = [
        ("E", filters.ExposureFilter), ("G", filters.GammaFilter), ("CCM", filters.CCMFilter),
        ("Shr", filters.SharpenFilter), ("NLM", filters.DenoiseFilter), ("T", filters.ToneFilter),
        ("Ct", filters.ContrastFilter), ("S+", filters.SaturationPlusFilter), ("BW", filters.WNBFilter),
        ("W", filters.ImprovedWhiteBalanceFilter), ("USM", filters.SharpenUSMFilter),
        ("ShrV2", filters.SharpenFilterV2), ("C", filters.ColorFilter),
    ]
    rng = np.random.default_rng(5)
    with torch.no_grad():
        for name, cls in classes:
            f = cls(cfg, predict=False)
            n = f.get_num_filter_parameters()
            feat = rng.normal(0.0, 1.2, (2, n)).astype(np.float32)
            if name == "USM":
                feat[:, 0] = np.abs(feat[:, 0]) * 0.5 + 0.2      # keep sigma away from 0 (NaN kernel)
            param = f.filter_param_regressor(T(feat))
            proc = f.process(T(img), param)
            fwd, _, _ = f.forward(T(img), specified_parameter=param)
            key = name.replace("+", "p")
            out[f"{key}.feat"] = feat
            out[f"{key}.param"] = param.reshape(2, -1).numpy()
            out[f"{key}.process"] = proc.numpy()
            out[f"{key}.forward"] = fwd.numpy()
    np.savez_compressed(os.path.join(OUT, "filters.npz"), **out)

    # ---------------------------------------------------------------- A2. parameter gradients (autograd of the reference)
    gout = {}
    G = np.random.default_rng(6).normal(0.0, 1.0, img.shape).astype(np.float32)
    gout["grad_out"] = G
    for name, cls in classes:
        key = name.replace("+", "p")
        f = cls(cfg, predict=False)
        param = T(out[f"{key}.param"]).clone()
        if name in ("T", "C"):
            param = f.filter_param_regressor(T(out[f"{key}.feat"])).detach()
        param.requires_grad_(True)
        for mode in ("process", "forward"):
            if param.grad is not None:
                param.grad = None
            y = f.process(T(img), param)
            if mode == "forward":
                y = torch.clip(y, 0.0, 1.0)
            (y * T(G)).sum().backward()
            gout[f"{key}.{mode}"] = param.grad.reshape(2, -1).numpy().copy()
    np.savez_compressed(os.path.join(OUT, "filters_grad.npz"), **gout)

    # ---------------------------------------------------------------- B. NLM wrap-around cases
    out = {}
    nlm = filters.DenoiseFilter(cfg, predict=False)
    with torch.no_grad():
        for tag, shape, hs, seed in (("a", (2, 20, 28), [0.08, 0.5], 21), ("tiny", (1, 6, 9), [0.3], 22),
                                     ("odd", (1, 37, 70), [0.02], 23)):
            x = test_image(shape[0], shape[1], shape[2], seed=seed, special=False)
            x += np.random.default_rng(seed).normal(0, 0.02, x.shape).astype(np.float32)
            h = np.asarray(hs, np.float32).reshape(-1, 1)
            out[f"{tag}.img"] = x
            out[f"{tag}.h"] = h
            out[f"{tag}.out"] = nlm.process(T(x), T(h)).numpy()
    np.savez_compressed(os.path.join(OUT, "nlm.npz"), **out)

    # ---------------------------------------------------------------- C. adaptive 64x64 pooling
    out = {}
    pool = torch.nn.AdaptiveAvgPool2d((64, 64))
    for tag, (B, H, W) in (("a", (1, 72, 100)), ("small", (2, 30, 50)), ("exact", (1, 128, 64)), ("hd", (1, 90, 160))):
        x = test_image(B, H, W, seed=31 + H, special=False)
        out[f"{tag}.img"] = x
        out[f"{tag}.out"] = pool(T(x)).numpy()
    np.savez_compressed(os.path.join(OUT, "pool64.npz"), **out)

    # ---------------------------------------------------------------- D. Agent.forward (eval), E. Value.forward
    out = {}
    ag = agent_mod.Agent(cfg, shape=(6 + len(cfg.filters), 64, 64), device="cpu")
    ag.load_state_dict(synth_state_dict(ag, seed=0))
    ag.eval()
    va = value_mod.Value(cfg, shape=(9 + len(cfg.filters), 64, 64))
    va.load_state_dict(synth_state_dict(va, seed=1))
    va.eval()
    x = test_image(2, 32, 48, seed=41)
    z = np.random.default_rng(42).random((2, cfg.z_dim)).astype(np.float32)
    s0 = np.zeros((2, cfg.num_state_dim), np.float32)
    s1 = s0.copy(); s1[:, 2] = 4.0; s1[0, 3 + 5] = 1.0; s1[1, 3 + 2] = 1.0     # step 4 (-> last), some usage
    out["x"], out["z"], out["s0"], out["s1"] = x, z, s0, s1
    cap = {}
    ag.fc2.register_forward_hook(lambda m, i, o: cap.__setitem__("logits", o.detach().clone()))
    ag.feature_extractor.register_forward_hook(lambda m, i, o: cap.__setitem__("features", o.detach().clone()))
    with torch.no_grad():
        for tag, st, prog in (("s0", s0, 1.0), ("s1", s1, 0.25)):
            (xo, ns, sur, pen), dbg, _ = ag((T(x), T(z), T(st)), prog)
            out[f"{tag}.progress"] = np.float32(prog)
            out[f"{tag}.x"] = xo.numpy(); out[f"{tag}.new_states"] = ns.numpy()
            out[f"{tag}.surrogate"] = sur.numpy(); out[f"{tag}.penalty"] = pen.numpy()
            out[f"{tag}.selected"] = dbg["selected_filter"].numpy().astype(np.int64)
            out[f"{tag}.pdf0"] = dbg["pdf"].numpy()
            out[f"{tag}.logits"] = cap["logits"].numpy()
            out[f"{tag}.features"] = cap["features"].numpy()
        # teacher-forced selection of every filter (same params, progress 1.0, states s0)
        for k in range(len(cfg.filters)):
            (xo, ns, sur, pen), dbg, _ = ag((T(x), T(z), T(s0)), 1.0, selected_filter_id=k)
            out[f"forced{k}.x"] = xo.numpy()
            out[f"forced{k}.new_states"] = ns.numpy()
            out[f"forced{k}.penalty"] = pen.numpy()
            out[f"forced{k}.param0"] = dbg["filter_debug_info"][k]["filter_parameters"].reshape(-1).numpy()
        # high-res path: params from the low-res image applied to a second tensor (agent.py:155-157)
        xh = test_image(2, 40, 72, seed=43)
        (xo, ns, hro), dbg, _ = ag((T(x), T(z), T(s0)), 1.0, high_res=T(xh), selected_filter_id=5)
        out["hr.in"], out["hr.x"], out["hr.out"] = xh, xo.numpy(), hro.numpy()
        # 5-step teacher-forced trajectory, schedule S_mixed = [E, CCM, NLM, Shr, T]
        xt, st = T(x), T(s0)
        for step, k in enumerate([0, 2, 4, 3, 5]):
            (xt, st, sur, pen), dbg, _ = ag((xt, T(z), st), 1.0, selected_filter_id=k)
            out[f"traj{step}.x"] = xt.numpy(); out[f"traj{step}.states"] = st.numpy()
            out[f"traj{step}.penalty"] = pen.numpy()
        out["value.s0"] = va(T(x), T(s0)).numpy()
        out["value.s1"] = va(T(x), T(s1)).numpy()
        out["value.none"] = value_mod.Value(cfg, shape=(6, 64, 64)).eval()(T(x)).numpy() * 0  # shape check only
    np.savez_compressed(os.path.join(OUT, "agent.npz"), **out)
    import json
    with open(os.path.join(OUT, "state_dict_keys.json"), "w") as f:
        json.dump({"agent": {k: list(v.shape) for k, v in ag.state_dict().items()},
                   "value": {k: list(v.shape) for k, v in va.state_dict().items()}}, f, indent=0)

    # ---------------------------------------------------------------- F. integer stages: pdf_sample / one_hot
    out = {}
    r = np.random.default_rng(7)
    pdf = r.random((24, 10)).astype(np.float32) ** 3
    pdf /= pdf.sum(1, keepdims=True)
    u = r.random((24, 1)).astype(np.float32)
    u[0, 0] = 0.0                                  # -> index -1 -> all-zero one-hot
    u[1, 0] = np.nextafter(np.float32(1.0), np.float32(0.0))
    u[2, 0] = pdf[2, 0]                            # exactly on a cdf boundary
    idx = agent_mod.pdf_sample(T(pdf), T(u))
    out["pdf"], out["u"], out["idx"] = pdf, u, idx.numpy().astype(np.int64)
    out["one_hot"] = agent_mod.one_hot(10, idx.to(torch.int64)).numpy()
    np.savez_compressed(os.path.join(OUT, "select.npz"), **out)



def gen_nlm_general():
    """NonLocalMeansGray(search, patch) of the reference (isp/denoise.py:93-119) for window sizes other than the ISP's 11 / 5:
    the class default 21 / 7 and small ones, on images inside AND outside [0, 1] (only the luminance is clipped by the class)."""
    import_reference()
    import torch
    from isp import denoise
    out = {}
    cases = [("s7p3", 7, 3, (2, 24, 31), [0.1, 0.6], 31, False), ("s21p7", 21, 7, (1, 26, 33), [0.25], 32, False),
             ("s5p5", 5, 5, (1, 9, 12), [0.4], 33, False), ("s3p1", 3, 1, (1, 7, 8), [0.05], 34, False),
             ("s9p3_out_of_range", 9, 3, (1, 20, 22), [0.3], 35, True), ("s11p5", 11, 5, (1, 18, 25), [0.2], 36, False)]
    with torch.no_grad():
        for tag, search, patch, shape, hs, seed, wide in cases:
            x = test_image(shape[0], shape[1], shape[2], seed=seed, special=False)
            x += np.random.default_rng(seed).normal(0, 0.02, x.shape).astype(np.float32)
            if wide:
                x = (x * 3.0 - 0.4).astype(np.float32)
            else:
                x = np.clip(x, 0.0, 1.0).astype(np.float32)
            h = np.asarray(hs, np.float32).reshape(-1, 1, 1, 1)
            out[f"{tag}.img"], out[f"{tag}.h"] = x, h.reshape(-1)
            out[f"{tag}.sizes"] = np.asarray([search, patch], np.int32)
            out[f"{tag}.out"] = denoise.NonLocalMeansGray(search, patch)(torch.from_numpy(x), torch.from_numpy(h)).numpy()
    np.savez_compressed(os.path.join(OUT, "nlm_general.npz"), **out)


def gen_value_path():
    """The critic-to-actor gradient (train.py:281-305 with cfg.use_TD): L = -mean(V(retouch, new_states)) where retouch
    comes out of Agent.forward — the reference back-propagates through AdaptiveAvgPool2d and the selected filter into
    that filter's heads. Eval-mode modules (no dropout, BN running statistics) with autograd on; teacher-forced so every
    filter is covered. Records the gradient w.r.t. the selected filter's fc_filter weight/bias and agent.fc? none."""
    filters, cfg, agent_mod, value_mod = import_reference()
    T = torch.from_numpy
    ag = agent_mod.Agent(cfg, shape=(6 + len(cfg.filters), 64, 64), device="cpu")
    ag.load_state_dict(synth_state_dict(ag, seed=0))
    ag.eval()
    va = value_mod.Value(cfg, shape=(9 + len(cfg.filters), 64, 64))
    va.load_state_dict(synth_state_dict(va, seed=1))
    va.eval()
    x = test_image(2, 72, 88, seed=41, special=False)
    z = np.random.default_rng(42).random((2, cfg.z_dim)).astype(np.float32)
    s0 = np.zeros((2, cfg.num_state_dim), np.float32)
    out = {"x": x, "z": z, "s0": s0}
    for k, flt in enumerate(ag.filters):
        ag.zero_grad(set_to_none=True)
        (xo, ns, sur, pen), dbg, _ = ag((T(x), T(z), T(s0)), 1.0, selected_filter_id=k)
        v = va(xo, ns)
        (-v.mean()).backward()
        out[f"f{k}.value"] = v.detach().numpy()
        out[f"f{k}.gw"] = flt.fc_filter.weight.grad.numpy().copy()
        out[f"f{k}.gb"] = flt.fc_filter.bias.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "value_path.npz"), **out)
    print("value_path.npz:", {k: (v.shape, float(np.abs(v).max())) for k, v in out.items() if k.endswith(".gb")})


def _greedy_nms(boxes, scores, iou_thres):
    """Stand-in for torchvision.ops.nms while generating the eval-harness fixture (torchvision is not installed):
    its documented algorithm in plain Python, fp32 arithmetic. Only the suppression core comes from here; candidate
    selection, multi-label expansion, class offsets and max_det are the reference's own code."""
    order = torch.argsort(scores, descending=True, stable=True)
    b = boxes[order].float()
    n = b.shape[0]
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    removed = torch.zeros(n, dtype=torch.bool)
    keep = []
    for i in range(n):
        if removed[i]:
            continue
        keep.append(i)
        w = (torch.minimum(b[i, 2], b[i + 1:, 2]) - torch.maximum(b[i, 0], b[i + 1:, 0])).clamp(min=0)
        h = (torch.minimum(b[i, 3], b[i + 1:, 3]) - torch.maximum(b[i, 1], b[i + 1:, 1])).clamp(min=0)
        inter = w * h
        removed[i + 1:] |= inter / (area[i] + area[i + 1:] - inter) > iou_thres
    return order[torch.tensor(keep, dtype=torch.long)]


def gen_eval():
    """Eval-harness fixture: the reference's non_max_suppression / box helpers / process_batch / ap_per_class on
    seeded synthetic predictions (yolov3/utils/general.py:732-966, utils/metrics.py:31-123,262-280, val.py
    process_batch == val_adaptiveisp.py:79-103)."""
    import importlib, io, contextlib
    import_reference_yolo()
    gen = importlib.import_module("utils.general")
    met = importlib.import_module("utils.metrics")
    with contextlib.redirect_stdout(io.StringIO()):
        val = importlib.import_module("val")
    sys.modules["torchvision.ops"].nms = _greedy_nms
    gen.torchvision.ops = sys.modules["torchvision.ops"]
    # the reference aborts NMS after a wall-clock limit (general.py:891,962); the slow Python stand-in would trip it
    gen.time = types.SimpleNamespace(time=lambda: 0.0)
    rng = np.random.default_rng(77)
    out = {}
    # predictions: 2 images x 400 candidates x (5 + 6 classes), clustered boxes so that NMS has work to do
    B, N, nc = 2, 400, 6
    centers = rng.uniform(40, 280, (B, 12, 2))
    pred = np.zeros((B, N, 5 + nc), np.float32)
    for b in range(B):
        k = rng.integers(0, 12, N)
        pred[b, :, 0:2] = centers[b, k] + rng.normal(0, 4, (N, 2))
        pred[b, :, 2:4] = rng.uniform(20, 90, (N, 2))
        pred[b, :, 4] = rng.random(N) ** 2
        pred[b, :, 5:] = rng.random((N, nc)) ** 3
    out["pred"] = pred
    for tag, kw in (("ml", dict(conf_thres=0.05, iou_thres=0.6, multi_label=True, max_det=300)),
                    ("best", dict(conf_thres=0.25, iou_thres=0.45, multi_label=False, max_det=50)),
                    ("agn", dict(conf_thres=0.1, iou_thres=0.5, multi_label=True, agnostic=True, max_det=20)),
                    ("cls", dict(conf_thres=0.1, iou_thres=0.5, multi_label=False, classes=[1, 4], max_det=300))):
        res = gen.non_max_suppression(torch.from_numpy(pred.copy()), **kw)
        for b, r in enumerate(res):
            out[f"nms.{tag}.{b}"] = r.numpy()
    # box helpers
    boxes = rng.uniform(-20, 700, (50, 4)).astype(np.float32)
    out["boxes"] = boxes
    out["xywh2xyxy"] = gen.xywh2xyxy(torch.from_numpy(boxes.copy())).numpy()
    out["xyxy2xywh"] = gen.xyxy2xywh(torch.from_numpy(boxes.copy())).numpy()
    out["scale_auto"] = gen.scale_boxes((512, 512), torch.from_numpy(boxes.copy()), (375, 500)).numpy()
    out["scale_ratio_pad"] = gen.scale_boxes((512, 512), torch.from_numpy(boxes.copy()), (375, 500),
                                             ((1.024, 1.024), (0.0, 64.0))).numpy()
    # matching + AP: 300 detections over 5 classes against 60 labels
    det = np.zeros((300, 6), np.float32)
    lab = np.zeros((60, 5), np.float32)
    lab[:, 0] = rng.integers(0, 5, 60)
    lab[:, 1:3] = rng.uniform(0, 400, (60, 2)); lab[:, 3:5] = lab[:, 1:3] + rng.uniform(20, 120, (60, 2))
    src = rng.integers(0, 60, 300)
    det[:, :4] = lab[src, 1:] + rng.normal(0, 6, (300, 4)) * (rng.random((300, 1)) < 0.7)
    det[:, :4] += rng.uniform(-80, 80, (300, 4)) * (rng.random((300, 1)) < 0.25)
    det[:, 4] = rng.random(300)
    det[:, 5] = np.where(rng.random(300) < 0.8, lab[src, 0], rng.integers(0, 5, 300))
    iouv = torch.linspace(0.5, 0.95, 10)
    correct = val.process_batch(torch.from_numpy(det), torch.from_numpy(lab), iouv)
    out["det"], out["lab"], out["correct"] = det, lab, correct.numpy()
    out["iou"] = met.box_iou(torch.from_numpy(lab[:, 1:]), torch.from_numpy(det[:, :4])).numpy()
    tp, fp, p, r, f1, ap, cls = met.ap_per_class(correct.numpy(), det[:, 4], det[:, 5], lab[:, 0], names={})
    out.update(ap_tp=tp, ap_fp=fp, ap_p=p, ap_r=r, ap_f1=f1, ap_ap=ap, ap_cls=cls)
    rec = np.sort(rng.random(40)); prec = np.sort(rng.random(40))[::-1].copy()
    a, mpre, mrec = met.compute_ap(rec, prec)
    out.update(cap_rec=rec, cap_prec=prec, cap_ap=np.float64(a), cap_mpre=mpre, cap_mrec=mrec)
    np.savez_compressed(os.path.join(OUT, "evalharness.npz"), **out)
    print("evalharness.npz written")


def gen_ckpt():
    """Detector-checkpoint fixture: a width-0.0625 YOLOv3 built by the REFERENCE's Model class and saved the way its
    trainer saves `yolov3.pt` (yolov3/train.py: {'epoch', 'best_fitness', 'model': deepcopy(model).half(), 'ema',
    'updates', 'optimizer', 'opt', 'date'}) — i.e. a pickled module whose classes are models.yolo.* / models.common.*.
    The .pt holds tensors and class NAMES only (no reference source). Golden output = that module, .float().eval()."""
    import io, contextlib, copy, re
    Model = import_reference_yolo()
    text = open("/root/reference/yolov3/models/yolov3.yaml").read()
    text = re.sub(r"width_multiple:\s*[0-9.]+", "width_multiple: 0.0625", text)
    os.makedirs("/tmp/adaisp_gen", exist_ok=True)
    ypath = "/tmp/adaisp_gen/yolov3_w0625.yaml"
    open(ypath, "w").write(text)
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        m = Model(ypath, ch=3, nc=7)
    m.load_state_dict(synth_yolo_state_dict(m, seed=4))
    m.names = {i: f"c{i}" for i in range(7)}
    ckpt = {"epoch": 3, "best_fitness": np.array([0.5]), "model": copy.deepcopy(m).half(), "ema": None, "updates": 0,
            "optimizer": None, "opt": {"weights": "yolov3.pt", "imgsz": 512}, "date": "2025-01-01T00:00:00"}
    path = os.path.join(OUT, "yolov3_w0625_refpickle.pt")
    torch.save(ckpt, path)
    ref = torch.load(path, map_location="cpu", weights_only=False)["model"].float().eval()
    x = test_image(1, 64, 96, seed=52, special=False)
    with torch.no_grad():
        pred, raws = ref(torch.from_numpy(x))
    fused = copy.deepcopy(ref).fuse().eval()
    fpath = os.path.join(OUT, "yolov3_w0625_refpickle_fused.pt")
    torch.save({"model": copy.deepcopy(fused).half()}, fpath)
    fused = torch.load(fpath, map_location="cpu", weights_only=False)["model"].float().eval()
    with torch.no_grad():
        pred_f, _ = fused(torch.from_numpy(x))
    np.savez_compressed(os.path.join(OUT, "ckpt_import.npz"), x=x, pred=pred.numpy(), raw0=raws[0].numpy(),
                        pred_fused_fp32=pred_f.numpy(), nparams=np.int64(sum(p.numel() for p in ref.parameters())))
    print("checkpoint fixtures written:", os.path.getsize(path), os.path.getsize(fpath))


def gen_replay():
    """Replay-memory fixture: the REFERENCE's ReplayMemory methods (replay_memory.py:120-221) driven by a fake dataset
    object and a scripted state update, with Python's `random` seeded — records the order in which records are drawn."""
    import random
    import_reference()
    import_reference_yolo()
    sys.path.insert(0, "/root/reference")
    import replay_memory as rm
    from config import cfg

    class FakeDataset:
        def __init__(self):
            self.n = 0

        def get_next_batch(self, bs):
            ims, lbs, paths, shapes = [], [], [], []
            for _ in range(bs):
                ims.append(np.full((3, 2, 2), self.n, np.float32))
                lbs.append(np.array([[0, self.n % 5, 0.5, 0.5, 0.2, 0.2]], np.float32))
                paths.append(f"img{self.n}")
                shapes.append(((2, 2), ((1.0, 1.0), (0.0, 0.0))))
                self.n += 1
            return ims, lbs, paths, shapes

    mem = rm.ReplayMemory.__new__(rm.ReplayMemory)
    mem.cfg, mem.dataset, mem.image_pool = cfg, FakeDataset(), []
    mem.target_pool_size, mem.batch_size, mem.fake_output = 16, 4, None
    random.seed(1234)
    mem.fill_pool()
    drawn, steps_after, pool_sizes = [], [], []
    script = np.random.default_rng(8)
    for it in range(40):
        ims, lbs, paths, shapes, states = mem.get_next_fake_batch(4)
        drawn.append([int(p[3:]) for p in paths])
        new_states = []
        for s in states:
            s = s.copy()
            s[2] += 1                                                        # step
            if script.random() < 0.25 or s[2] >= 5:
                s[1] = 1                                                     # stopped
            if script.random() < 0.2:
                s[2] = 9                                                     # over-long trajectory (> 7)
            new_states.append(s)
        steps_after.append([[float(s[1]), float(s[2])] for s in new_states])
        mem.replace_memory(mem.images_and_states_to_records([i + 100 for i in ims], lbs, paths, shapes, new_states))
        pool_sizes.append(len(mem.image_pool))
    np.savez_compressed(os.path.join(OUT, "replay.npz"), drawn=np.array(drawn), script=np.array(steps_after),
                        pool_sizes=np.array(pool_sizes), final_paths=np.array([int(r.path[3:]) for r in mem.image_pool]),
                        final_im=np.array([float(np.asarray(r.im).reshape(-1)[0]) for r in mem.image_pool]),
                        max_traj=np.int64(cfg.maximum_trajectory_length), keep_prob=np.float64(cfg.over_length_keep_prob))
    print("replay.npz written")


def gen_mosaic():
    """Bayer packing fixture: the reference's `mosaic` and `reconstruct_bayer` (isp/unprocess_np.py:82-128) on a seeded
    image — what the demosaic extension must be the inverse of at the sampled sites."""
    import_reference()
    from isp import unprocess_np as up
    rng = np.random.default_rng(12)
    img = rng.random((10, 14, 3)).astype(np.float32)                 # HWC, as unprocess uses
    packed = up.mosaic(img, "RGGB")
    out = {"img": img, "packed": packed}
    for pat in ("rggb", "bggr", "grbg", "gbrg"):
        out[f"plane_{pat}"] = up.reconstruct_bayer(packed, pat)
    np.savez_compressed(os.path.join(OUT, "mosaic.npz"), **out)
    print("mosaic.npz written")


def gen_midsize():
    """One MID-SIZE fixture per stencil stage: 1 x 3 x 96 x 160 — an interior larger than one NLM
    tile (60 x 24) and one conv strip, 1.5 x 2.5 px pool windows — through the reference's NLM, USM, Sharpen, SharpenV2 and
    AdaptiveAvgPool2d((64, 64)) of each result; plus a 180 x 160 image for the pooling alone (2.8-row overlapping windows)."""
    filters, cfg, _, _ = import_reference()
    T = torch.from_numpy
    out = {}
    img = test_image(1, 96, 160, seed=47)
    img += np.random.default_rng(47).normal(0, 0.01, img.shape).astype(np.float32)
    out["img"] = img
    pool = torch.nn.AdaptiveAvgPool2d((64, 64))
    cases = (("NLM", filters.DenoiseFilter, [[0.15]]), ("USM", filters.SharpenUSMFilter, [[1.1, 1.4]]),
             ("Shr", filters.SharpenFilter, [[3.5]]), ("ShrV2", filters.SharpenFilterV2, [[1.7]]))
    with torch.no_grad():
        for name, cls, p in cases:
            f = cls(cfg, predict=False)
            param = torch.tensor(p, dtype=torch.float32)
            fwd, _, _ = f.forward(T(img), specified_parameter=param)
            out[f"{name}.param"] = np.asarray(p, np.float32)
            out[f"{name}.forward"] = fwd.numpy()
            out[f"{name}.pooled"] = pool(fwd).numpy()
        x = test_image(1, 180, 160, seed=48, special=False)
        out["pool.img"] = x
        out["pool.out"] = pool(T(x)).numpy()
    np.savez_compressed(os.path.join(OUT, "midsize.npz"), **out)
    print("wrote midsize.npz")




def gen_td():
    """Reward / TD arithmetic of the RL iteration (train.py:264-305) — the statements of DynamicISP.train themselves, executed
    on seeded [B, 1] tensors.

    The statements are cut out of the reference's syntax tree at generation time (the two `torch.clip` weightings of the
    detection losses, then everything from `reward = ...` to `agent_loss = ...`) and run with a stub `self` (cfg / args /
    max_bri / a `value` that hands back the seeded critic outputs); nothing of the reference's text is stored — the fixture
    holds inputs, switch settings and the resulting reward, q_value, advantage, both losses and autograd's gradients of each
    loss w.r.t. the raw retouch detection loss, the penalty, the surrogate and both critic values."""
    import ast
    import types
    import_reference()
    import util as ref_util
    from config import cfg as ref_cfg
    src = open("/root/reference/train.py").read()
    tree = ast.parse(src)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "DynamicISP")
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "train")
    loop = next(n for n in ast.walk(fn) if isinstance(n, ast.For) and getattr(n.target, "id", "") == "iter")

    def assigns(stmt, name):
        return isinstance(stmt, ast.Assign) and any(getattr(t, "id", None) == name for t in stmt.targets)

    def is_clip(stmt, name):
        return (assigns(stmt, name) and isinstance(stmt.value, ast.Call) and
                ast.unparse(stmt.value.func) == "torch.clip")
    body = loop.body
    clips = [s for s in body if is_clip(s, "detect_input_loss") or is_clip(s, "detect_retouch_loss")]
    first = next(i for i, s in enumerate(body) if assigns(s, "reward"))
    last = next(i for i, s in enumerate(body) if assigns(s, "agent_loss"))
    assert len(clips) == 2 and first < last
    stmts = clips + body[first:last + 1]
    code = compile(ast.Module(body=stmts, type_ignores=[]), "<train.py:264-305>", "exec")
    print("gen_td: executing reference statements at lines", [s.lineno for s in stmts])

    B, F = 12, len(ref_cfg["filters"])
    out = {}
    T = torch.tensor
    case_no = 0
    for use_td in (True, False):
        for use_truncated in (True, False):
            for use_penalty in (True, False):
                g = torch.Generator().manual_seed(100 + case_no)
                r = lambda *sh: torch.rand(sh, generator=g)          # noqa: E731
                cfgd = types.SimpleNamespace(**{k: v for k, v in ref_cfg.items() if isinstance(k, str)})
                cfgd.use_TD, cfgd.use_penalty = use_td, use_penalty
                if case_no % 2:                                       # exercise the constants away from their defaults too
                    cfgd.all_reward, cfgd.detect_loss_weight, cfgd.discount_factor = 0.7, 1.5, 0.9
                    cfgd.critic_logit_multiplier, cfgd.parameter_lr_mul = 50.0, 0.5
                # states: [reward, stopped, step, usage...]; a stopped sample, steps on both sides of maximum_trajectory_length
                stopped_col = (r(B, 1) < 0.4).float()
                stopped_col[0, 0], stopped_col[1, 0] = 1.0, 0.0
                step_col = (r(B, 1) * 6).floor()
                step_col[2, 0] = float(cfgd.maximum_trajectory_length)          # == : not cleared (torch.gt)
                step_col[3, 0] = float(cfgd.maximum_trajectory_length) + 1.0    # > 7-style sample: cleared
                step_col[4, 0] = 8.0
                new_states = torch.cat([stopped_col, stopped_col, step_col, (r(B, F) < 0.3).float()], dim=1)
                means = torch.full((B,), 0.5)
                means[5], means[6], means[7], means[8] = 0.005, 0.95, 0.01, 0.9     # below / above / on both thresholds
                means[9:] = r(B - 9) * 0.8 + 0.05
                retouch = (means.view(B, 1, 1, 1) * torch.ones(B, 3, 4, 4)).contiguous()
                leaves = dict(l_re=r(B, 1) * 1.3 - 0.1, penalty=r(B, 1) * 0.2, surrogate=-r(B, 1) * 3.0,
                              old_value=r(B, 1) * 4.0 - 2.0, new_value=r(B, 1) * 4.0 - 2.0)
                leaves["l_re"][10, 0], leaves["l_re"][11, 0] = 1.7, -0.3           # both sides of the [0, 1] clip
                leaves = {k: v.requires_grad_(True) for k, v in leaves.items()}
                l_in = r(B, 1) * 1.3 - 0.1
                vals = iter([leaves["old_value"], leaves["new_value"]])
                me = types.SimpleNamespace(cfg=cfgd, args=types.SimpleNamespace(use_truncated=use_truncated), max_bri=0.9,
                                           value=lambda *a, **k: next(vals), device="cpu")
                env = dict(torch=torch, self=me, imgs=retouch, states=new_states, retouch=retouch, new_states=new_states,
                           stopped=new_states[:, ref_util.STATE_STOPPED_DIM:ref_util.STATE_STOPPED_DIM + 1],
                           penalty=leaves["penalty"], surrogate=leaves["surrogate"], detect_input_loss=l_in.clone(),
                           detect_retouch_loss=leaves["l_re"], STATE_STEP_DIM=ref_util.STATE_STEP_DIM,
                           STATE_STOPPED_DIM=ref_util.STATE_STOPPED_DIM, STATE_REWARD_DIM=ref_util.STATE_REWARD_DIM)
                exec(code, env)
                tag = f"c{case_no}"
                out[f"{tag}.switches"] = np.asarray([use_td, use_truncated, use_penalty], np.int64)
                out[f"{tag}.consts"] = np.asarray([cfgd.detect_loss_weight, cfgd.all_reward, cfgd.critic_logit_multiplier,
                                                   cfgd.discount_factor, cfgd.parameter_lr_mul,
                                                   cfgd.maximum_trajectory_length, 0.9], np.float64)
                out[f"{tag}.l_in"], out[f"{tag}.new_states"] = l_in.numpy(), new_states.numpy()
                out[f"{tag}.retouch_mean"] = torch.mean(retouch, dim=(1, 2, 3)).unsqueeze(-1).numpy()
                for k, v in leaves.items():
                    out[f"{tag}.{k}"] = v.detach().numpy().copy()
                for k in ("reward", "q_value", "advantage", "value_loss", "agent_loss"):
                    out[f"{tag}.out.{k}"] = env[k].detach().numpy().copy()
                for loss in ("value_loss", "agent_loss"):
                    grads = torch.autograd.grad(env[loss], list(leaves.values()), retain_graph=True, allow_unused=True)
                    for (k, v), gr in zip(leaves.items(), grads):
                        out[f"{tag}.d_{loss}.{k}"] = (torch.zeros_like(v) if gr is None else gr).numpy().copy()
                case_no += 1
    np.savez_compressed(os.path.join(OUT, "td.npz"), **out)
    print("td.npz written:", case_no, "switch settings")


GENERATORS = dict(core=gen_core, yolo=gen_yolo, eval=gen_eval, ckpt=gen_ckpt, replay=gen_replay, mosaic=gen_mosaic,
                  nlm_general=gen_nlm_general, value_path=gen_value_path, midsize=gen_midsize, td=gen_td)
_OLD_FLAGS = {"--eval-only": "eval", "--ckpt-only": "ckpt", "--replay-only": "replay", "--mosaic-only": "mosaic",
              "--midsize-only": "midsize", "--value-path-only": "value_path", "--nlm-general-only": "nlm_general"}


def main(argv):
    """`gen_golden.py` regenerates everything (each generator in its own process: the two reference trees both own a
    top-level `utils` / `models` name); `--only NAME[,NAME]` runs those in this process; `--out DIR` writes there."""
    import subprocess
    global OUT
    names = []
    i = 0
    while i < len(argv):
        a = argv[i]
        if a == "--out":
            OUT = os.path.abspath(argv[i + 1]); i += 1
            os.makedirs(OUT, exist_ok=True)
        elif a == "--only":
            names += argv[i + 1].split(","); i += 1
        elif a in _OLD_FLAGS:
            names.append(_OLD_FLAGS[a])
        else:
            raise SystemExit(f"unknown argument {a}; generators: {', '.join(GENERATORS)}")
        i += 1
    if names:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot  # noqa: F401
        for n in names:
            GENERATORS[n]()
        return
    for n in GENERATORS:                       # `core` first: it creates state_dict_keys.json, `yolo` adds to it
        subprocess.run([sys.executable, os.path.abspath(__file__), "--only", n, "--out", OUT], check=True)
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT) if f.endswith(".npz"))
    print("fixtures written, total bytes:", tot)


if __name__ == "__main__":
    main(sys.argv[1:])
