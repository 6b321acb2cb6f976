"""Detector module tree vs the reference (golden vectors from the reference's own DetectionModel)."""
import json
import os

import numpy as np
import pytest
import torch

from _synth import synth_yolo_state_dict
from adaptiveisp_amd.yolo.model import yolov3


def test_state_dict_layout_matches_reference():
    here = os.path.dirname(os.path.abspath(__file__))
    ref = json.load(open(os.path.join(here, "golden", "state_dict_keys.json")))["yolo"]
    m = yolov3()
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == ref
    assert sum(p.numel() for p in m.parameters()) == 61949149


def test_torch_forward_matches_reference(golden):
    g = golden("yolo")
    m = yolov3().eval()
    m.load_state_dict(synth_yolo_state_dict(m))
    with torch.no_grad():
        pred, raws = m(torch.from_numpy(g["x"]))
    np.testing.assert_allclose(pred.numpy(), g["pred"], rtol=1e-5, atol=1e-6)
    for i, r in enumerate(raws):
        np.testing.assert_allclose(r.numpy(), g[f"raw{i}"], rtol=1e-5, atol=1e-6)
    m.train()
    assert isinstance(m(torch.from_numpy(g["x"]).repeat(2, 1, 1, 1)), list)       # train mode returns the raw maps


def test_libadayolo_exports_header_symbols():
    import re
    from adaptiveisp_amd.yolo import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(root, "include", "adayolo.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(adayolo_\w+)\s*\(", text)))
    L = _lib.load()
    assert set(_lib.EXPORTS) == set(names)
    for n in names:
        assert hasattr(L, n)
    assert L.adayolo_abi_version() == _lib.ABI_VERSION
    assert L.adayolo_conv_fwd(None, 8, None, None, None, 0, None, 8, 1, 4, 4, 8, 8, 1, 1, 0, None) == -1


def test_batched_per_sample_loss_equals_the_loop(golden):
    """One batched pass == the reference's per-sample Python loop (train.py:184-196), incl. images without targets,
    and == the reference's own numbers (detloss.npz sample0/sample1)."""
    import torch
    from adaptiveisp_amd.yolo.loss import DetectionLoss, batched_per_sample_loss, per_sample_loss
    g = golden("detloss")
    hyp = dict(box=0.05, cls=0.5, obj=1.0 * (96 / 640) ** 2, anchor_t=4.0, cls_pw=1.0, obj_pw=1.0, fl_gamma=0.0,
               label_smoothing=0.0)
    crit = DetectionLoss(torch.from_numpy(g["anchors"]), nc=80, hyp=hyp)
    preds = [torch.from_numpy(g[f"p{i}"]) for i in range(3)]
    T = torch.from_numpy(g["targets"])
    labels = [T[T[:, 0] == b].clone() for b in range(2)]
    got = batched_per_sample_loss(crit, preds, labels)
    for b in range(2):
        assert abs(float(got[b]) - float(g[f"sample{b}"].sum())) < 2e-5 * max(1.0, abs(float(got[b])))
    # 4 images, one of them without any target, a duplicate-cell pair in another
    gen = torch.Generator().manual_seed(3)
    preds4 = [torch.randn(4, 3, 8, 12, 85, generator=gen), torch.randn(4, 3, 4, 6, 85, generator=gen),
              torch.randn(4, 3, 2, 3, 85, generator=gen)]
    labels4 = [labels[0], torch.zeros((0, 6)), labels[1], torch.tensor([[0, 7, 0.5, 0.5, 0.2, 0.2], [0, 7, 0.5, 0.5, 0.2, 0.2]])]
    a = batched_per_sample_loss(crit, preds4, labels4)
    b = per_sample_loss(crit, preds4, labels4)
    assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), (a, b)
    # gradients agree too
    for p in preds4:
        p.requires_grad_(True)
    ga = torch.autograd.grad(batched_per_sample_loss(crit, preds4, labels4).sum(), preds4)
    gb = torch.autograd.grad(per_sample_loss(crit, preds4, labels4).sum(), preds4)
    for x, y in zip(ga, gb):
        assert torch.allclose(x, y, rtol=1e-4, atol=1e-7)
    # the target assignment depends on labels and map shapes only: computed once (rl.train_iteration) and reused for a
    # second batch of predictions, it gives the same losses as assigning inside each call
    from adaptiveisp_amd.yolo.loss import assign_labels
    shared = assign_labels(crit, preds4, labels4)
    other = [torch.randn(p.shape, generator=gen) for p in preds4]
    for pr in (preds4, other):
        assert torch.equal(batched_per_sample_loss(crit, pr, labels4, shared), batched_per_sample_loss(crit, pr, labels4))


def test_pack_assigned_layout():
    """The form in which the target assignment reaches csrc/yolo_loss.hip: idx int32 [n,5] = (image, anchor, gj, gi,
    class), box fp32 [n,6] = (tx, ty, tw, th, anchor_w, anchor_h), one pair per head layer, rows in assignment order."""
    from adaptiveisp_amd.yolo import yolov3
    from adaptiveisp_amd.yolo.loss import DetectionLoss, assign_labels, default_hyp, pack_assigned
    det = yolov3()
    loss_fn = DetectionLoss(det.model[-1].anchors, nc=80, hyp=default_hyp(80, 128))
    import types
    shapes = [types.SimpleNamespace(shape=(2, 3, 16 // k, 16 // k, 85), device=torch.device("cpu")) for k in (1, 2, 4)]
    labels = [torch.tensor([[0, 5, 0.5, 0.5, 0.3, 0.4], [0, 9, 0.2, 0.7, 0.1, 0.2]]), torch.zeros(0, 6)]
    assigned = assign_labels(loss_fn, shapes, labels)
    packed = pack_assigned(assigned)
    assert len(packed) == 3
    for (idx, box), m, sh in zip(packed, assigned, shapes):
        n = m["b"].shape[0]
        assert idx.shape == (n, 5) and idx.dtype == torch.int32 and box.shape == (n, 6) and box.dtype == torch.float32
        assert idx.is_contiguous() and box.is_contiguous()
        if n:
            assert torch.equal(idx[:, 0].long(), m["b"]) and torch.equal(idx[:, 4].long(), m["cls"])
            assert (idx[:, 0] == 0).all()                      # image 1 has no labels
            assert (idx[:, 2] < sh.shape[2]).all() and (idx[:, 3] < sh.shape[3]).all() and (idx[:, 1] < 3).all()
            assert torch.equal(box[:, :4], m["box"]) and torch.equal(box[:, 4:], m["anchors"])
    assert sum(p[0].shape[0] for p in packed) > 0


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_host_target_assignment_equals_the_torch_one(seed):
    """loss.assign_labels_packed (numpy, host) == pack_assigned(assign_labels(...)) (the torch restatement of
    build_targets): same rows in the same order, bit for bit, including images without labels and no labels at all."""
    import types
    from adaptiveisp_amd.yolo import yolov3
    from adaptiveisp_amd.yolo.loss import DetectionLoss, assign_labels, assign_labels_packed, default_hyp, pack_assigned
    det = yolov3()
    loss_fn = DetectionLoss(det.model[-1].anchors, nc=80, hyp=default_hyp(80, 512))
    g = torch.Generator().manual_seed(seed)
    B = 5
    labels = []
    for b in range(B):
        n = 0 if (seed == 3 or b == 2) else int(torch.randint(0, 7, (1,), generator=g))
        t = torch.zeros(n, 6)
        t[:, 1] = torch.randint(0, 80, (n,), generator=g).float()
        t[:, 2:4] = torch.rand(n, 2, generator=g)
        t[:, 4:6] = torch.rand(n, 2, generator=g) * 0.6 + 0.01
        if n and b == 0:
            t[0, 2] = 1.0                              # centre ON the far edge: gxy == nx, cell index clamped to nx - 1 and
        if n > 1 and b == 1:                           # the box offset taken against the CLAMPED cell (loss.py:372-376)
            t[1, 3] = 1.0
        labels.append(t)
    shapes = [types.SimpleNamespace(shape=(B, 3, 512 // s, (512 + 64 * seed) // s, 85), device=torch.device("cpu")) for s in (8, 16, 32)]
    want = pack_assigned(assign_labels(loss_fn, shapes, labels))
    got = assign_labels_packed(loss_fn, shapes, labels, "cpu")
    assert len(got) == len(want) == 3
    for (gi, gb), (wi, wb) in zip(got, want):
        assert gi.dtype == torch.int32 and gb.dtype == torch.float32
        assert gi.shape == wi.shape and torch.equal(gi, wi)
        assert gb.shape == wb.shape and torch.equal(gb, wb)
    # pair form (one detector forward over [input batch; retouched batch], yolo.YoloTrainPairEngine): the same rows, then the
    # same rows again with image ids b + B; the B-image assignment is its leading half — what the labels duplicated would give
    one, two = assign_labels_packed(loss_fn, shapes, labels, "cpu", pair=True)
    dup = [types.SimpleNamespace(shape=(2 * B,) + tuple(sh.shape[1:]), device=sh.device) for sh in shapes]
    ref2 = assign_labels_packed(loss_fn, dup, labels + labels, "cpu")
    for (oi, ob), (ti, tb), (wi, wb), (ri, rb) in zip(one, two, want, ref2):
        n = wi.shape[0]
        assert torch.equal(oi, wi) and torch.equal(ob, wb) and oi.is_contiguous() and ob.is_contiguous()
        assert ti.shape == (2 * n, 5) and tb.shape == (2 * n, 6) and torch.equal(ti[:n], wi) and torch.equal(tb[n:], wb)
        assert torch.equal(ti[n:, 0], wi[:, 0] + B) and torch.equal(ti[n:, 1:], wi[:, 1:])
        # the duplicated labels give the same ROWS per image in the same relative order (the order the kernels' per-image
        # sums run in); across images build_targets interleaves them differently
        for b in range(2 * B):
            assert torch.equal(ti[ti[:, 0] == b], ri[ri[:, 0] == b]) and torch.equal(tb[ti[:, 0] == b], rb[ri[:, 0] == b])
