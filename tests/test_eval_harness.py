"""Eval harness on CPU: the restated non_max_suppression / box helpers / matching / AP against
fixtures generated from the REFERENCE's own code (tests/golden/gen_golden.py::gen_eval), with the CPU oracle's
greedy NMS injected where the product uses the HIP kernel."""
import numpy as np
import pytest
import torch


def _oracle_nms_fn(oracle_mod, max_det):
    def fn(boxes, scores, thr):
        order = torch.argsort(scores, descending=True, stable=True)
        keep = oracle_mod.nms(boxes[order].numpy(), thr, max_det=max(boxes.shape[0], 1))
        return order[torch.from_numpy(keep)]
    return fn


CASES = {"ml": dict(conf_thres=0.05, iou_thres=0.6, multi_label=True, max_det=300),
         "best": dict(conf_thres=0.25, iou_thres=0.45, multi_label=False, max_det=50),
         "agn": dict(conf_thres=0.1, iou_thres=0.5, multi_label=True, agnostic=True, max_det=20),
         "cls": dict(conf_thres=0.1, iou_thres=0.5, multi_label=False, classes=[1, 4], max_det=300)}


@pytest.mark.parametrize("tag", sorted(CASES))
def test_nms_wrapper_matches_reference(golden, oracle_mod, tag):
    from adaptiveisp_amd.val import non_max_suppression
    g = golden("evalharness")
    kw = CASES[tag]
    res = non_max_suppression(torch.from_numpy(g["pred"].copy()), nms_fn=_oracle_nms_fn(oracle_mod, kw["max_det"]), **kw)
    for b, r in enumerate(res):
        ref = g[f"nms.{tag}.{b}"]
        assert r.shape == ref.shape
        np.testing.assert_array_equal(r.numpy(), ref)          # same fp32 ops in the same order: bit-exact


def test_nms_rejects_cpu_tensors_on_product_path():
    from adaptiveisp_amd.val import hip_nms
    from adaptiveisp_amd.yolo._lib import AdayoloError
    with pytest.raises(AdayoloError):
        hip_nms(torch.zeros(3, 4), torch.zeros(3), 0.5)


def test_oracle_nms_basic(oracle_mod):
    b = np.array([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10.5], [0, 0, 0, 0]], np.float32)
    assert oracle_mod.nms(b, 0.5).tolist() == [0, 2, 4]         # degenerate box: IoU is NaN/0 -> never suppressed
    assert oracle_mod.nms(b, 0.5, max_det=2).tolist() == [0, 2]
    assert oracle_mod.nms(np.zeros((0, 4), np.float32), 0.5).tolist() == []
    # a box is suppressed only by a KEPT box: 1 kills 2? no - 0 kills 1, so 2 (overlapping 1 but not 0) survives
    c = np.array([[0, 0, 10, 10], [4, 0, 14, 10], [8, 0, 18, 10]], np.float32)
    assert oracle_mod.nms(c, 0.4).tolist() == [0, 2]


def test_box_helpers(golden):
    from adaptiveisp_amd.val import scale_boxes, xywh2xyxy, xyxy2xywh
    g = golden("evalharness")
    b = g["boxes"]
    np.testing.assert_array_equal(xywh2xyxy(torch.from_numpy(b.copy())).numpy(), g["xywh2xyxy"])
    np.testing.assert_array_equal(xyxy2xywh(torch.from_numpy(b.copy())).numpy(), g["xyxy2xywh"])
    np.testing.assert_array_equal(xywh2xyxy(b.copy()), g["xywh2xyxy"])
    np.testing.assert_array_equal(scale_boxes((512, 512), torch.from_numpy(b.copy()), (375, 500)).numpy(), g["scale_auto"])
    np.testing.assert_array_equal(scale_boxes((512, 512), torch.from_numpy(b.copy()), (375, 500),
                                              ((1.024, 1.024), (0.0, 64.0))).numpy(), g["scale_ratio_pad"])


def test_letterbox_geometry():
    from adaptiveisp_amd.val import letterbox_geometry, letterbox_pad
    # reference numbers (augmentations.py:111-141) for a 375x500 image into 512 with auto=False: r=1.024, pad rows only
    ratio, unpad, (dw, dh), (t, b, l, r) = letterbox_geometry((375, 500), 512, auto=False)
    assert ratio == (1.024, 1.024) and unpad == (512, 384) and (dw, dh) == (0.0, 64.0) and (t, b, l, r) == (64, 64, 0, 0)
    # minimum-rectangle mode pads to the next stride multiple only
    _, unpad, (dw, dh), pads = letterbox_geometry((720, 1280), (736, 1280), auto=True)
    assert unpad == (1280, 720) and (dw, dh) == (0.0, 8.0) and pads == (8, 8, 0, 0)
    im = np.full((720, 1280, 3), 7, np.uint8)
    out, _, _ = letterbox_pad(im, (736, 1280))
    assert out.shape == (736, 1280, 3) and (out[:8] == 114).all() and (out[8:728] == 7).all() and (out[728:] == 114).all()


def test_matching_and_ap(golden):
    from adaptiveisp_amd.val import ap_per_class, box_iou, compute_ap, process_batch
    g = golden("evalharness")
    det, lab = torch.from_numpy(g["det"]), torch.from_numpy(g["lab"])
    np.testing.assert_array_equal(box_iou(lab[:, 1:], det[:, :4]).numpy(), g["iou"])
    correct = process_batch(det, lab, torch.linspace(0.5, 0.95, 10))
    np.testing.assert_array_equal(correct.numpy(), g["correct"])
    tp, fp, p, r, f1, ap, cls = ap_per_class(correct.numpy(), g["det"][:, 4], g["det"][:, 5], g["lab"][:, 0])
    for mine, key in ((tp, "ap_tp"), (fp, "ap_fp"), (p, "ap_p"), (r, "ap_r"), (f1, "ap_f1"), (ap, "ap_ap")):
        np.testing.assert_allclose(mine, g[key], rtol=0, atol=1e-12)
    np.testing.assert_array_equal(cls, g["ap_cls"])
    a, mpre, mrec = compute_ap(g["cap_rec"], g["cap_prec"])
    assert abs(a - float(g["cap_ap"])) < 1e-12
    np.testing.assert_array_equal(mpre, g["cap_mpre"])
    np.testing.assert_array_equal(mrec, g["cap_mrec"])


def test_run_eval_on_cpu_with_oracle(oracle_mod, tmp_path):
    """Plumbing of the whole loop: ISP steps through the oracle-backed agent, a fake detector that returns boxes
    around the labels, NMS (oracle core), matching, mAP, records.txt."""
    from _engine import cpu_agent
    from adaptiveisp_amd.config import cfg
    from adaptiveisp_amd.val import run_eval
    agent = cpu_agent(cfg)
    g = torch.Generator().manual_seed(3)
    imgs = torch.rand(2, 3, 64, 96, generator=g) * 0.5
    targets = torch.tensor([[0, 1, 0.30, 0.40, 0.20, 0.30], [0, 2, 0.70, 0.55, 0.25, 0.30], [1, 0, 0.52, 0.48, 0.30, 0.35]])

    def detector(x):
        pred = torch.zeros(2, 8, 5 + 3)
        for t in targets:
            b, c = int(t[0]), int(t[1])
            k = int((pred[b, :, 4] > 0).sum())
            pred[b, k, :4] = t[2:] * torch.tensor([96., 64., 96., 64.])
            pred[b, k, 4] = 0.9
            pred[b, k, 5 + c] = 0.95
        pred[0, 5, :4] = torch.tensor([10., 10., 8., 8.]); pred[0, 5, 4] = 0.5; pred[0, 5, 5] = 0.9   # a false positive
        return pred

    rec = tmp_path / "records.txt"
    res = run_eval(agent, detector, [(imgs, targets, ["a.png", "b.png"], [((64, 96), ((1.0, 1.0), (0.0, 0.0)))] * 2)],
                   cfg, steps=3, conf_thres=0.001, iou_thres=0.6, nc=3, records_path=str(rec),
                   nms_fn=_oracle_nms_fn(oracle_mod, 300), param_dir=str(tmp_path / "param_results"))
    import json
    pj = json.loads((tmp_path / "param_results" / "a.json").read_text())
    assert len(pj["pipeline"]) == 3 and all(res["filter_names"][k] in pj for k in pj["pipeline"])
    assert res["seen"] == 2 and res["nt"].tolist() == [1, 1, 1]
    assert res["map50"] > 0.99 and 0.0 < res["mp"] <= 1.0
    lines = rec.read_text().strip().split("\n")
    assert lines[0].split(",") == res["filter_names"] and len(lines) == 3 and lines[1].startswith("a.png,")
    assert all(len(l.split(",")) == 4 for l in lines[1:])


def test_nms_label_priors_and_empty_images(oracle_mod):
    """Autolabelling priors (general.py:911-918) enter as confidence-1 detections of their class even when no
    prediction passes the threshold; images without candidates give empty [0,6] results."""
    from adaptiveisp_amd.val import non_max_suppression
    pred = torch.zeros(3, 6, 5 + 4)
    pred[..., :4] = torch.tensor([50.0, 50.0, 10.0, 10.0])
    pred[1, 2, 4], pred[1, 2, 5 + 1] = 0.9, 0.8                      # one real detection in image 1
    labels = [torch.tensor([[2.0, 10.0, 10.0, 4.0, 4.0]]), torch.zeros((0, 5)), torch.zeros((0, 5))]
    out = non_max_suppression(pred, conf_thres=0.25, iou_thres=0.5, labels=labels, nms_fn=_oracle_nms_fn(oracle_mod, 300))
    assert [o.shape for o in out] == [(1, 6), (1, 6), (0, 6)]
    np.testing.assert_allclose(out[0][0].numpy(), [8, 8, 12, 12, 1.0, 2.0])
    np.testing.assert_allclose(out[1][0].numpy(), [45, 45, 55, 55, 0.9 * 0.8, 1.0], rtol=1e-6)
    with pytest.raises(ValueError):
        non_max_suppression(pred, conf_thres=1.5)


def test_matching_rule_edge_cases():
    from adaptiveisp_amd.val import process_batch
    iouv = torch.linspace(0.5, 0.95, 10)
    lab = torch.tensor([[0.0, 0, 0, 10, 10], [1.0, 20, 20, 30, 30]])
    det = torch.tensor([[0, 0, 10, 10, 0.9, 0.0],        # perfect on label 0
                        [0, 0, 10, 9, 0.8, 0.0],         # IoU 0.9 with label 0, already taken by detection 0
                        [20, 20, 30, 30, 0.7, 0.0],      # right box, wrong class
                        [21, 20, 30, 30, 0.6, 1.0]])     # IoU 0.9 with label 1
    c = process_batch(det, lab, iouv)
    assert c[0].all() and not c[1].any() and not c[2].any()
    assert c[3].tolist() == [True] * 9 + [False]
    assert process_batch(det[:0], lab, iouv).shape == (0, 10) and not process_batch(det, lab[:0], iouv).any()
