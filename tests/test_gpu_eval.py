"""GPU parity of the eval harness: the HIP greedy-NMS kernel (adayolo_nms) against the CPU oracle — index-exact,
same fp32 expression on both sides — and the whole evaluation loop on the HIP path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _boxes(n, seed, spread=300.0):
    r = np.random.default_rng(seed)
    c = r.uniform(0, spread, (max(n // 8, 1), 2))
    k = r.integers(0, c.shape[0], n)
    xy = c[k] + r.normal(0, 6, (n, 2))
    wh = r.uniform(10, 80, (n, 2))
    b = np.concatenate([xy - wh / 2, xy + wh / 2], 1).astype(np.float32)
    s = r.random(n).astype(np.float32)
    return b, s


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 500, 4097, 12000])
@pytest.mark.parametrize("thr,max_det", [(0.6, 300), (0.45, 7), (0.0, 300), (1.0, 50)])
def test_hip_nms_vs_oracle(oracle_mod, n, thr, max_det):
    from adaptiveisp_amd.val import hip_nms
    b, s = _boxes(n, 1000 + n)
    if n > 3:
        b[3] = b[2]                              # exact duplicates (IoU = 1) and a degenerate box
        b[1, 2:] = b[1, :2]
    order = np.argsort(-s, kind="stable")
    ref = order[oracle_mod.nms(b[order], thr, max_det=max_det)] if n else np.zeros(0, np.int64)
    got = hip_nms(torch.from_numpy(b).to(DEV), torch.from_numpy(s).to(DEV), thr, max_det).cpu().numpy()
    np.testing.assert_array_equal(got, ref)


def test_nms_wrapper_on_gpu_matches_golden(golden):
    from adaptiveisp_amd.val import non_max_suppression
    g = golden("evalharness")
    pred = torch.from_numpy(g["pred"].copy()).to(DEV)
    for tag, kw in (("ml", dict(conf_thres=0.05, iou_thres=0.6, multi_label=True, max_det=300)),
                    ("best", dict(conf_thres=0.25, iou_thres=0.45, multi_label=False, max_det=50)),
                    ("agn", dict(conf_thres=0.1, iou_thres=0.5, multi_label=True, agnostic=True, max_det=20))):
        res = non_max_suppression(pred.clone(), **kw)
        for b, r in enumerate(res):
            ref = g[f"nms.{tag}.{b}"]
            assert r.shape == ref.shape, (tag, b)
            # selection is exact; the stored numbers may differ in the last bit (GPU vs CPU elementwise mul / div)
            np.testing.assert_allclose(r.cpu().numpy(), ref, rtol=1e-6, atol=1e-6)


def test_run_eval_hip_path(tmp_path):
    """ISP (HIP) -> YoloEngine (HIP) -> HIP NMS -> mAP, random-init weights: checks the plumbing and that identical
    runs give identical metrics."""
    from _synth import synth_state_dict
    from adaptiveisp_amd.agent import Agent
    from adaptiveisp_amd.config import cfg
    from adaptiveisp_amd.val import run_eval
    from adaptiveisp_amd.yolo import YoloEngine, yolov3
    torch.manual_seed(0)
    agent = Agent(cfg, shape=(6 + len(cfg.filters), 64, 64), device=DEV).to(DEV)
    agent.load_state_dict(synth_state_dict(agent, seed=0))
    agent.eval()
    torch.manual_seed(1)
    det = yolov3().eval()
    eng = YoloEngine(det, 2, 96, 128, device=DEV)
    g = torch.Generator().manual_seed(5)
    imgs = torch.rand(2, 3, 96, 128, generator=g) * 0.6
    targets = torch.tensor([[0, 3, 0.3, 0.4, 0.2, 0.3], [1, 17, 0.7, 0.55, 0.5, 0.6], [1, 0, 0.5, 0.5, 0.1, 0.15]])
    shapes = [((96, 128), ((1.0, 1.0), (0.0, 0.0)))] * 2
    np.random.seed(0)
    r1 = run_eval(agent, eng, [(imgs, targets, ["a.png", "b.png"], shapes)], cfg, steps=5, conf_thres=0.3,
                  records_path=str(tmp_path / "records.txt"))
    np.random.seed(0)
    r2 = run_eval(agent, eng, [(imgs, targets, ["a.png", "b.png"], shapes)], cfg, steps=5, conf_thres=0.3)
    assert r1["seen"] == 2 and r1["nt"].sum() == 3
    assert r1["records"] == r2["records"] and r1["map50"] == r2["map50"]
    assert len(r1["records"][0][1]) == 5
