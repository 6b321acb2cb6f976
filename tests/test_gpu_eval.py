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


def test_run_eval_graph_replay_equals_the_eager_loop():
    """run_eval(graph=True): each batch's ISP episode + detector forward as one hipGraph replay (the batch-1 loop of config 3
    is host-bound). Same records, same detections after NMS, same `correct` matrices and mAP as the eager loop — over several
    batches of one shape (one capture, replayed) and a second shape (a second capture); and the reference's early exit still
    wins when `stopped` is raised before the last step (steps beyond cfg.test_steps: the batch is redone eagerly)."""
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
    engines = {(1, 128, 160): YoloEngine(det, 1, 128, 160, device=DEV), (2, 96, 128): YoloEngine(det, 2, 96, 128, device=DEV)}
    detector = lambda x: engines[(x.shape[0], x.shape[2], x.shape[3])](x)          # noqa: E731
    g = torch.Generator().manual_seed(9)
    batches = []
    for i in range(5):
        B, H, W = (1, 128, 160) if i != 2 else (2, 96, 128)
        t = torch.zeros(2 * B, 6)
        t[:, 0] = torch.arange(2 * B) // 2
        t[:, 1] = torch.randint(0, 80, (2 * B,), generator=g).float()
        t[:, 2:4] = torch.rand(2 * B, 2, generator=g) * 0.5 + 0.25
        t[:, 4:6] = torch.rand(2 * B, 2, generator=g) * 0.3 + 0.1
        batches.append((torch.rand(B, 3, H, W, generator=g) * 0.6, t, [f"im{i}_{b}.png" for b in range(B)],
                        [((H, W), ((1.0, 1.0), (0.0, 0.0)))] * B))
    for steps in (5, 7):                                   # 7 > cfg.test_steps: `stopped` is raised at step 5 -> early exit
        out = {}
        for mode in (False, True):
            np.random.seed(3)
            det_list = []
            out[mode] = (run_eval(agent, detector, batches, cfg, steps=steps, conf_thres=0.2, graph=mode, details=det_list), det_list)
        (ra, da), (rb, db) = out[False], out[True]
        assert ra["records"] == rb["records"] and ra["seen"] == rb["seen"] == 6
        assert ra["map50"] == rb["map50"] and ra["map"] == rb["map"] and np.array_equal(ra["nt"], rb["nt"])
        for a, b in zip(da, db):
            assert a["path"] == b["path"] and torch.equal(a["retouch"], b["retouch"]) and torch.equal(a["pred"], b["pred"])
            assert (a["correct"] is None and b["correct"] is None) or torch.equal(a["correct"], b["correct"])
        if steps == 7:
            assert all(row[5:] == ["-1", "-1"] for _, row in ra["records"])       # the loop stopped after step 5


def _write_lod_folder(root):
    """A synthetic LOD-style folder: images/ + labels/ (YOLO txt), four PNGs of different native sizes and aspect ratios."""
    import os
    from PIL import Image
    os.makedirs(root / "images"); os.makedirs(root / "labels")
    rng = np.random.default_rng(11)
    paths = []
    for i, (h, w) in enumerate([(600, 800), (512, 384), (333, 500), (720, 1280)]):
        base = rng.random((h // 8 + 1, w // 8 + 1, 3))
        im = np.kron(base, np.ones((8, 8, 1)))[:h, :w] * 0.35 + rng.random((h, w, 3)) * 0.1       # dark, blocky, noisy
        Image.fromarray((im * 255).astype(np.uint8)).save(root / "images" / f"img{i}.png")
        n = 2 + i
        lb = np.concatenate([rng.integers(0, 7, (n, 1)).astype(np.float64), rng.uniform(0.25, 0.75, (n, 2)),
                             rng.uniform(0.1, 0.4, (n, 2))], 1)
        np.savetxt(root / "labels" / f"img{i}.txt", lb, fmt="%.6f")
        paths.append(str(root / "images" / f"img{i}.png"))
    return paths


def test_config3_eval_loop_at_its_shape(tmp_path, oracle_mod):
    """BASELINE config 3 at its own shape (val_adaptiveisp.py:287-310,337,466-467): the LOD loader at img_size 512, batch 1
    (512 x 512 square letterbox, images of different native sizes), a detector imported from a reference-pickled
    checkpoint, five ISP steps with the per-step early-exit check, NMS at conf 0.001, box rescaling, matching, mAP.
    The GPU run (fused policy + HIP filters with fused pooling, YoloEngine bf16, HIP NMS) against a CPU run of the SAME
    loop with the oracle standing in for every kernel (tests/_engine.py: C oracle ISP + pooling, torch-CPU heads, oracle
    greedy NMS): filter ids per step (records.txt) equal, retouched images within the episode tolerance, and — the CPU
    loop being fed the engine's predictions, so that bf16-vs-fp32 logits do not enter — the detections after NMS, every
    per-image `correct` matrix and the mAP figures EQUAL. The engine's predictions themselves are held to the fp32
    module tree on the same retouched image (bf16 tolerance)."""
    import os
    from _engine import cpu_agent
    from _synth import synth_state_dict
    from adaptiveisp_amd.agent import Agent
    from adaptiveisp_amd.config import cfg
    from adaptiveisp_amd.val import run_eval
    from adaptiveisp_amd.val.loader import LODImages
    from adaptiveisp_amd.yolo import YoloEngine
    from adaptiveisp_amd.yolo.checkpoint import load_detector_checkpoint
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    paths = _write_lod_folder(tmp_path)
    data = LODImages(str(tmp_path / "images"), img_size=512, batch_size=1)
    batches = list(data)
    assert len(batches) == 4 and all(tuple(b[0].shape) == (1, 3, 512, 512) for b in batches)
    assert sorted(os.path.basename(b[2][0]) for b in batches) == [f"img{i}.png" for i in range(4)]
    det = load_detector_checkpoint(os.path.join(gold, "yolov3_w0625_refpickle.pt")).eval()
    nc = det.model[-1].nc
    agent = Agent(cfg, shape=(6 + len(cfg.filters), 64, 64), device=DEV)
    agent.load_state_dict(synth_state_dict(agent, seed=0))
    agent = agent.to(DEV).eval()
    eng = YoloEngine(det, 1, 512, 512, device=DEV)

    # ---- the GPU loop -------------------------------------------------------------------------------------------
    cache = {}

    def gpu_detector(x):
        p = eng(x).clone()
        cache[len(cache)] = (x.detach().cpu().clone(), p.detach().cpu().clone())
        return p

    np.random.seed(0)
    dg = []
    rg = run_eval(agent, gpu_detector, batches, cfg, steps=5, conf_thres=0.001, iou_thres=0.6, nc=nc,
                  records_path=str(tmp_path / "records.txt"), details=dg)
    assert agent._fast is not None and rg["seen"] == 4
    rows = open(tmp_path / "records.txt").read().strip().splitlines()
    assert rows[0] == ",".join(f.get_short_name() for f in agent.filters) and len(rows) == 5
    for (name, ids), line in zip(rg["records"], rows[1:]):
        assert line == name + "," + ",".join(ids) and len(ids) == 5
        # test_steps = 5: `stopped` is raised by the fifth step, never earlier (agent.py:234-259) — all five ids are real
        assert all(0 <= int(k) < len(agent.filters) for k in ids)

    # ---- the same loop on the CPU, the oracle standing in for the kernels ------------------------------------------
    cag = cpu_agent(cfg, seed=0)
    feed = iter(range(len(cache)))
    cpu_retouch = []

    def cpu_detector(x):
        i = next(feed)
        cpu_retouch.append(x.detach().clone())
        return cache[i][1].clone()                          # the engine's predictions: bf16 logits stay out of the comparison

    def oracle_nms_fn(boxes, scores, thr):                  # (boxes arrive sorted by descending score)
        b = boxes.detach().cpu().numpy().astype(np.float32)
        return torch.from_numpy(oracle_mod.nms(b, float(thr), max_det=max(len(b), 1)))

    np.random.seed(0)
    dc = []
    rc = run_eval(cag, cpu_detector, batches, cfg, steps=5, conf_thres=0.001, iou_thres=0.6, nc=nc, nms_fn=oracle_nms_fn,
                  details=dc)
    assert rc["records"] == rg["records"], (rc["records"], rg["records"])
    for i, (a, b) in enumerate(zip(dg, dc)):
        assert a["path"] == b["path"]
        d = (cache[i][0] - cpu_retouch[i]).abs()
        # five chained steps whose PARAMETERS come from two head implementations (fused HIP policy vs torch-CPU modules:
        # ~1e-6 apart), passed through sharpen's (1 + 2f) gain and CCM's row normalisation: a few 1e-5 on a fraction of a
        # per cent of the pixels, never more than 2e-4 (measured on the MI355X: 0.18 % of the pixels above 1e-5 rel + 2e-6)
        assert float(d.max()) < 2e-4, (i, float(d.max()))
        assert float((d > 1e-5 * cpu_retouch[i].abs() + 2e-6).float().mean()) < 1e-2
        assert a["pred"].shape == b["pred"].shape and a["pred"].shape[0] > 0, (i, a["pred"].shape, b["pred"].shape)
        np.testing.assert_allclose(a["pred"].numpy(), b["pred"].numpy(), rtol=1e-6, atol=1e-6)    # same boxes, same order
        assert (a["correct"] is None) == (b["correct"] is None)
        if a["correct"] is not None:
            assert torch.equal(a["correct"], b["correct"]), i
    for k in ("mp", "mr", "map50", "map75", "map", "seen"):
        assert rg[k] == pytest.approx(rc[k], rel=1e-6, abs=1e-9), k
    assert np.array_equal(rg["nt"], rc["nt"]) and rg["nt"].sum() == 2 + 3 + 4 + 5
    # ---- and the engine against the fp32 module tree on the same retouched image ---------------------------------------
    with torch.no_grad():
        for i in (0, 3):
            ref = det(cache[i][0])[0].numpy()
            got = cache[i][1].numpy()
            rel = np.abs(got - ref) / (np.abs(ref) + 1.0)
            assert rel.max() < 3e-2, (i, rel.max())
