"""Checkpoint import: the detector pickle written by the REFERENCE's classes loads into this
package's DetectionModel without importing (or having) the reference, and reproduces the reference module's output;
hostile pickles execute nothing; ISP checkpoints keep the reference's dict keys."""
import io
import os
import pickle
import sys

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_detector_pickle_loads_without_reference(golden):
    from adaptiveisp_amd.yolo.checkpoint import load_detector_checkpoint, read_detector_pickle
    assert "models" not in sys.modules and "models.yolo" not in sys.modules
    g = golden("ckpt_import")
    info = read_detector_pickle(os.path.join(GOLD, "yolov3_w0625_refpickle.pt"))
    assert info["nc"] == 7 and info["source"] == "model" and info["epoch"] == 3
    assert info["names"] == {i: f"c{i}" for i in range(7)}
    assert "models" not in sys.modules                      # nothing of the reference got imported by the load
    m = load_detector_checkpoint(os.path.join(GOLD, "yolov3_w0625_refpickle.pt"))
    assert sum(p.numel() for p in m.parameters()) == int(g["nparams"])
    with torch.no_grad():
        pred, raws = m(torch.from_numpy(g["x"]))
    np.testing.assert_allclose(pred.numpy(), g["pred"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(raws[0].numpy(), g["raw0"], rtol=1e-5, atol=1e-5)


def test_fused_detector_pickle(golden):
    from adaptiveisp_amd.yolo.checkpoint import load_detector_checkpoint
    g = golden("ckpt_import")
    m = load_detector_checkpoint(os.path.join(GOLD, "yolov3_w0625_refpickle_fused.pt"))
    with torch.no_grad():
        pred, _ = m(torch.from_numpy(g["x"]))
    np.testing.assert_allclose(pred.numpy(), g["pred_fused_fp32"], rtol=2e-5, atol=2e-5)
    w, b = m.model[0].folded()                               # identity BN: folding returns the stored conv exactly
    assert torch.equal(w, m.model[0].conv.weight) and torch.equal(b, m.model[0].bn.bias)


class _Evil:
    def __reduce__(self):
        return (os.system, ("touch /tmp/adaisp_pwned",))


def test_hostile_pickle_runs_nothing(tmp_path):
    from adaptiveisp_amd.yolo.checkpoint import read_detector_pickle
    marker = "/tmp/adaisp_pwned"
    if os.path.exists(marker):
        os.remove(marker)
    p = tmp_path / "evil.pt"
    torch.save({"model": torch.nn.Linear(2, 2), "opt": _Evil(), "ema": None}, p)
    info = read_detector_pickle(str(p))                       # Linear is a torch.nn module: allowed, inert payload ignored
    assert not os.path.exists(marker)
    assert "weight" in info["state_dict"]
    buf = io.BytesIO()
    pickle.dump(_Evil(), buf)
    buf.seek(0)
    from adaptiveisp_amd.yolo.checkpoint import _RestrictedUnpickler
    _RestrictedUnpickler(buf).load()
    assert not os.path.exists(marker)


def test_isp_checkpoint_roundtrip(tmp_path):
    from _engine import cpu_agent, cpu_value
    from adaptiveisp_amd.config import cfg
    from adaptiveisp_amd.yolo.checkpoint import ISP_KEYS, load_isp_checkpoint, save_isp_checkpoint
    a, v = cpu_agent(cfg, seed=3), cpu_value(cfg, seed=4)
    opt = torch.optim.Adam(a.parameters(), lr=3e-5)
    p = tmp_path / "ckpt.pth"
    save_isp_checkpoint(str(p), 1000, a, v, opt, None)
    raw = torch.load(str(p), weights_only=True)
    assert tuple(raw) == ISP_KEYS and raw["iter"] == 1000      # train.py:475-485
    a2, v2 = cpu_agent(cfg, seed=5), cpu_value(cfg, seed=6)
    load_isp_checkpoint(str(p), a2, v2)
    for k, t in a.state_dict().items():
        assert torch.equal(t, a2.state_dict()[k]), k
    for k, t in v.state_dict().items():
        assert torch.equal(t, v2.state_dict()[k]), k


def test_width_multiple_matches_reference_channel_rounding():
    from adaptiveisp_amd.yolo.model import DetectionModel, make_divisible
    assert [make_divisible(c * 0.0625, 8) for c in (32, 64, 128, 256, 512, 1024)] == [8, 8, 8, 16, 32, 64]
    m = DetectionModel(nc=7, width=0.0625)
    assert m.model[0].conv.out_channels == 8 and m.model[9].conv.out_channels == 64
    assert m.model[28].m[0].out_channels == 3 * 12


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
def test_imported_pickle_through_the_hip_engine(golden, fused):
    """f2 end to end on the device: a detector pickled by the REFERENCE's classes (yolov3/models/experimental.py:73-89
    is the loader this replaces) -> restricted unpickler -> DetectionModel -> YoloEngine (bf16 MFMA kernels) -> the
    output the reference module itself produced (`ckpt_import.npz`). The fixture is a width-0.0625, nc=7 network, so
    this also drives the generic first conv (letterbox pack + Cin=8 conv) and the narrow-channel tiles."""
    from adaptiveisp_amd.yolo import YoloEngine
    from adaptiveisp_amd.yolo.checkpoint import load_detector_checkpoint
    g = golden("ckpt_import")
    name = "yolov3_w0625_refpickle_fused.pt" if fused else "yolov3_w0625_refpickle.pt"
    m = load_detector_checkpoint(os.path.join(GOLD, name)).eval()
    x = torch.from_numpy(g["x"])
    B, _, H, W = x.shape
    eng = YoloEngine(m, B, H, W, device="cuda:0")
    assert eng._stem is None and eng.plan[0][0] == "pack"
    pred = eng(x.to("cuda:0")).cpu().numpy()
    ref = g["pred_fused_fp32"] if fused else g["pred"]
    rel = np.abs(pred - ref) / (np.abs(ref) + 1.0)
    assert rel.max() < 2e-2, rel.max()
    raw0 = eng.raw_maps()[0].cpu().numpy()
    if not fused:
        assert np.abs(raw0 - g["raw0"]).max() <= 3e-2 * np.abs(g["raw0"]).max()
    # the detections that survive a confidence threshold are the same boxes
    conf = ref[..., 4] * ref[..., 5:].max(-1)
    top = np.argsort(-conf[0])[:20]
    np.testing.assert_allclose(pred[0, top, :4], ref[0, top, :4], rtol=3e-2, atol=0.5)
