"""Detector module tree vs the reference (golden vectors from the reference's own DetectionModel)."""
import json
import os

import numpy as np
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
