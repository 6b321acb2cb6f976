"""Deterministic synthetic weights shared by the golden-fixture generator (which loads them into the
reference modules) and the tests (which load them into this repo's modules). Values depend only on
(seed, key name, shape) — never on torch's RNG or on module construction order."""
import zlib

import numpy as np
import torch


def synth_tensor(key, shape, dtype, seed=0, gain=1.4):
    rng = np.random.default_rng([seed, zlib.crc32(key.encode())])
    shape = tuple(shape)
    if key.endswith("num_batches_tracked"):
        return torch.zeros(shape, dtype=dtype)
    if key.endswith("running_var"):
        a = rng.uniform(0.5, 1.5, shape)
    elif key.endswith("running_mean"):
        a = rng.normal(0.0, 0.1, shape)
    elif len(shape) >= 2:                      # conv / linear weight
        fan_in = int(np.prod(shape[1:]))
        a = rng.normal(0.0, 1.0, shape) * (gain / np.sqrt(fan_in))
    elif key.endswith("weight"):               # batch-norm scale
        a = rng.uniform(0.8, 1.2, shape)
    else:                                      # biases
        a = rng.normal(0.0, 0.05, shape)
    return torch.from_numpy(np.asarray(a, dtype=np.float32)).to(dtype)


def synth_state_dict(module, seed=0, gain=1.4):
    return {k: synth_tensor(k, v.shape, v.dtype, seed, gain) for k, v in module.state_dict().items()}


def synth_yolo_state_dict(model, seed=2):
    """Detector weights: unit gain keeps the logits O(1) through the 75 conv layers; real anchors kept."""
    sd = synth_state_dict(model, seed=seed, gain=1.0)
    sd["model.28.anchors"] = model.state_dict()["model.28.anchors"].clone()
    return sd


def test_image(B, H, W, seed, special=True):
    """Dark linear-like image (U^2.2 * 0.5) with a strip of edge-case pixels."""
    rng = np.random.default_rng(seed)
    x = (rng.random((B, 3, H, W)) ** 2.2 * 0.5).astype(np.float32)
    if special and W >= 16 and H >= 4:
        x[:, :, 0, 0] = 0.0                                 # max == 0
        x[:, :, 0, 1] = 0.25                                # grey: min == max
        x[:, 0, 0, 2] = 0.5; x[:, 1, 0, 2] = 0.5; x[:, 2, 0, 2] = 0.1    # R == G > B
        x[:, 0, 0, 3] = 0.1; x[:, 1, 0, 3] = 0.6; x[:, 2, 0, 3] = 0.6    # G == B > R
        x[:, 0, 0, 4] = 0.7; x[:, 1, 0, 4] = 0.2; x[:, 2, 0, 4] = 0.7    # R == B > G
        x[:, :, 0, 5] = np.array([1.3, 0.4, -0.2], np.float32)           # out of [0,1]
        x[:, :, 0, 6] = np.array([0.9, 0.95, 1.0], np.float32)
        x[:, :, 0, 7] = np.array([0.3, 0.1, 0.2], np.float32)            # R max, G < B -> negative hue
        x[:, :, 1, 0:8] = rng.random((B, 3, 8)).astype(np.float32)       # bright full-range pixels
        x[:, :, 1, 8:12] = (rng.random((B, 3, 4)) * 2.0 - 0.5).astype(np.float32)
        x[:, :, 2, 0:8] = np.float32(0.125) * np.arange(8, dtype=np.float32)   # tone-curve breakpoints
        x[:, :, 3, 0] = 0.0005                                                # below the gamma floor
    return x
