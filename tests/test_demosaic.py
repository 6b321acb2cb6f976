"""Bayer demosaic extension. CPU: the oracle's defining properties and its consistency with the
reference's Bayer packing (fixture from isp/unprocess_np.py:82-128); GPU: the HIP kernel bit-exact against the oracle."""
import numpy as np
import pytest
import torch

PATTERNS = {"RGGB": (0, 0), "GRBG": (0, 1), "GBRG": (1, 0), "BGGR": (1, 1)}


def _cfa_from_rgb(rgb_u16, pattern):
    """Sample an RGB uint16 image [B,3,H,W] through the colour filter array."""
    ry, rx = PATTERNS[pattern]
    B, _, H, W = rgb_u16.shape
    raw = np.empty((B, H, W), np.uint16)
    raw[:, ry::2, rx::2] = rgb_u16[:, 0, ry::2, rx::2]
    raw[:, ry::2, 1 - rx::2] = rgb_u16[:, 1, ry::2, 1 - rx::2]
    raw[:, 1 - ry::2, rx::2] = rgb_u16[:, 1, 1 - ry::2, rx::2]
    raw[:, 1 - ry::2, 1 - rx::2] = rgb_u16[:, 2, 1 - ry::2, 1 - rx::2]
    return raw


@pytest.mark.parametrize("pattern", sorted(PATTERNS))
def test_oracle_properties(oracle_mod, pattern):
    pid = 2 * PATTERNS[pattern][0] + PATTERNS[pattern][1]
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 65536, (2, 3, 12, 18)).astype(np.uint16)
    raw = _cfa_from_rgb(rgb, pattern)
    out = oracle_mod.demosaic(raw, pid)
    ry, rx = PATTERNS[pattern]
    scale = np.float32(1.0) / np.float32(65535.0)
    # the sampled colour comes back untouched at its own site
    np.testing.assert_array_equal(out[:, 0, ry::2, rx::2], raw[:, ry::2, rx::2].astype(np.float32) * scale)
    np.testing.assert_array_equal(out[:, 2, 1 - ry::2, 1 - rx::2], raw[:, 1 - ry::2, 1 - rx::2].astype(np.float32) * scale)
    np.testing.assert_array_equal(out[:, 1, ry::2, 1 - rx::2], raw[:, ry::2, 1 - rx::2].astype(np.float32) * scale)
    # a flat colour is reproduced exactly, borders included
    flat = np.empty((1, 3, 8, 10), np.uint16)
    flat[:, 0], flat[:, 1], flat[:, 2] = 16384, 32768, 49152
    o = oracle_mod.demosaic(_cfa_from_rgb(flat, pattern), pid, black=0.0, white=65536.0)
    np.testing.assert_array_equal(o, np.broadcast_to(np.array([0.25, 0.5, 0.75], np.float32)[None, :, None, None], o.shape))
    # bilinear interpolation is exact on a linear ramp away from the border
    yy, xx = np.meshgrid(np.arange(16), np.arange(20), indexing="ij")
    ramp = np.stack([100 + 8 * xx + 4 * yy, 200 + 2 * xx + 6 * yy, 50 + 10 * xx + 2 * yy])[None].astype(np.uint16)
    o = oracle_mod.demosaic(_cfa_from_rgb(ramp, pattern), pid, black=0.0, white=1.0)
    np.testing.assert_array_equal(o[:, :, 1:-1, 1:-1], ramp[:, :, 1:-1, 1:-1].astype(np.float32))
    # black / white levels
    o2 = oracle_mod.demosaic(raw, pid, black=64.0, white=1023.0)
    assert abs(float(o2[0, 0, ry, rx]) - (float(raw[0, ry, rx]) - 64.0) / 959.0) < 1e-6 * max(1.0, abs(float(o2[0, 0, ry, rx])))


def test_consistent_with_reference_bayer_packing(golden, oracle_mod):
    """mosaic() (RGB -> packed RGGB) then our plane layout == the reference's reconstruct_bayer; demosaicing that plane
    returns the packed samples at their sites."""
    from adaptiveisp_amd._lib import pack_rggb_to_plane
    g = golden("mosaic")
    packed = torch.from_numpy(g["packed"])
    plane = pack_rggb_to_plane(packed).numpy()
    np.testing.assert_array_equal(plane, g["plane_rggb"])
    img = g["img"]                                                    # HWC
    np.testing.assert_array_equal(plane[0::2, 0::2], img[0::2, 0::2, 0])
    np.testing.assert_array_equal(plane[1::2, 1::2], img[1::2, 1::2, 2])
    raw = np.round(plane * 65535.0).astype(np.uint16)[None]
    out = oracle_mod.demosaic(raw, 0)
    s = np.float32(1.0) / np.float32(65535.0)
    np.testing.assert_array_equal(out[0, 0, 0::2, 0::2], raw[0, 0::2, 0::2].astype(np.float32) * s)
    np.testing.assert_array_equal(out[0, 1, 0::2, 1::2], raw[0, 0::2, 1::2].astype(np.float32) * s)
    np.testing.assert_array_equal(out[0, 1, 1::2, 0::2], raw[0, 1::2, 0::2].astype(np.float32) * s)
    np.testing.assert_array_equal(out[0, 2, 1::2, 1::2], raw[0, 1::2, 1::2].astype(np.float32) * s)


def test_demosaic_rejects_cpu_and_bad_shapes():
    from adaptiveisp_amd import _lib
    with pytest.raises(_lib.AdaispError):
        _lib.demosaic(torch.zeros(1, 4, 4, dtype=torch.uint16))
    L = _lib.load()
    import ctypes
    buf = (ctypes.c_uint16 * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert L.adaisp_demosaic(None, p, 1, 4, 4, 0, 0.0, 1.0, None) == -1
    assert L.adaisp_demosaic(p, p, 1, 3, 4, 0, 0.0, 1.0, None) != 0          # odd height
    assert L.adaisp_demosaic(p, p, 1, 4, 4, 7, 0.0, 1.0, None) == -1         # unknown pattern
    assert L.adaisp_demosaic(p, p, 1, 4, 4, 0, 1.0, 1.0, None) == -1         # empty range


@pytest.mark.gpu
@pytest.mark.parametrize("pattern", sorted(PATTERNS))
@pytest.mark.parametrize("shape", [(1, 2, 2), (2, 34, 130), (1, 66, 258), (3, 30, 50), (1, 720, 1280)])
def test_hip_demosaic_bit_exact(oracle_mod, pattern, shape):
    from adaptiveisp_amd import _lib
    rng = np.random.default_rng(sum(shape))
    raw = rng.integers(0, 4096, shape).astype(np.uint16)
    pid = 2 * PATTERNS[pattern][0] + PATTERNS[pattern][1]
    ref = oracle_mod.demosaic(raw, pid, black=64.0, white=4095.0)
    got = _lib.demosaic(torch.from_numpy(raw).to("cuda:0"), pattern, 64.0, 4095.0).cpu().numpy()
    np.testing.assert_array_equal(got, ref)


@pytest.mark.gpu
def test_hip_demosaic_feeds_the_isp(oracle_mod):
    """raw Bayer -> demosaic -> one ISP step, against the oracle chain."""
    from adaptiveisp_amd import _lib
    rng = np.random.default_rng(9)
    raw = (rng.random((2, 48, 64)) ** 2.2 * 0.5 * 65535).astype(np.uint16)
    x = _lib.demosaic(torch.from_numpy(raw).to("cuda:0"))
    p = torch.tensor([[0.8], [1.4]], device="cuda:0")
    out = _lib.process(_lib.OP_EXPOSURE, x, p, clip=True).cpu().numpy()
    ref = oracle_mod.forward(oracle_mod.demosaic(raw, 0), 0, p.cpu().numpy(), clip=True)
    np.testing.assert_allclose(out, ref, rtol=1e-5, atol=2e-6)
