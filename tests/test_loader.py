"""LOD image loader (adaptiveisp_amd/val/loader.py): the resampling kernels by their defining properties (cv2 is absent,
so parity with cv2 itself is unpinned — stated in the module), the geometry against the reference's letterbox numbers,
and the batches against what the eval loop expects."""
import os

import numpy as np
import pytest
import torch


def _img(h, w, seed=0):
    return np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)


def test_linear_resize_properties():
    from adaptiveisp_amd.val.loader import resize_linear_u8
    im = _img(37, 53)
    assert np.array_equal(resize_linear_u8(im, (53, 37)), im)                       # same size: copy
    flat = np.full((20, 30, 3), 137, np.uint8)
    assert (resize_linear_u8(flat, (77, 41)) == 137).all()                           # weights sum to one
    # against float64 bilinear with pixel-centre mapping: fixed point stays within 1 LSB
    for (h, w), (nw, nh) in (((37, 53), (80, 64)), ((64, 96), (48, 32)), ((9, 7), (10, 31))):
        src = _img(h, w, seed=h).astype(np.float64)
        fx = np.clip((np.arange(nw) + 0.5) * w / nw - 0.5, 0, w - 1)
        fy = np.clip((np.arange(nh) + 0.5) * h / nh - 0.5, 0, h - 1)
        x0, y0 = np.floor(fx).astype(int), np.floor(fy).astype(int)
        x1, y1 = np.minimum(x0 + 1, w - 1), np.minimum(y0 + 1, h - 1)
        ax, ay = (fx - x0)[None, :, None], (fy - y0)[:, None, None]
        ref = (src[y0][:, x0] * (1 - ax) + src[y0][:, x1] * ax) * (1 - ay) + (src[y1][:, x0] * (1 - ax) + src[y1][:, x1] * ax) * ay
        got = resize_linear_u8(src.astype(np.uint8), (nw, nh)).astype(np.float64)
        assert np.abs(got - ref).max() <= 1.0
    # exact 2x up-sampling of a ramp keeps the corners
    ramp = np.arange(16, dtype=np.uint8).reshape(1, 16, 1).repeat(4, 0).repeat(3, 2) * 10
    up = resize_linear_u8(ramp, (32, 8))
    assert up[0, 0, 0] == 0 and up[0, -1, 0] == 150


def test_area_resize_properties():
    from adaptiveisp_amd.val.loader import resize_area_u8
    im = _img(64, 96)
    half = resize_area_u8(im, (48, 32))
    ref = im.reshape(32, 2, 48, 2, 3).astype(np.float32).mean((1, 3))
    assert np.abs(half.astype(np.float32) - ref).max() <= 0.5                        # integer factor: block means
    flat = np.full((50, 70, 3), 201, np.uint8)
    assert (resize_area_u8(flat, (33, 17)) == 201).all()
    third = resize_area_u8(_img(30, 45, 2), (30, 20))                                # 1.5x: fractional footprints
    assert third.shape == (20, 30, 3)
    a = _img(30, 45, 2).astype(np.float32)
    assert abs(third.astype(np.float32).mean() - a.mean()) < 0.6                     # area filter preserves the mean


def test_letterbox_matches_reference_geometry():
    from adaptiveisp_amd.val.loader import letterbox
    im = _img(375, 500)
    out, ratio, (dw, dh) = letterbox(im, 512, color=(0, 0, 0), auto=False, scaleup=True)   # numbers of augmentations.py:111-141
    assert out.shape == (512, 512, 3) and ratio == (1.024, 1.024) and (dw, dh) == (0.0, 64.0)
    assert not out[:64].any() and not out[448:].any() and out[64:448].any()
    out, ratio, (dw, dh) = letterbox(_img(720, 1280), (736, 1280), auto=True)
    assert out.shape == (736, 1280, 3) and (out[:8] == 114).all() and (dw, dh) == (0.0, 8.0)
    out, ratio, _ = letterbox(_img(100, 200), 512, auto=False, scaleup=False)        # never scaled up for validation
    assert ratio == (1.0, 1.0) and out.shape == (512, 512, 3)


def test_lod_batches(tmp_path):
    from PIL import Image
    from adaptiveisp_amd.val import scale_boxes, xywh2xyxy
    from adaptiveisp_amd.val.loader import LODImages, imread_bgr
    root = tmp_path / "lod"
    (root / "images" / "val").mkdir(parents=True)
    (root / "labels" / "val").mkdir(parents=True)
    sizes = [(300, 400), (512, 384), (640, 1024)]
    for i, (h, w) in enumerate(sizes):
        Image.fromarray(_img(h, w, i)[:, :, ::-1].copy()).save(root / "images" / "val" / f"im{i}.png")
        if i != 1:
            (root / "labels" / "val" / f"im{i}.txt").write_text("3 0.5 0.5 0.2 0.4\n1 0.25 0.75 0.1 0.1\n")
    assert np.array_equal(imread_bgr(str(root / "images" / "val" / "im0.png")), _img(300, 400, 0))     # BGR round trip
    ds = LODImages(str(root / "images" / "val"), img_size=512, batch_size=2)
    batches = list(ds)
    assert len(ds) == 2 and [b[0].shape[0] for b in batches] == [2, 1]
    imgs, targets, paths, shapes = batches[0]
    assert imgs.shape == (2, 3, 512, 512) and imgs.dtype == torch.float32 and 0 <= float(imgs.min()) and float(imgs.max()) <= 1
    assert targets.shape == (2, 6) and (targets[:, 0] == 0).all()                       # image 1 has no label file
    (h0, w0), ((rh, rw), (dw, dh)) = shapes[0]
    assert (h0, w0) == (300, 400) and abs(rh - 384 / 300) < 1e-9 and (dw, dh) == (0.0, 64.0)
    # a label box mapped into the letterboxed frame and back through scale_boxes lands on the native box
    box = xywh2xyxy(targets[:1, 2:] * 512)
    scale_boxes((512, 512), box, (h0, w0), shapes[0][1])
    np.testing.assert_allclose(box[0].numpy(), [0.4 * 400, 0.3 * 300, 0.6 * 400, 0.7 * 300], atol=0.5)
    # RGB order: channel 0 of the tensor is the R plane = last channel of the BGR array
    src = _img(300, 400, 0)
    assert abs(float(imgs[0, 0, 64 + 5, 5]) * 255 - src[:, :, 2].astype(np.float32)[3:5, 3:5].mean()) < 60
    big = batches[1][0]
    assert big.shape == (1, 3, 512, 512) and not big[0, :, :96].any()                 # 640x1024 -> 320x512, 96 black rows above


def test_area_resize_integer_factors_round_like_resizeareafast():
    """Exact 2x shrink: OpenCV's uint8 INTER_AREA takes the integer ResizeAreaFast path, (sum + 2) >> 2 — ties round half
    UP, not to even; other integer factors cvRound(sum / area)."""
    from adaptiveisp_amd.val.loader import resize_area_u8
    im = np.zeros((4, 4, 1), np.uint8)
    im[0:2, 0:2, 0] = [[1, 1], [0, 0]]            # sum 2 -> 0.5 -> 1 (half-to-even would give 0)
    im[0:2, 2:4, 0] = [[3, 3], [2, 2]]            # sum 10 -> 2.5 -> 3 (half-to-even would give 2)
    im[2:4, 0:2, 0] = [[255, 255], [255, 254]]    # sum 1019 -> 254.75 -> 255
    out = resize_area_u8(im, (2, 2))[..., 0]
    assert out.tolist() == [[1, 3], [255, 0]]
    rng = np.random.default_rng(0)
    big = rng.integers(0, 256, (9, 12, 3), dtype=np.uint8)
    got = resize_area_u8(big, (4, 3))
    want = np.rint(big.astype(np.float32).reshape(3, 3, 4, 3, 3).sum(axis=(1, 3)) * np.float32(1 / 9)).astype(np.uint8)
    assert np.array_equal(got, want)
