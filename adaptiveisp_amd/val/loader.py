"""Image loader of the LOD evaluation: decode -> resize -> letterbox -> [B,3,H,W] fp32 RGB in [0,1]
with the `(targets, paths, shapes)` the eval loop consumes — the counterpart of `LoadImagesAndLabels.load_image`
(yolov3/utils/dataloaders.py:735-750), `letterbox` (yolov3/utils/augmentations.py:111-141) and
`LoadImagesAndLabelsNormalize.__getitem__` / `collate_fn` (dataset.py:597-668).

The reference does the pixel work with OpenCV, which this image does not have (and which is not on the GPU path), so
the two resampling kernels are restated here in numpy from OpenCV's published uint8 algorithms:
  * INTER_LINEAR: pixel-centre mapping, 11-bit fixed-point coefficients, horizontal pass into int32, vertical pass
    `(((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2`;
  * INTER_AREA (shrinking): every output pixel is the area-weighted mean of the source pixels its footprint covers,
    accumulated in fp32, rounded half-to-even.
PARITY UNPINNED against cv2 itself (absent: no fixture can be generated); the geometry (`letterbox_geometry`, sizes,
`shapes` tuples) is pinned by the reference's numbers in tests/test_eval_harness.py, the kernels by their defining
properties in tests/test_loader.py. Decoding is PIL's (libjpeg / libpng, as cv2.imread's).
"""
import glob
import functools
import math
import os

import numpy as np
import torch

from .boxes import letterbox_geometry

IMG_FORMATS = ("bmp", "jpeg", "jpg", "png", "tif", "tiff", "webp")
_COEF_BITS = 11
_ONE = 1 << _COEF_BITS


def imread_bgr(path):
    """cv2.imread(path): HWC uint8, BGR channel order."""
    from PIL import Image
    with Image.open(path) as im:
        rgb = np.asarray(im.convert("RGB"))
    return np.ascontiguousarray(rgb[:, :, ::-1])


def _linear_taps(src, dst):
    """Per destination index: left source index (clamped), and the two 11-bit weights."""
    scale = src / dst
    f = (np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5
    i0 = np.floor(f).astype(np.int64)
    frac = (f - i0).astype(np.float32)
    frac[i0 < 0] = 0.0
    i0 = np.maximum(i0, 0)
    edge = i0 >= src - 1
    frac[edge] = 0.0
    i0[edge] = src - 1
    w1 = np.rint(frac * np.float32(_ONE)).astype(np.int32)
    w0 = np.rint((np.float32(1.0) - frac) * np.float32(_ONE)).astype(np.int32)
    return i0, np.minimum(i0 + 1, src - 1), w0, w1


def resize_linear_u8(img, size):
    """cv2.resize(img, (w, h), interpolation=cv2.INTER_LINEAR) for HWC uint8."""
    w, h = int(size[0]), int(size[1])
    H, W = img.shape[:2]
    if (W, H) == (w, h):
        return img.copy()
    x0, x1, a0, a1 = _linear_taps(W, w)
    y0, y1, b0, b1 = _linear_taps(H, h)
    s = img.astype(np.int32)
    rows = s[:, x0] * a0[None, :, None] + s[:, x1] * a1[None, :, None]            # [H, w, C], scaled by 2^11
    top, bot = rows[y0] >> 4, rows[y1] >> 4
    out = (((b0[:, None, None] * top) >> 16) + ((b1[:, None, None] * bot) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


@functools.lru_cache(maxsize=64)
def _area_weights(src, dst):
    """Dense [dst, src] matrix of footprint overlaps, rows normalised to 1 (fp32); cached per (src, dst)."""
    scale = src / dst
    m = np.zeros((dst, src), np.float32)
    for d in range(dst):
        lo, hi = d * scale, (d + 1) * scale
        i0, i1 = int(math.floor(lo)), min(int(math.ceil(hi)), src)
        for i in range(i0, i1):
            m[d, i] = max(0.0, min(hi, i + 1) - max(lo, i))
        m[d] /= m[d].sum()
    return m


def resize_area_u8(img, size):
    """cv2.resize(img, (w, h), interpolation=cv2.INTER_AREA) for HWC uint8 when shrinking (the only use in
    load_image); growing falls back to the bilinear kernel as OpenCV does."""
    w, h = int(size[0]), int(size[1])
    H, W = img.shape[:2]
    if (W, H) == (w, h):
        return img.copy()
    if w > W or h > H:
        return resize_linear_u8(img, size)
    if W % w == 0 and H % h == 0:
        # integer shrink factors: OpenCV's ResizeAreaFast path — an INTEGER block sum, then (sum + 2) >> 2 for 2 x 2 blocks
        # (ties round half UP there) and cvRound(sum * (1.f / area)) otherwise
        fx, fy = W // w, H // h
        blk = img.astype(np.int32).reshape(h, fy, w, fx, -1).sum(axis=(1, 3))
        if fx == 2 and fy == 2:
            return ((blk + 2) >> 2).astype(np.uint8)
        return np.clip(np.rint(blk.astype(np.float32) * np.float32(1.0 / (fx * fy))), 0, 255).astype(np.uint8)
    mx, my = _area_weights(W, w), _area_weights(H, h)
    acc = np.einsum("yi,ijc->yjc", my, np.einsum("xj,ijc->ixc", mx, img.astype(np.float32)))
    return np.clip(np.rint(acc), 0, 255).astype(np.uint8)


def load_image(path, img_size, augment=False):
    """dataloaders.py:735-750: decode, then scale the LONGER side to img_size (area filter when shrinking for
    evaluation, bilinear otherwise). Returns (im BGR uint8, (h0, w0), (h, w))."""
    im = imread_bgr(path)
    h0, w0 = im.shape[:2]
    r = img_size / max(h0, w0)
    if r != 1:
        size = (math.ceil(w0 * r), math.ceil(h0 * r))
        im = resize_linear_u8(im, size) if (augment or r > 1) else resize_area_u8(im, size)
    return im, (h0, w0), im.shape[:2]


def letterbox(im, new_shape=(640, 640), color=(114, 114, 114), auto=True, scaleFill=False, scaleup=True, stride=32):
    """augmentations.py:111-141 on an HWC uint8 image: bilinear resize to the un-padded size, constant border.
    Returns (image, ratio (w, h), (dw, dh))."""
    ratio, new_unpad, (dw, dh), (top, bottom, left, right) = letterbox_geometry(im.shape[:2], new_shape, auto, scaleFill,
                                                                                 scaleup, stride)
    if (im.shape[1], im.shape[0]) != tuple(new_unpad):
        im = resize_linear_u8(im, new_unpad)
    out = np.empty((im.shape[0] + top + bottom, im.shape[1] + left + right, im.shape[2]), np.uint8)
    out[...] = np.asarray(color, np.uint8)
    out[top:top + im.shape[0], left:left + im.shape[1]] = im
    return out, ratio, (dw, dh)


def _label_path(image_path):
    """…/images/…/x.png -> …/labels/…/x.txt (dataloaders.py:428-432 img2label_paths)."""
    sa, sb = f"{os.sep}images{os.sep}", f"{os.sep}labels{os.sep}"
    return sb.join(image_path.rsplit(sa, 1)).rsplit(".", 1)[0] + ".txt"


def read_labels(path):
    """YOLO txt: rows of `class x y w h`, normalised. Missing file -> no objects."""
    if not os.path.isfile(path):
        return np.zeros((0, 5), np.float32)
    rows = [ln.split() for ln in open(path).read().strip().splitlines() if ln.strip()]
    lb = np.array(rows, dtype=np.float32).reshape(-1, 5) if rows else np.zeros((0, 5), np.float32)
    if lb.size and (lb[:, 1:] > 1).any():
        raise ValueError(f"{path}: non-normalised box coordinates")
    return lb


class LODImages:
    """Iterates a dataset the way the evaluation of val_adaptiveisp.py sees it (rect=False, augment=False, pad 0):
    every image is scaled so its longer side is `img_size`, letterboxed to img_size x img_size with a BLACK border
    (dataset.py:615 passes color=(0,0,0)), converted BGR->RGB, /255. Yields
        imgs [B,3,S,S] fp32, targets [n,6] (image index, class, xywh normalised to the letterboxed image), paths, shapes
    with shapes[i] = ((h0, w0), ((h/h0, w/w0), (dw, dh))) — what `scale_boxes` needs to map boxes back."""

    def __init__(self, source, img_size=512, batch_size=1, device="cpu"):
        if isinstance(source, (list, tuple)):
            files = list(source)
        elif os.path.isdir(source):
            files = sorted(glob.glob(os.path.join(source, "**", "*.*"), recursive=True))
        else:
            base = os.path.dirname(source)
            files = [ln.strip() for ln in open(source) if ln.strip()]
            files = [os.path.join(base, f[2:]) if f.startswith("./") else f for f in files]
        self.files = [f for f in files if f.rsplit(".", 1)[-1].lower() in IMG_FORMATS]
        if not self.files:
            raise FileNotFoundError(f"no images under {source}")
        self.img_size, self.batch_size, self.device = int(img_size), int(batch_size), device

    def __len__(self):
        return (len(self.files) + self.batch_size - 1) // self.batch_size

    def item(self, i):
        path = self.files[i]
        im, (h0, w0), (h, w) = load_image(path, self.img_size, augment=False)
        im, ratio, pad = letterbox(im, self.img_size, color=(0, 0, 0), auto=False, scaleup=False)
        shapes = (h0, w0), ((h / h0, w / w0), pad)
        lb = read_labels(_label_path(path)).copy()
        if lb.size:                                   # normalised xywh (native) -> pixels in the letterboxed frame -> normalised
            cx, cy = lb[:, 1] * (ratio[0] * w) + pad[0], lb[:, 2] * (ratio[1] * h) + pad[1]
            bw, bh = lb[:, 3] * (ratio[0] * w), lb[:, 4] * (ratio[1] * h)
            x1, y1, x2, y2 = cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2
            H, W = im.shape[:2]
            x1, x2 = np.clip(x1, 0, W - 1e-3), np.clip(x2, 0, W - 1e-3)
            y1, y2 = np.clip(y1, 0, H - 1e-3), np.clip(y2, 0, H - 1e-3)
            lb[:, 1], lb[:, 2] = (x1 + x2) / 2 / W, (y1 + y2) / 2 / H
            lb[:, 3], lb[:, 4] = (x2 - x1) / W, (y2 - y1) / H
        chw = np.ascontiguousarray(im.transpose(2, 0, 1)[::-1])          # HWC BGR -> CHW RGB
        return torch.from_numpy(chw).float() / 255.0, lb, path, shapes

    def __iter__(self):
        for s in range(0, len(self.files), self.batch_size):
            items = [self.item(i) for i in range(s, min(s + self.batch_size, len(self.files)))]
            imgs = torch.stack([it[0] for it in items]).to(self.device)
            tg = []
            for k, it in enumerate(items):
                t = torch.zeros((len(it[1]), 6))
                if len(it[1]):
                    t[:, 1:] = torch.from_numpy(it[1])
                t[:, 0] = k
                tg.append(t)
            yield imgs, torch.cat(tg, 0), [it[2] for it in items], [it[3] for it in items]
