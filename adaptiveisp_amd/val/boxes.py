"""Box geometry used by the eval loop. Restates yolov3/utils/general.py:722-739 (xyxy2xywh / xywh2xyxy),
:799-812 (scale_boxes), :834-843 (clip_boxes) and the padding arithmetic of letterbox
(yolov3/utils/augmentations.py:111-141). The reference resizes with cv2 (absent here, and not on the GPU path);
`letterbox_pad` only pads — resizing to the network size is the data loader's business."""
import numpy as np
import torch


def xyxy2xywh(x):
    y = x.clone() if isinstance(x, torch.Tensor) else np.copy(x)
    y[..., 0] = (x[..., 0] + x[..., 2]) / 2
    y[..., 1] = (x[..., 1] + x[..., 3]) / 2
    y[..., 2] = x[..., 2] - x[..., 0]
    y[..., 3] = x[..., 3] - x[..., 1]
    return y


def xywh2xyxy(x):
    y = x.clone() if isinstance(x, torch.Tensor) else np.copy(x)
    y[..., 0] = x[..., 0] - x[..., 2] / 2
    y[..., 1] = x[..., 1] - x[..., 3] / 2
    y[..., 2] = x[..., 0] + x[..., 2] / 2
    y[..., 3] = x[..., 1] + x[..., 3] / 2
    return y


def clip_boxes(boxes, shape):
    """In place: clip xyxy boxes to an image of (height, width)."""
    if isinstance(boxes, torch.Tensor):
        boxes[..., 0].clamp_(0, shape[1])
        boxes[..., 1].clamp_(0, shape[0])
        boxes[..., 2].clamp_(0, shape[1])
        boxes[..., 3].clamp_(0, shape[0])
    else:
        boxes[..., [0, 2]] = boxes[..., [0, 2]].clip(0, shape[1])
        boxes[..., [1, 3]] = boxes[..., [1, 3]].clip(0, shape[0])


def scale_boxes(img1_shape, boxes, img0_shape, ratio_pad=None):
    """In place: map xyxy boxes from the letterboxed network image (img1_shape) back to the native image."""
    if ratio_pad is None:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad = (img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2
    else:
        gain = ratio_pad[0][0]
        pad = ratio_pad[1]
    boxes[..., [0, 2]] -= pad[0]
    boxes[..., [1, 3]] -= pad[1]
    boxes[..., :4] /= gain
    clip_boxes(boxes, img0_shape)
    return boxes


def letterbox_geometry(shape, new_shape=(640, 640), auto=True, scale_fill=False, scaleup=True, stride=32):
    """The numbers letterbox() derives from an image of `shape` (h, w): (ratio (w,h), new_unpad (w,h), (dw, dh),
    (top, bottom, left, right))."""
    if isinstance(new_shape, int):
        new_shape = (new_shape, new_shape)
    r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
    if not scaleup:
        r = min(r, 1.0)
    ratio = r, r
    new_unpad = int(round(shape[1] * r)), int(round(shape[0] * r))
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
    if auto:
        dw, dh = np.mod(dw, stride), np.mod(dh, stride)
    elif scale_fill:
        dw, dh = 0.0, 0.0
        new_unpad = (new_shape[1], new_shape[0])
        ratio = new_shape[1] / shape[1], new_shape[0] / shape[0]
    dw /= 2
    dh /= 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return ratio, new_unpad, (dw, dh), (top, bottom, left, right)


def letterbox_pad(im, new_shape=(640, 640), color=(114, 114, 114), auto=True, stride=32):
    """Pad an HWC image that already has its un-padded target size (no resize) with `color` like letterbox does."""
    ratio, new_unpad, (dw, dh), (top, bottom, left, right) = letterbox_geometry(im.shape[:2], new_shape, auto,
                                                                                 False, False, stride)
    if (im.shape[1], im.shape[0]) != new_unpad:
        raise ValueError("letterbox_pad does not resize: bring the image to its un-padded size first")
    out = np.empty((im.shape[0] + top + bottom, im.shape[1] + left + right, im.shape[2]), im.dtype)
    out[...] = np.asarray(color, im.dtype)
    out[top:top + im.shape[0], left:left + im.shape[1]] = im
    return out, ratio, (dw, dh)
