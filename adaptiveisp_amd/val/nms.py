"""non_max_suppression of the eval loop (yolov3/utils/general.py:856-966) with the greedy NMS itself on the GPU
(csrc/yolo_nms.hip through the C-ABI `adayolo_nms`, replacing `torchvision.ops.nms` :949).

Differences from the reference, deliberate: no wall-clock time limit (general.py:891,962-964 aborts after
0.5 s + 0.05 s/image, which makes results machine-dependent) and no merge-NMS branch (dead code there, `merge = False`).
"""
import ctypes

import torch

from .boxes import xywh2xyxy

MAX_WH = 7680      # class offset (pixels), general.py:888
MAX_NMS = 30000    # boxes entering NMS, general.py:889


def hip_nms(boxes, scores, iou_thres, max_det=300):
    """Kept indices (int64, device) of greedy IoU NMS. `boxes` [n,4] xyxy fp32 on a HIP device, any score order."""
    from ..yolo import _lib
    if boxes.device.type != "cuda":
        raise _lib.AdayoloError("hip_nms needs device tensors: there is no CPU path")
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros(0, dtype=torch.int64, device=boxes.device)
    order = scores.argsort(descending=True, stable=True)
    b = boxes.float()[order].contiguous()
    L = _lib.load()
    ws = torch.empty(L.adayolo_nms_workspace_bytes(n), dtype=torch.uint8, device=boxes.device)
    keep = torch.empty(max_det, dtype=torch.int32, device=boxes.device)
    cnt = torch.empty(1, dtype=torch.int32, device=boxes.device)
    with torch.cuda.device(boxes.device):
        rc = L.adayolo_nms(ctypes.c_void_p(b.data_ptr()), n, float(iou_thres), int(max_det),
                           ctypes.c_void_p(ws.data_ptr()), ctypes.c_void_p(keep.data_ptr()),
                           ctypes.c_void_p(cnt.data_ptr()), _lib.stream_ptr())
    _lib.check(rc, "adayolo_nms")
    k = int(cnt.item())
    return order[keep[:k].long()]


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False,
                        labels=(), max_det=300, nm=0, nms_fn=None):
    """prediction [B, N, 5+nc(+nm)] (xywh, obj, cls...) -> list of [n,6(+nm)] (xyxy, conf, cls) per image.
    `nms_fn(boxes, scores, iou_thres)` defaults to the HIP kernel; tests inject the CPU oracle."""
    assert 0 <= conf_thres <= 1, f"Invalid Confidence threshold {conf_thres}, valid values are between 0.0 and 1.0"
    assert 0 <= iou_thres <= 1, f"Invalid IoU {iou_thres}, valid values are between 0.0 and 1.0"
    if isinstance(prediction, (list, tuple)):
        prediction = prediction[0]
    if nms_fn is None:
        nms_fn = lambda b, s, t: hip_nms(b, s, t, max_det)       # noqa: E731
    bs = prediction.shape[0]
    nc = prediction.shape[2] - nm - 5
    xc = prediction[..., 4] > conf_thres
    multi_label &= nc > 1
    mi = 5 + nc
    output = [torch.zeros((0, 6 + nm), device=prediction.device)] * bs
    for xi, x in enumerate(prediction):
        x = x[xc[xi]]
        if labels and len(labels[xi]):
            lb = labels[xi]
            v = torch.zeros((len(lb), nc + nm + 5), device=x.device)
            v[:, :4] = lb[:, 1:5]
            v[:, 4] = 1.0
            v[range(len(lb)), lb[:, 0].long() + 5] = 1.0
            x = torch.cat((x, v), 0)
        if not x.shape[0]:
            continue
        x[:, 5:] *= x[:, 4:5]                                      # conf = obj_conf * cls_conf
        box = xywh2xyxy(x[:, :4])
        mask = x[:, mi:]
        if multi_label:
            i, j = (x[:, 5:mi] > conf_thres).nonzero(as_tuple=False).T
            x = torch.cat((box[i], x[i, 5 + j, None], j[:, None].float(), mask[i]), 1)
        else:
            conf, j = x[:, 5:mi].max(1, keepdim=True)
            x = torch.cat((box, conf, j.float(), mask), 1)[conf.view(-1) > conf_thres]
        if classes is not None:
            x = x[(x[:, 5:6] == torch.tensor(classes, device=x.device)).any(1)]
        n = x.shape[0]
        if not n:
            continue
        x = x[x[:, 4].argsort(descending=True)[:MAX_NMS]]
        c = x[:, 5:6] * (0 if agnostic else MAX_WH)
        boxes, scores = x[:, :4] + c, x[:, 4]
        i = nms_fn(boxes, scores, iou_thres)
        i = i[:max_det]
        output[xi] = x[i]
    return output
