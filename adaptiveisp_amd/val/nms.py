"""non_max_suppression of the eval loop (yolov3/utils/general.py:856-966) with the greedy NMS itself on the GPU
(csrc/yolo_nms.hip through the C-ABI `adayolo_nms`, replacing `torchvision.ops.nms` :949).

Differences from the reference, deliberate: no wall-clock time limit (general.py:891,962-964 aborts after
0.5 s + 0.05 s/image, which makes results machine-dependent), no merge-NMS branch (dead code there, `merge = False`),
and the candidate selection runs once for the batch instead of once per image.
"""
import ctypes

import torch

from .boxes import xywh2xyxy

MAX_WH = 7680      # class offset (pixels), general.py:888
MAX_NMS = 30000    # boxes entering NMS, general.py:889


def hip_nms(boxes, scores, iou_thres, max_det=300, presorted=False):
    """Kept indices (int64, device) of greedy IoU NMS. `boxes` [n,4] xyxy fp32 on a HIP device, any score order
    (`presorted`: the caller hands them over in descending score order already — no second sort, no gather)."""
    from ..yolo import _lib
    if boxes.device.type != "cuda":
        raise _lib.AdayoloError("hip_nms needs device tensors: there is no CPU path")
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros(0, dtype=torch.int64, device=boxes.device)
    order = None if presorted else scores.argsort(descending=True, stable=True)
    b = boxes.float().contiguous() if presorted else boxes.float()[order].contiguous()
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
    kept = keep[:k].long()
    return kept if presorted else order[kept]


def _with_label_rows(pred, labels, nc, nm):
    """Autolabelling priors (general.py:911-918): per image, rows (xywh of the label, obj 1, its class 1) are added to
    the candidates unconditionally. Returns the padded prediction and a mask of the rows that are such priors."""
    B, N, C = pred.shape
    extra = max((len(lb) for lb in labels), default=0)
    forced = torch.zeros((B, N + extra), dtype=torch.bool, device=pred.device)
    if extra == 0:
        return pred, forced
    pred = torch.cat((pred, pred.new_zeros((B, extra, C))), 1)
    for b, lb in enumerate(labels):
        k = len(lb)
        if k:
            rows = pred[b, N:N + k]
            rows[:, :4] = lb[:, 1:5]
            rows[:, 4] = 1.0
            rows[torch.arange(k), lb[:, 0].long() + 5] = 1.0
            forced[b, N:N + k] = True
    return pred, forced


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False,
                        labels=(), max_det=300, nm=0, nms_fn=None):
    """prediction [B, N, 5+nc(+nm)] (xywh, obj, cls...) -> list of [n,6(+nm)] (xyxy, conf, cls) per image — the
    results of yolov3/utils/general.py:856-966 (pinned by tests/golden/evalharness.npz).

    The candidates of the WHOLE batch are extracted once (one threshold, one gather, one score product, one box
    conversion); images are then contiguous segments of that list, each sorted by score and handed to the greedy NMS
    with the per-class coordinate offset. `nms_fn(boxes, scores, iou_thres)` defaults to the HIP kernel; tests inject
    the CPU oracle."""
    if not 0 <= conf_thres <= 1:
        raise ValueError(f"confidence threshold {conf_thres} outside [0, 1]")
    if not 0 <= iou_thres <= 1:
        raise ValueError(f"IoU threshold {iou_thres} outside [0, 1]")
    if isinstance(prediction, (list, tuple)):
        prediction = prediction[0]
    if nms_fn is None:
        nms_fn = lambda b, s, t: hip_nms(b, s, t, max_det, presorted=True)       # noqa: E731  (every segment below is sorted)
    B = prediction.shape[0]
    nc = prediction.shape[2] - nm - 5
    first_mask = 5 + nc
    candidate = prediction[..., 4] > conf_thres
    if labels and any(len(lb) for lb in labels):
        prediction, forced = _with_label_rows(prediction, labels, nc, nm)
        candidate = torch.cat((candidate, candidate.new_zeros((B, forced.shape[1] - candidate.shape[1]))), 1) | forced
    image, row = candidate.nonzero(as_tuple=True)                  # image-major: every image is one contiguous run
    cand = prediction[image, row]
    scaled = cand[:, 5:] * cand[:, 4:5]                            # class scores (and mask coefficients) x objectness
    scores, coeff = scaled[:, :nc], scaled[:, nc:]
    box = xywh2xyxy(cand[:, :4])
    if multi_label and nc > 1:                                     # one detection per (box, class) above the threshold
        r, c = (scores > conf_thres).nonzero(as_tuple=True)
        det = torch.cat((box[r], scores[r, c, None], c[:, None].float(), coeff[r]), 1)
        image = image[r]
    else:                                                          # best class only
        conf, c = scores.max(1, keepdim=True)
        keep = conf.view(-1) > conf_thres
        det = torch.cat((box, conf, c.float(), coeff), 1)[keep]
        image = image[keep]
    if classes is not None:
        wanted = (det[:, 5:6] == torch.tensor(classes, device=det.device)).any(1)
        det, image = det[wanted], image[wanted]
    # (one image: its segment is the whole list — no count kernel, no host read)
    counts = [int(det.shape[0])] if B == 1 else torch.bincount(image, minlength=B).tolist()
    output, start = [], 0
    for n in counts:
        seg = det[start:start + n]
        start += n
        if n == 0:
            output.append(torch.zeros((0, 6 + nm), device=prediction.device))
            continue
        # stable: equal scores keep candidate order, so the CPU and the GPU run of the loop see the same list (the
        # reference's argsort leaves the order of ties to the backend)
        seg = seg[seg[:, 4].argsort(descending=True, stable=True)[:MAX_NMS]]
        offset = seg[:, 5:6] * (0 if agnostic else MAX_WH)         # classes never suppress each other
        kept = nms_fn(seg[:, :4] + offset, seg[:, 4], iou_thres)[:max_det]
        output.append(seg[kept])
    return output
