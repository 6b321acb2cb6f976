"""Detection metrics of the eval loop: IoU matrix, prediction/label matching at the 10 IoU levels, 101-point AP.

Same results as the reference's yolov3/utils/metrics.py (`ap_per_class` :31-96, `compute_ap` :98-123, `box_iou`
:262-280, `smooth` :21-28) and yolov3/val_adaptiveisp.py:79-103 (`process_batch`) — pinned bit-exactly by
tests/golden/evalharness.npz — but organised around what the computation IS rather than how the reference spells it:

  * matching: a detection can only ever be credited to its best same-class label (that pairing does not depend on the
    IoU level), and a label keeps the lowest-index detection that claims it — so all 10 levels come from one arg-max and
    one scatter-min instead of a per-level sort / unique / unique;
  * AP: detections are grouped by class with one stable sort; per-class cumulative TP/FP are segment cumsums over the
    whole array (exact integers), all IoU levels at once; only numpy's own interpolation runs per class.
"""
import numpy as np
import torch


def box_iou(box1, box2, eps=1e-7):
    """IoU matrix [N,M] of xyxy boxes (torch)."""
    lt = torch.maximum(box1[:, None, :2], box2[None, :, :2])
    rb = torch.minimum(box1[:, None, 2:4], box2[None, :, 2:4])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    area1 = (box1[:, 2] - box1[:, 0]) * (box1[:, 3] - box1[:, 1])
    area2 = (box2[:, 2] - box2[:, 0]) * (box2[:, 3] - box2[:, 1])
    return inter / (area1[:, None] + area2[None, :] - inter + eps)


def process_batch(detections, labels, iouv):
    """detections [N,6] (xyxy, conf, cls), labels [M,5] (cls, xyxy), iouv [T] ascending -> bool [N,T]: detection n counts
    as a true positive at level t.

    Rule (what the reference's sort-by-IoU + unique-by-detection + unique-by-label amounts to): detection d claims the
    same-class label with which it has the highest IoU, provided that IoU reaches the level; of the detections claiming
    one label, the one with the lowest index (= highest confidence after NMS) is credited."""
    N, M, T = detections.shape[0], labels.shape[0], iouv.shape[0]
    correct = torch.zeros((N, T), dtype=torch.bool, device=iouv.device)
    if N == 0 or M == 0:
        return correct
    iou = box_iou(labels[:, 1:], detections[:, :4])                                # [M,N]
    iou = torch.where(labels[:, 0:1] == detections[:, 5], iou, iou.new_full((), -1.0))
    best_iou, best_label = iou.max(dim=0)                                          # per detection
    claims = best_iou[:, None] >= iouv.to(best_iou.device)[None, :]                # [N,T]
    det_index = torch.arange(N, device=iou.device)[:, None].expand(N, T)
    first = torch.full((M, T), N, dtype=torch.int64, device=iou.device)
    first.scatter_reduce_(0, best_label[:, None].expand(N, T), torch.where(claims, det_index, N), reduce="amin")
    credited = claims & (first[best_label] == det_index)
    return credited.to(iouv.device)


def smooth(y, f=0.05):
    """Box filter over a fraction f of the curve, edges replicated."""
    nf = round(len(y) * f * 2) // 2 + 1            # odd window
    return np.convolve(np.pad(y, nf // 2, mode="edge"), np.full(nf, 1.0) / nf, mode="valid")


def _envelope(precision):
    """Monotone (non-increasing) envelope along axis 0: p'[i] = max(p[i:])."""
    return np.maximum.accumulate(precision[::-1], axis=0)[::-1]


_trapz = getattr(np, "trapezoid", None) or np.trapz          # numpy 2 renamed trapz
_GRID101 = np.linspace(0, 1, 101)


def compute_ap(recall, precision):
    """101-point interpolated AP (COCO style) of one precision/recall curve -> (ap, envelope precision, recall) with
    the sentinels (0,1) and (1,0) attached."""
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = _envelope(np.concatenate(([1.0], precision, [0.0])))
    return _trapz(np.interp(_GRID101, mrec, mpre), _GRID101), mpre, mrec


def ap_per_class(tp, conf, pred_cls, target_cls, eps=1e-16):
    """tp [n,T] bool, conf [n], pred_cls [n], target_cls [m] -> (tp, fp, p, r, f1, ap [nc,T], classes) over the
    classes present in the targets; p/r/f1 are taken at the confidence that maximises the smoothed mean F1."""
    classes, n_labels = np.unique(target_cls, return_counts=True)
    nc, T = classes.shape[0], tp.shape[1]
    grid = np.linspace(0, 1, 1000)
    ap, p_curve, r_curve = np.zeros((nc, T)), np.zeros((nc, 1000)), np.zeros((nc, 1000))
    # one ordering: by class, then by descending confidence inside a class
    by_conf = np.argsort(-conf)
    order = by_conf[np.argsort(pred_cls[by_conf], kind="stable")]
    cls_sorted, conf_sorted, hits = pred_cls[order], conf[order], tp[order].astype(np.int64)
    lo = np.searchsorted(cls_sorted, classes, side="left")
    hi = np.searchsorted(cls_sorted, classes, side="right")
    run = np.cumsum(hits, axis=0)                                   # running TP count over the whole ordering
    for ci in range(nc):
        a, b = lo[ci], hi[ci]
        if a == b or n_labels[ci] == 0:
            continue
        tpc = run[a:b] - (run[a - 1] if a else 0)                   # segment cumsum: TP so far inside this class
        fpc = np.arange(1, b - a + 1)[:, None] - tpc                # everything seen so far that was not a TP
        recall = tpc / (n_labels[ci] + eps)
        precision = tpc / (tpc + fpc)
        c = conf_sorted[a:b]
        r_curve[ci] = np.interp(-grid, -c, recall[:, 0], left=0)     # curves at IoU level 0 (mAP@0.5) over confidence
        p_curve[ci] = np.interp(-grid, -c, precision[:, 0], left=1)
        for j in range(T):
            ap[ci, j] = compute_ap(recall[:, j], precision[:, j])[0]
    f1_curve = 2 * p_curve * r_curve / (p_curve + r_curve + eps)
    k = smooth(f1_curve.mean(0), 0.1).argmax()
    p, r, f1 = p_curve[:, k], r_curve[:, k], f1_curve[:, k]
    tp_count = (r * n_labels).round()
    fp_count = (tp_count / (p + eps) - tp_count).round()
    return tp_count, fp_count, p, r, f1, ap, classes.astype(int)
