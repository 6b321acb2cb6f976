"""Detection loss used as the RL reward signal (CIoU box + BCE objectness + BCE class, YOLOv5-style target
assignment). Restates the reference's `ComputeLossBatch` / `ComputeLoss` (yolov3/utils/loss.py:237-380,91-234),
`build_targets` (:320-380) and `bbox_iou(..., CIoU=True)` (yolov3/utils/metrics.py:222-260).

Small, gather-heavy tensors ([n_targets, 85]); stays at PyTorch level. Inputs are the three raw
head maps [B, na, ny, nx, 5+nc] (DetectionModel in train mode, or YoloEngine.raw_maps()).
"""
import math

import torch
import torch.nn.functional as F

# yolov3/data/hyps/hyp.scratch-low.yaml with the scaling applied by the reference's train.py:141-144
def default_hyp(nc=80, imgsz=512, nl=3):
    return dict(box=0.05 * 3 / nl, cls=0.5 * nc / 80 * 3 / nl, obj=1.0 * (imgsz / 640) ** 2 * 3 / nl,
                anchor_t=4.0, cls_pw=1.0, obj_pw=1.0, fl_gamma=0.0, label_smoothing=0.0)


def ciou(box1, box2, eps=1e-7):
    """Complete-IoU of xywh boxes [n,4] vs [n,4] -> [n,1]."""
    x1, y1, w1, h1 = box1.chunk(4, -1)
    x2, y2, w2, h2 = box2.chunk(4, -1)
    l1, r1, t1, b1 = x1 - w1 / 2, x1 + w1 / 2, y1 - h1 / 2, y1 + h1 / 2
    l2, r2, t2, b2 = x2 - w2 / 2, x2 + w2 / 2, y2 - h2 / 2, y2 + h2 / 2
    inter = (r1.minimum(r2) - l1.maximum(l2)).clamp(0) * (b1.minimum(b2) - t1.maximum(t2)).clamp(0)
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    cw = r1.maximum(r2) - l1.minimum(l2)
    ch = b1.maximum(b2) - t1.minimum(t2)
    c2 = cw ** 2 + ch ** 2 + eps
    rho2 = ((l2 + r2 - l1 - r1) ** 2 + (t2 + b2 - t1 - b1) ** 2) / 4
    v = (4 / math.pi ** 2) * (torch.atan(w2 / h2) - torch.atan(w1 / h1)).pow(2)
    with torch.no_grad():
        alpha = v / (v - iou + (1 + eps))
    return iou - (rho2 / c2 + v * alpha)


class DetectionLoss:
    """loss(preds, targets) -> (lbox*bs, lobj*bs, lcls*bs); targets [n,6] = (image, class, x, y, w, h) normalised."""

    balance3 = (4.0, 1.0, 0.4)

    def __init__(self, anchors_grid, nc=80, hyp=None, device="cpu"):
        self.anchors = anchors_grid.to(device)          # [nl, na, 2] in grid units (Detect.anchors)
        self.nl, self.na = self.anchors.shape[:2]
        self.nc = nc
        self.hyp = hyp or default_hyp(nc)
        self.device = torch.device(device)
        eps = self.hyp.get("label_smoothing", 0.0)
        self.cp, self.cn = 1.0 - 0.5 * eps, 0.5 * eps
        self.balance = list(self.balance3) if self.nl == 3 else [4.0, 1.0, 0.25, 0.06, 0.02][: self.nl]
        self._off = torch.tensor([[0, 0], [1, 0], [0, 1], [-1, 0], [0, -1]], device=self.device).float() * 0.5

    def assign(self, preds, targets):
        """Target assignment: every target is matched to the anchors whose w/h ratio is < anchor_t and replicated
        into the (up to 2) neighbouring cells its centre is closest to."""
        na, nt = self.na, targets.shape[0]
        ai = torch.arange(na, device=self.device).float().view(na, 1).repeat(1, nt)
        tgt = torch.cat((targets.repeat(na, 1, 1), ai[..., None]), 2)            # [na, nt, 7]
        out = []
        gain = torch.ones(7, device=self.device)
        for i in range(self.nl):
            anchors, shape = self.anchors[i], preds[i].shape
            gain[2:6] = torch.tensor(shape, device=self.device)[[3, 2, 3, 2]].float()
            t = tgt * gain
            if nt:
                ratio = t[..., 4:6] / anchors[:, None]
                t = t[torch.max(ratio, 1 / ratio).max(2)[0] < self.hyp["anchor_t"]]
                gxy = t[:, 2:4]
                gxi = gain[[2, 3]] - gxy
                j, k = ((gxy % 1 < 0.5) & (gxy > 1)).T
                l, m = ((gxi % 1 < 0.5) & (gxi > 1)).T
                sel = torch.stack((torch.ones_like(j), j, k, l, m))
                t = t.repeat((5, 1, 1))[sel]
                offsets = (torch.zeros_like(gxy)[None] + self._off[:, None])[sel]
            else:
                t, offsets = tgt[0], 0
            bc, gxy, gwh, a = t.chunk(4, 1)
            a, (b, c) = a.long().view(-1), bc.long().T
            gij = (gxy - offsets).long()
            gi, gj = gij.T
            out.append(dict(b=b, a=a, gj=gj.clamp_(0, shape[2] - 1), gi=gi.clamp_(0, shape[3] - 1),
                            box=torch.cat((gxy - gij, gwh), 1), anchors=anchors[a], cls=c))
        return out

    def __call__(self, preds, targets):
        lcls = torch.zeros(1, device=self.device)
        lbox = torch.zeros(1, device=self.device)
        lobj = torch.zeros(1, device=self.device)
        assigned = self.assign(preds, targets)
        for i, (pi, m) in enumerate(zip(preds, assigned)):
            tobj = torch.zeros(pi.shape[:4], dtype=pi.dtype, device=self.device)
            n = m["b"].shape[0]
            if n:
                pxy, pwh, _, pcls = pi[m["b"], m["a"], m["gj"], m["gi"]].split((2, 2, 1, self.nc), 1)
                pxy = pxy.sigmoid() * 2 - 0.5
                pwh = (pwh.sigmoid() * 2) ** 2 * m["anchors"]
                iou = ciou(torch.cat((pxy, pwh), 1), m["box"]).squeeze()
                lbox = lbox + (1.0 - iou).mean()
                tobj[m["b"], m["a"], m["gj"], m["gi"]] = iou.detach().clamp(0).type(tobj.dtype)
                if self.nc > 1:
                    t = torch.full_like(pcls, self.cn)
                    t[range(n), m["cls"]] = self.cp
                    lcls = lcls + F.binary_cross_entropy_with_logits(
                        pcls, t, pos_weight=torch.tensor([self.hyp["cls_pw"]], device=self.device))
            lobj = lobj + F.binary_cross_entropy_with_logits(
                pi[..., 4], tobj, pos_weight=torch.tensor([self.hyp["obj_pw"]], device=self.device)) * self.balance[i]
        bs = preds[0].shape[0]
        return lbox * self.hyp["box"] * bs, lobj * self.hyp["obj"] * bs, lcls * self.hyp["cls"] * bs


def per_sample_loss(loss_fn, preds, labels):
    """Per-image detection loss [B,1] (reference train.py:175-197: each sample is scored alone with its targets'
    image index set to 0). `labels` is a list of [n_b, 6] tensors."""
    B = preds[0].shape[0]
    rows = []
    for b in range(B):
        one = [p[b:b + 1] for p in preds]
        t = labels[b].clone().to(preds[0].device)
        t[:, 0] = 0
        lbox, lobj, lcls = loss_fn(one, t)
        rows.append(lbox + lobj + lcls)
    return torch.stack(rows).view(B, 1)


def assign_labels(loss_fn, preds, labels):
    """Target assignment of a labelled batch (depends on the labels and the SHAPES of the head maps only, not on their
    values): computed once per training iteration and shared by the loss of the input batch and of the retouched one."""
    dev = preds[0].device
    rows = []
    for b, lb in enumerate(labels):
        t = torch.as_tensor(lb).clone().to(dev).float()
        t[:, 0] = b
        rows.append(t)
    targets = torch.cat(rows, 0) if rows else torch.zeros((0, 6), device=dev)
    return loss_fn.assign(preds, targets)


def pack_assigned(assigned):
    """The target assignment in the form the fused loss kernels read (include/adayolo.h, adayolo_loss_layer): per layer
    idx int32 [n,5] = (image, anchor, gj, gi, class) and box fp32 [n,6] = (tx, ty, tw, th, anchor_w, anchor_h)."""
    packed = []
    for m in assigned:
        idx = torch.stack((m["b"], m["a"], m["gj"], m["gi"], m["cls"]), 1).to(torch.int32).contiguous()
        box = torch.cat((m["box"], m["anchors"]), 1).float().contiguous()
        packed.append((idx, box))
    return packed


def assign_labels_host(loss_fn, shapes, labels):
    """The target assignment of `pack_assigned(assign_labels(...))` computed on the HOST in numpy: per layer (idx int32 [n,5],
    box fp32 [n,6]). Same float32 arithmetic, same row order as DetectionLoss.assign (build_targets,
    yolov3/utils/loss.py:320-380); tests/test_yolo_cpu.py checks equality."""
    import numpy as np
    f32 = np.float32
    rows = []
    for b, lb in enumerate(labels):
        t = np.array(torch.as_tensor(lb).detach().cpu().numpy(), dtype=f32).reshape(-1, 6)
        t[:, 0] = b
        rows.append(t)
    targets = np.concatenate(rows, 0) if rows else np.zeros((0, 6), f32)
    anc = getattr(loss_fn, "_anchors_host", None)
    if anc is None:
        anc = loss_fn._anchors_host = loss_fn.anchors.detach().float().cpu().numpy()
    na, nt = loss_fn.na, targets.shape[0]
    ai = np.repeat(np.arange(na, dtype=f32)[:, None], nt, 1)
    tgt = np.concatenate((np.repeat(targets[None], na, 0), ai[..., None]), 2)                 # [na, nt, 7]
    off = np.array([[0, 0], [1, 0], [0, 1], [-1, 0], [0, -1]], f32) * f32(0.5)
    thr = f32(loss_fn.hyp["anchor_t"])
    host = []
    for i in range(loss_fn.nl):
        ny, nx = int(shapes[i].shape[2]), int(shapes[i].shape[3])
        gain = np.array([1, 1, nx, ny, nx, ny, 1], f32)
        t = tgt * gain
        if nt:
            ratio = t[..., 4:6] / anc[i][:, None]
            t = t[np.maximum(ratio, f32(1) / ratio).max(2) < thr]
            gxy = t[:, 2:4]
            gxi = gain[[2, 3]] - gxy
            j, k = ((gxy % f32(1) < f32(0.5)) & (gxy > f32(1))).T
            l, m = ((gxi % f32(1) < f32(0.5)) & (gxi > f32(1))).T
            sel = np.stack((np.ones_like(j), j, k, l, m))
            t = np.repeat(t[None], 5, 0)[sel]
            offsets = (np.zeros_like(gxy)[None] + off[:, None])[sel]
        else:
            t, offsets = tgt[0], f32(0)
        gxy, gwh = t[:, 2:4], t[:, 4:6]
        a = t[:, 6].astype(np.int64)
        gij = (gxy - offsets).astype(np.int64)
        # the reference clamps gi / gj IN PLACE (views of gij) before tbox = gxy - gij (loss.py:372-376): a label centred on the
        # map's far edge (x or y == 1.0) gets its offset against the CLAMPED cell
        gij[:, 0] = np.clip(gij[:, 0], 0, nx - 1)
        gij[:, 1] = np.clip(gij[:, 1], 0, ny - 1)
        idx = np.stack((t[:, 0].astype(np.int64), a, gij[:, 1], gij[:, 0], t[:, 1].astype(np.int64)), 1).astype(np.int32)
        box = np.concatenate((gxy - gij.astype(f32), gwh, anc[i][a]), 1).astype(f32)
        host.append((np.ascontiguousarray(idx), np.ascontiguousarray(box)))
    return host


def assign_labels_packed(loss_fn, shapes, labels, device, pair=False):
    """`pack_assigned(assign_labels(...))` computed on the HOST in numpy (assign_labels_host: the labels are a handful of host
    rows; as device-side PyTorch the assignment is ~200 launches and several synchronising boolean gathers, 2 ms per training
    iteration) and moved to `device` in one copy."""
    import numpy as np
    host = assign_labels_host(loss_fn, shapes, labels)
    packed, packed2 = [], []                # (`pair`: also the assignment of the batch [labels; labels], see below)
    # ONE upload for all layers (int32 words; the fp32 boxes travel as their bit patterns) from pinned memory: a copy per array
    # from pageable memory makes torch synchronise the stream each time — the training loop would drain the GPU every iteration
    B = len(labels)
    parts = []
    for idx, box in host:
        parts += [idx.reshape(-1)] + ([(idx + np.array([B, 0, 0, 0, 0], np.int32)).reshape(-1)] if pair else [])
        parts += [box.reshape(-1).view(np.int32)] * (2 if pair else 1)
    flat = np.concatenate(parts) if parts else np.zeros((0,), np.int32)
    dflat = torch.from_numpy(flat)
    if torch.device(device).type == "cuda":
        dflat = dflat.pin_memory().to(device, non_blocking=True)
    else:
        dflat = dflat.to(device)
    off = 0
    for idx, box in host:
        n = idx.shape[0]
        k = 2 if pair else 1
        d_idx = dflat[off:off + k * n * 5].view(k * n, 5)
        off += k * n * 5
        d_box = dflat[off:off + k * n * 6].view(torch.float32).view(k * n, 6)
        off += k * n * 6
        packed.append((d_idx[:n], d_box[:n]))
        packed2.append((d_idx, d_box))
    return (packed, packed2) if pair else packed


class StaticLabelTables:
    """The packed target assignment of an iteration in buffers of FIXED address and row count — what a training iteration
    captured in a hipGraph reads (train.Trainer, graph mode): per layer `cap` rows of the [labels; labels] pair form, the rows
    beyond the iteration's own filled with image index -1. The loss kernels walk a layer's list by image (csrc/yolo_loss.hip:
    `if (id[0] != b) continue`, the same-cell scans compare the image index first), so a row of image -1 is a row of no image:
    the results are those of the exact-size tables, bit for bit (tests/test_gpu_train_graph.py). `extra_words` int32 words
    behind the tables travel in the same upload (the iteration's device scalars)."""

    def __init__(self, loss_fn, shapes, batch, device, cap=512, extra_words=0):
        self.loss_fn, self.shapes, self.B, self.cap, self.nl = loss_fn, shapes, int(batch), int(cap), loss_fn.nl
        words = self.nl * self.cap * 11 + int(extra_words)
        self.host = torch.empty((words,), dtype=torch.int32, pin_memory=torch.device(device).type == "cuda")
        self.dev = torch.zeros((words,), dtype=torch.int32, device=device)
        self.extra_host = self.host[self.nl * self.cap * 11:]
        self.extra_dev = self.dev[self.nl * self.cap * 11:]
        self.packed_pair, self._hviews = [], []
        for i in range(self.nl):
            o = i * self.cap * 11
            self.packed_pair.append((self.dev[o:o + 5 * self.cap].view(self.cap, 5),
                                     self.dev[o + 5 * self.cap:o + 11 * self.cap].view(torch.float32).view(self.cap, 6)))
            self._hviews.append((self.host[o:o + 5 * self.cap].view(self.cap, 5).numpy(),
                                 self.host[o + 5 * self.cap:o + 11 * self.cap].view(torch.float32).view(self.cap, 6).numpy()))
        # the B-image engine of the pair (the backward over the retouched half) reads the SAME tables: the rows of images
        # [B, 2B) are rows of no image of its own
        self.packed = self.packed_pair

    def fill(self, labels):
        """Assign `labels` (B host label sets) on the host and write the pair tables into the staging block. False — nothing
        written — when a layer has more rows than `cap` (the caller runs that iteration uncaptured or rebuilds with a larger cap)."""
        host = assign_labels_host(self.loss_fn, self.shapes, labels)
        if any(2 * idx.shape[0] > self.cap for idx, _ in host):
            return False
        import numpy as np
        for (idx, box), (hi, hb) in zip(host, self._hviews):
            n = idx.shape[0]
            hi[:n] = idx
            hi[n:2 * n] = idx + np.array([self.B, 0, 0, 0, 0], np.int32)
            hi[2 * n:] = -1
            hb[:n] = box
            hb[n:2 * n] = box
            hb[2 * n:] = 0.0
        return True

    def upload(self):
        """One copy of the host block (tables + extra words) to the device, on the current stream (capturable: the copy node of
        a graph reads the pinned block at every replay)."""
        self.dev.copy_(self.host, non_blocking=True)


def batched_per_sample_loss(loss_fn, preds, labels, assigned=None):
    """The same [B,1] per-image losses as `per_sample_loss` in ONE batched pass (the reference loops over the samples
    in Python, train.py:184-196: 2*B loss evaluations of ~150 tiny launches each per iteration). Every reduction the
    per-sample call does over "its" batch of one becomes a per-image segment reduction here:
      lbox_b = mean over image b's matched targets of (1 - CIoU);  lcls_b = mean over (matches of b) x nc of BCE;
      lobj_b = mean over image b's (na, ny, nx) cells of BCE, weighted by the layer balance."""
    dev = preds[0].device
    B = preds[0].shape[0]
    if assigned is None:
        assigned = assign_labels(loss_fn, preds, labels)
    pw = getattr(loss_fn, "_pw_cache", None)                 # the two pos_weight scalars as device tensors, made once
    if pw is None or pw[0].device != dev:
        pw = (torch.tensor([loss_fn.hyp["cls_pw"]], device=dev), torch.tensor([loss_fn.hyp["obj_pw"]], device=dev))
        loss_fn._pw_cache = pw
    lbox = torch.zeros(B, device=dev)
    lobj = torch.zeros(B, device=dev)
    lcls = torch.zeros(B, device=dev)
    for i, (pi, m) in enumerate(zip(preds, assigned)):
        tobj = torch.zeros(pi.shape[:4], dtype=pi.dtype, device=dev)
        n = m["b"].shape[0]
        if n:
            bidx = m["b"]
            cnt = torch.zeros(B, device=dev).index_add_(0, bidx, torch.ones(n, device=dev))
            inv = torch.where(cnt > 0, 1.0 / cnt.clamp(min=1.0), torch.zeros_like(cnt))
            pxy, pwh, _, pcls = pi[bidx, m["a"], m["gj"], m["gi"]].split((2, 2, 1, loss_fn.nc), 1)
            pxy = pxy.sigmoid() * 2 - 0.5
            pwh = (pwh.sigmoid() * 2) ** 2 * m["anchors"]
            iou = ciou(torch.cat((pxy, pwh), 1), m["box"]).reshape(-1)
            lbox = lbox + torch.zeros(B, device=dev).index_add_(0, bidx, 1.0 - iou) * inv
            tobj[bidx, m["a"], m["gj"], m["gi"]] = iou.detach().clamp(0).type(tobj.dtype)
            if loss_fn.nc > 1:
                t = torch.full_like(pcls, loss_fn.cn)
                t[range(n), m["cls"]] = loss_fn.cp
                bce = F.binary_cross_entropy_with_logits(pcls, t, reduction="none", pos_weight=pw[0])
                lcls = lcls + torch.zeros(B, device=dev).index_add_(0, bidx, bce.sum(1)) * inv / loss_fn.nc
        obj = F.binary_cross_entropy_with_logits(pi[..., 4], tobj, reduction="none", pos_weight=pw[1])
        lobj = lobj + obj.mean(dim=(1, 2, 3)) * loss_fn.balance[i]
    total = lbox * loss_fn.hyp["box"] + lobj * loss_fn.hyp["obj"] + lcls * loss_fn.hyp["cls"]
    return total.view(B, 1)
