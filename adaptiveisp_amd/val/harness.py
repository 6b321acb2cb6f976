"""Evaluation loop — the counterpart of `run()` in yolov3/val_adaptiveisp.py:105-460 for the HIP path:
per batch, `steps` RL steps of the ISP (early exit when image 0 reports `stopped`, :308), the detector forward,
NMS, per-image matching, then mAP over the whole set. Dataset I/O is out of scope: the
caller hands over an iterable of (images [B,3,H,W] fp32 in [0,1], targets [n,6] = (image, class, x, y, w, h)
normalised, paths, shapes) — the tuple the reference's dataloader yields."""
import os

import numpy as np
import torch

from ..util import STATE_STOPPED_DIM
from .boxes import scale_boxes, xywh2xyxy
from .metrics import ap_per_class, process_batch
from .nms import non_max_suppression


_SIDE = {}


def _side_stream(dev):
    key = str(dev)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=dev)
    return _SIDE[key]


class _EpisodeGraph:
    """One batch's ISP episode + detector forward as ONE hipGraph replay (run_eval(graph=True)). At batch 1 the loop is bound by
    the host — ~60 launches and one device-to-host read per RL step for 2 ms of kernels — so the fixed-shape part is captured
    once per (batch, H, W, steps, pipeline): `steps` x Agent.forward, then the detector, with the per-step selections and image
    0's `stopped` flags gathered on the device and read in ONE copy after the replay. The reference's early exit
    (val_adaptiveisp.py:302-303) can only be honoured afterwards: if a flag is set before the last step the caller redoes that
    batch with the eager loop (with the default states, `stopped` is raised on the last step only)."""

    def __init__(self, agent, detector, im, noises, states, steps, pipeline):
        dev = im.device
        # the agent and the detector engine work on buffers of their own: nothing of theirs may be in flight — another shape's
        # replay on the side stream — while this warm-up and capture run them
        torch.cuda.synchronize(dev)
        self.im, self.z, self.s0 = im.clone(), noises.clone(), states.clone()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():       # warm-up outside the capture (lazy initialisation, workspaces)
            self._body(agent, detector, steps, pipeline)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.retouch, self.host_view, self.preds = self._body(agent, detector, steps, pipeline)

    def _body(self, agent, detector, steps, pipeline):
        retouch, states, sel, stop = self.im, self.s0, [], []
        for i in range(steps):
            pipe = None if pipeline is None else pipeline[i]
            (retouch, states, _, _), dbg, _ = agent((retouch, self.z[i], states), 1.0, None, pipe)
            sel.append(dbg["selected_filter"].to(torch.float32))
            stop.append(states[0:1, STATE_STOPPED_DIM].to(torch.float32))
        preds = detector(retouch)
        return retouch, torch.cat(sel + stop), preds

    def run(self, im, noises, states):
        self.im.copy_(im); self.z.copy_(noises); self.s0.copy_(states)
        self.graph.replay()
        return self.host_view.cpu().tolist()

    # ---- two-stage form (run_eval(graph=True) pipelines the batches: the replay of batch i + 1 runs on a side stream while the
    #      host does NMS / matching of batch i): `launch` enqueues copy-in -> replay -> copy-out into result slot i % 2 on the
    #      side stream and returns at once; `result` waits for that slot's event only. The graph's own output tensors are
    #      overwritten by the next replay, so what the back half needs is copied out behind the replay: the predictions (5.5 MB at
    #      1 x 512 x 512), the host words (pinned), the retouched batch only if the caller wants it.
    def launch(self, im, noises, states, keep_retouch=False):
        dev = self.im.device
        if not hasattr(self, "_side"):
            # ONE side stream per device for every graph (a second batch shape has its own graph, but the same agent and detector
            # buffers underneath: their replays must not overlap each other either)
            self._side = _side_stream(dev)
            self._slots, self._n = [None, None], 0
        k = self._n % 2
        self._n += 1
        cur = torch.cuda.current_stream(dev)
        slot = self._slots[k]
        if slot is None:
            slot = self._slots[k] = dict(preds=torch.empty_like(self.preds), retouch=torch.empty_like(self.retouch),
                                         host=torch.empty(self.host_view.shape, dtype=self.host_view.dtype, pin_memory=True),
                                         done=torch.cuda.Event(), free=None)
        self._side.wait_stream(cur)                           # the uploads of im / noises / states were enqueued on `cur`
        if slot["free"] is not None:
            self._side.wait_event(slot["free"])               # ... and the back half of batch i - 2 is done with this slot
        with torch.cuda.stream(self._side):
            self.im.copy_(im); self.z.copy_(noises); self.s0.copy_(states)
            self.graph.replay()
            slot["preds"].copy_(self.preds)
            if keep_retouch:
                slot["retouch"].copy_(self.retouch)
            slot["host"].copy_(self.host_view, non_blocking=True)
            slot["done"].record(self._side)
        for t in (im, noises, states):
            t.record_stream(self._side)
        return slot

    @staticmethod
    def result(slot):
        """Host words of a launched batch (blocks until ITS replay and copies are done, not the stream's later work); the
        current stream may then read slot["preds"] / slot["retouch"]."""
        slot["done"].synchronize()
        torch.cuda.current_stream(slot["preds"].device).wait_event(slot["done"])
        return slot["host"].tolist()

    @staticmethod
    def release(slot):
        """The back half is done reading the slot (recorded on the current stream): the replay two batches on may overwrite it."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(slot["preds"].device))
        slot["free"] = ev


def run_eval(agent, detector, batches, cfg, steps=5, conf_thres=0.001, iou_thres=0.6, max_det=300, single_cls=False,
             pipeline=None, records_path=None, nc=80, nms_fn=None, param_dir=None, details=None, graph=False):
    """Returns dict(mp, mr, map50, map75, map, seen, nt, ap_class, ap, records). `detector(x)` -> [B, N, 5+nc]
    decoded predictions (YoloEngine or the module tree in eval mode). `pipeline`: optional list of forced filter ids
    per step (val_adaptiveisp.py:292, --pipeline). `param_dir`: write one JSON per batch (named after its first image) with
    the chosen filter ids and image 0's regressed parameters per step, as `--save_param` does (:296-301,324-327).
    `details`: a list that receives one dict per image (path, retouched image, detections after NMS, `correct` matrix).
    `graph`: replay each batch's ISP episode + detector forward as one hipGraph (captured once per batch shape; HIP device,
    eval-mode agent, no `param_dir`) — same records, detections and mAP as the eager loop, which stays the default."""
    import collections
    import json
    from ..util import get_initial_states, get_noise, to_device_async
    dev = next(agent.parameters()).device
    iouv = torch.linspace(0.5, 0.95, 10, device=dev)
    niou = iouv.numel()
    stats, records, seen = [], [], 0
    graphs = {}
    filter_names = [f.get_short_name() for f in agent.filters]
    def front(batch):
        """Uploads + the ISP episode + the detector forward of one batch: eagerly, or (graph=True) as an asynchronous replay."""
        im, targets, paths, shapes = batch
        # host arrays go up from pinned memory, asynchronously: a copy from pageable memory synchronises the stream, i.e. waits for
        # the previous batch's NMS / matching launches before this batch's first kernel can even be enqueued
        nb, _, height, width = im.shape
        targets = targets.clone()
        targets[:, 2:] *= torch.tensor((width, height, width, height), dtype=targets.dtype, device=targets.device)   # to pixels, where the batch lives (the host, for a loader's batches)
        im = to_device_async(im, dev).float()
        targets = to_device_async(targets, dev)
        noises = to_device_async(np.array([get_noise(nb, cfg.z_type, cfg.z_dim) for _ in range(steps)]), dev)
        states = to_device_async(get_initial_states(nb, cfg.num_state_dim, len(agent.filters)), dev)
        ctx = dict(im=im, targets=targets, paths=paths, shapes=shapes, noises=noises, states=states, nb=nb, slot=None)
        if graph and im.is_cuda and not param_dir and not agent.training:
            key = (tuple(im.shape), steps, None if pipeline is None else tuple(pipeline))
            eg = graphs.get(key)
            if eg is None:
                eg = graphs[key] = _EpisodeGraph(agent, detector, im, noises, states, steps, pipeline)
            ctx["slot"] = eg.launch(im, noises, states, keep_retouch=details is not None)
        return ctx

    def back(ctx):
        """Everything behind the forward: the step records, NMS, matching. Reads the replay's results (graph mode) or runs the
        eager loop."""
        nonlocal seen
        im, targets, paths, shapes, noises, states, nb = (ctx[k] for k in ("im", "targets", "paths", "shapes", "noises", "states", "nb"))
        retouch, ids = im, []
        params = collections.OrderedDict(pipeline=[])
        replayed, slot = False, ctx["slot"]
        if slot is not None:
            host = _EpisodeGraph.result(slot)
            sel, stop = host[:steps * nb], host[steps * nb:]
            if not any(v > 0 for v in stop[:-1]):                      # no early exit before the last step: the replay IS the loop
                ids = [[int(v) for v in sel[i * nb:(i + 1) * nb]] for i in range(steps)]
                retouch, preds, replayed = slot["retouch"], slot["preds"], True
        if not replayed and graph and im.is_cuda:
            # the eager loop below runs the agent and the detector on THIS stream while the next batch's replay may be in flight on
            # the side stream — on the same internal buffers: behind it (the next launch in turn waits for this stream)
            torch.cuda.current_stream(dev).wait_stream(_side_stream(dev))
        with torch.no_grad():
            for i in range(0 if not replayed else steps, steps):
                pipe = None if pipeline is None else pipeline[i]
                (retouch, states, _, _), dbg, _ = agent((retouch, noises[i], states), 1.0, None, pipe)
                # the step's two host reads — the chosen filter ids (records.txt) and image 0's "stopped" state (the early
                # exit below, val_adaptiveisp.py:302-303) — in ONE device-to-host copy: the loop is host-bound at batch 1
                host = torch.cat([dbg["selected_filter"].detach().to(torch.float32),
                                  states[0:1, STATE_STOPPED_DIM].detach().to(torch.float32)]).cpu().tolist()
                ids.append([int(v) for v in host[:-1]])
                if param_dir:
                    k = ids[-1][0]
                    params[filter_names[k]] = dbg["filter_debug_info"][k]["filter_parameters"].detach().cpu().numpy().tolist()
                    params["pipeline"].append(k)
                if host[-1] > 0:
                    break
            if not replayed:
                preds = detector(retouch)
        if param_dir:
            os.makedirs(param_dir, exist_ok=True)
            stem = os.path.splitext(os.path.split(str(paths[0]))[1])[0]
            with open(os.path.join(param_dir, stem + ".json"), "w") as f:
                json.dump(params, f, sort_keys=False, indent=4)
        for b in range(nb):
            row = ["-1"] * steps
            for i, step_ids in enumerate(ids):
                row[i] = str(step_ids[b])
            records.append((os.path.split(str(paths[b]))[1], row))
        preds = non_max_suppression(preds, conf_thres, iou_thres, multi_label=True, agnostic=single_cls,
                                    max_det=max_det, nms_fn=nms_fn)
        for si, pred in enumerate(preds):
            labels = targets[targets[:, 0] == si, 1:]
            nl, npr = labels.shape[0], pred.shape[0]
            shape = shapes[si][0]
            correct = torch.zeros(npr, niou, dtype=torch.bool, device=dev)
            seen += 1
            if details is not None:
                details.append(dict(path=str(paths[si]), retouch=retouch[si].detach().cpu(), pred=pred.detach().cpu().clone(),
                                    correct=None))
            if npr == 0:
                if nl:
                    stats.append((correct, *torch.zeros((2, 0), device=dev), labels[:, 0]))
                continue
            if single_cls:
                pred[:, 5] = 0
            predn = pred.clone()
            scale_boxes(im[si].shape[1:], predn[:, :4], shape, shapes[si][1])
            if nl:
                tbox = xywh2xyxy(labels[:, 1:5])
                scale_boxes(im[si].shape[1:], tbox, shape, shapes[si][1])
                labelsn = torch.cat((labels[:, 0:1], tbox), 1)
                correct = process_batch(predn, labelsn, iouv)
            if details is not None:
                details[-1]["correct"] = correct.detach().cpu()
            stats.append((correct, pred[:, 4], pred[:, 5], labels[:, 0]))
        if slot is not None:
            _EpisodeGraph.release(slot)

    # Graph mode is a two-stage pipeline over the batches: the replay of batch i + 1 (a side stream) runs while the host works
    # through NMS / matching of batch i (round 6: at batch 1 the loop was 1.8 ms of replay + 2.4 ms of host-bound NMS and matching
    # per image, one after the other). The eager loop keeps the reference's order: one batch at a time.
    pending = None
    for batch in batches:
        ctx = front(batch)
        if pending is not None:
            back(pending)
            pending = None
        if ctx["slot"] is not None:
            pending = ctx
        else:
            back(ctx)
    if pending is not None:
        back(pending)
    res = dict(mp=0.0, mr=0.0, map50=0.0, map75=0.0, map=0.0, seen=seen, ap_class=np.zeros(0, int), ap=np.zeros((0, niou)),
               records=records, filter_names=filter_names)
    stats = [torch.cat(x, 0).cpu().numpy() for x in zip(*stats)] if stats else []
    # (the copy above drained the device) a persistent conv chain whose dependency wait gave up has produced wrong detections
    # with every launch returning OK: every forward polls the host word of the one before it, this covers the last one
    if hasattr(detector, "check_chains"):
        detector.check_chains(sync=True)
    if len(stats) and stats[0].any():
        tp, fp, p, r, f1, ap, ap_class = ap_per_class(*stats)
        res.update(mp=float(p.mean()), mr=float(r.mean()), map50=float(ap[:, 0].mean()), map75=float(ap[:, 5].mean()),
                   map=float(ap.mean(1).mean()), ap=ap, ap_class=ap_class)
    res["nt"] = np.bincount(stats[3].astype(int), minlength=nc) if len(stats) else np.zeros(nc, int)
    if records_path:
        with open(records_path, "w") as f:                         # val_adaptiveisp.py:269-322 "records.txt"
            f.write(",".join(filter_names) + "\n")
            for name, row in records:
                f.write(name + "," + ",".join(row) + "\n")
    return res
