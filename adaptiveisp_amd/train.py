"""RL training loop of the hot path — the counterpart of `DynamicISP.train` (train.py:199-487) with the replay pool
in HBM and, under torchrun, batch-sharded data parallelism (one process per GPU, one flat gradient all-reduce per
model per iteration over RCCL, issued before the 1e-5 grad-norm clip: adaptiveisp_amd/dist.py).

What the reference's loop does per iteration and where it lives here:
  draw batch from replay memory (train.py:245-255)            DeviceReplayMemory.get_feed_dict_and_states (no H2D)
  agent / detector x2 / value x2 / TD losses (:258-305)         rl.train_iteration
  backward, clip 1e-5, Adam steps, LR schedule (:341-351)       rl.train_iteration + dist.synced_step, LambdaLR here
  NaN / brightness guard, replace_memory (:374-381)             here (device tensors; no D2H of the batch)
  checkpoint every save_model_freq (:471-486)                   yolo/checkpoint.save_isp_checkpoint
TensorBoard / console / visualisation glue is out of scope.
"""
import os

import torch

from . import dist as adist
from .rl import lr_lambda, train_iteration
from .yolo.checkpoint import save_isp_checkpoint


class Trainer:
    def __init__(self, cfg, agent, value, detector, loss_fn, replay, batch_size, lr=3e-5, epochs=800, save_dir=None,
                 use_truncated=True, max_bri=0.9, rank=0, world=1, sync_bn=False, graph=None):
        """`detector(x)` returns the three raw head maps with autograd to x (frozen reward model in train mode with
        BN in eval, train.py:236-243). `epochs` -> max_iter_step = epochs*1000//batch_size as train.py:156 (the global
        batch: per-rank batch x world). `sync_bn`: under data parallelism, compute the BatchNorm statistics of the
        agent / value CNNs (agent.py:40,51; value.py:22,34 run in train mode) over the GLOBAL batch with
        torch.nn.SyncBatchNorm — the single-GPU batch-64 semantics of the reference — instead of per rank.
        `graph` (default: ADAISP_TRAIN_GRAPH, "1"): after `graph_warmup` ordinary iterations the whole iteration — agent,
        detector pair, critic, TD losses, backward, clip + Adam of both models — is captured ONCE as a hipGraph and replayed
        (see _GraphIteration); the host's work per iteration is then the replay pool's bookkeeping, the label assignment and
        one upload. Under data parallelism (world > 1) the capture is split around the gradient all-reduce, which is issued as
        before. Needs a HIP device and the pair engine; anything else runs the ordinary loop."""
        if sync_bn and world > 1:
            agent = torch.nn.SyncBatchNorm.convert_sync_batchnorm(agent)
            value = torch.nn.SyncBatchNorm.convert_sync_batchnorm(value)
        self.cfg, self.agent, self.value, self.detector, self.loss_fn = cfg, agent, value, detector, loss_fn
        self.replay, self.batch_size, self.save_dir = replay, int(batch_size), save_dir
        self.use_truncated, self.max_bri, self.rank, self.world = use_truncated, max_bri, rank, world
        self.max_iter_step = max(1, int(epochs * 1000 // (self.batch_size * world)))
        # torch.optim.Adam as in train.py:208-209. On a GPU its single-kernel form (`fused=True`: the same update rule in one
        # launch per optimizer instead of ~10 tensor-list launches; state_dict layout unchanged): the iteration is bound by
        # the HOST's enqueue work once the detector runs split-K — 11.5-13.0 -> 10.9-11.2 ms interleaved on one
        # box. ADAISP_FUSED_ADAM=0: the default (foreach) form
        kw = dict(fused=True) if (os.environ.get("ADAISP_FUSED_ADAM", "1") == "1" and next(agent.parameters()).is_cuda) else {}
        self.agent_optimizer = torch.optim.Adam(agent.parameters(), lr=lr, **kw)
        self.value_optimizer = torch.optim.Adam(value.parameters(), lr=lr * float(cfg.value_lr_mul), **kw)   # train.py:208-209
        lf = lr_lambda(self.max_iter_step)
        self.agent_scheduler = torch.optim.lr_scheduler.LambdaLR(self.agent_optimizer, lr_lambda=lf)
        self.value_scheduler = torch.optim.lr_scheduler.LambdaLR(self.value_optimizer, lr_lambda=lf)
        self.buckets = [adist.GradBucket(agent, value)]          # ONE flattened bucket = one collective per iteration
        self.iter = 0
        want = os.environ.get("ADAISP_TRAIN_GRAPH", "1") == "1" if graph is None else bool(graph)
        self.graph_mode = bool(want and next(agent.parameters()).is_cuda and not (sync_bn and world > 1)   # (SyncBatchNorm: collectives inside the forward)
                               and getattr(detector, "per_sample_loss_pair", None) is not None)
        # world > 1: the gradient all-reduce stays OUTSIDE the capture (two graphs per iteration: forward + backward | clip + Adam);
        # graph="split" forces that form on one rank (tests)
        self.graph_split = bool(self.graph_mode and (world > 1 or graph == "split"))
        self.graph_warmup = 3            # ordinary iterations first: Adam's state and every lazily built table exist, every kernel has run
        self._git = None
        self._flag_host = self._states_host = None
        adist.broadcast_parameters([agent, value], src=0)
        self.history = []

    def step(self):
        if self.graph_mode and self.iter >= self.graph_warmup:
            return self._step_graph()
        return self._step_ordinary(self.replay.get_feed_dict_and_states(self.batch_size))

    def _step_ordinary(self, feed):
        it = self.iter
        if not self.agent.training:                          # (Module.train() walks ~130 modules: only when the mode changes)
            self.agent.train()
        if not self.value.training:
            self.value.train()
        progress = float(it) / self.max_iter_step
        labels = [torch.as_tensor(lb) for lb in feed["label"]]
        guard = {}

        def start_guard(retouch, stats, new_states):
            """The reference's check of the retouched batch (train.py:374-381: NaN / too dark / too bright -> the records are
            dropped instead of re-entering the pool) from the batch's per-image statistics (rl.retouch_stats: [B,2] mean /
            non-finite count). They exist as soon as the filters have run, so the flag is made THEN, on a second stream, and
            copied to pinned host memory behind an event: the host reads it after it has enqueued the rest of the iteration
            without waiting for that rest (a `bool(tensor)` on the main stream at the end of the iteration drains the GPU
            every iteration)."""
            def flag():
                mean = stats[:, 0].mean()
                return ((stats[:, 1].sum() > 0) | ~torch.isfinite(mean) | (mean < 0.01) | (mean > self.max_bri)).reshape(1)
            if not retouch.is_cuda:
                guard["bad"], guard["states"] = bool(flag()), new_states.cpu().numpy()
                return
            from .rl import _side_stream
            cur, side = torch.cuda.current_stream(), _side_stream(retouch.device)
            if self._flag_host is None or self._states_host.shape != new_states.shape:
                self._flag_host = torch.empty((1,), dtype=torch.bool, pin_memory=True)
                self._states_host = torch.empty(new_states.shape, dtype=new_states.dtype, pin_memory=True)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                self._flag_host.copy_(flag(), non_blocking=True)
                # the replay pool keeps the records' states on the host (replay.replace_memory): read back HERE, behind the same
                # event — a .cpu() at the end of the iteration waits for its whole backward and leaves the GPU idle while the
                # host enqueues the next iteration
                self._states_host.copy_(new_states, non_blocking=True)
                guard["event"] = torch.cuda.Event()
                guard["event"].record(side)
            stats.record_stream(side)
            new_states.record_stream(side)

        out = train_iteration(self.cfg, self.agent, self.value, self.detector, self.loss_fn, feed["im"], feed["z"],
                              feed["state"], labels, progress, [self.agent_optimizer, self.value_optimizer],
                              buckets=self.buckets, use_truncated=self.use_truncated, max_bri=self.max_bri,
                              on_retouch=start_guard)
        self.agent_scheduler.step()
        self.value_scheduler.step()
        retouch = out["retouch"]
        if "event" in guard:
            guard["event"].synchronize()                     # the flag and the states only: the iteration's backward may still be running
            bad, states_host = bool(self._flag_host[0]), self._states_host.numpy().copy()
        else:
            bad, states_host = guard["bad"], guard["states"]
        if bad:
            self.replay.drop_batch(feed["records"])
        else:
            self.replay.replace_memory(feed["records"], retouch, states_host, slots=feed.get("slots"))
        # losses stay device tensors here: converting them would wait for the whole iteration; `materialize()` (called by
        # train() at its end, by save(), and by anyone who wants numbers) turns them into floats
        rec = dict(iter=it, agent_loss=out["agent_loss"].detach(), value_loss=out["value_loss"].detach(),
                   reward=out["reward"].detach().mean(), dropped=bad)
        self.history.append(rec)
        self.iter += 1
        if it % 256 == 255:
            self.materialize()                               # (bounds the number of live 0-dim device tensors)
        if self.save_dir and self.rank == 0 and it % self.cfg.save_model_freq == 0 and it > 0:
            self.save(it)
        return rec

    # ---- the iteration as ONE hipGraph ----------------------------------------------------------------------------------
    def _device_feed(self, hf):
        """A host-only draw (replay.get_feed_dict_and_states(host_only=True)) as the ordinary loop's feed."""
        from .util import to_device_async
        dev = self.replay.device
        idx = to_device_async(hf["slots"], dev, dtype=torch.long)
        return dict(hf, im=self.replay.images.index_select(0, idx), state=to_device_async(hf["state"], dev),
                    z=to_device_async(hf["z"], dev), slots=idx)

    def _step_graph(self):
        """step() with the device side of the iteration replayed from a hipGraph (_GraphIteration). What stays on the host:
        drawing the batch (replay.py), the labels' target assignment (yolo.loss.assign_labels_host), filling ONE pinned block
        (tables, pool rows, states, noise, the iteration's three scalars), the replay, the guard's verdict and the pool's
        bookkeeping. Same arithmetic, same order of launches as the ordinary step."""
        it = self.iter
        progress = float(it) / self.max_iter_step
        feed = self.replay.get_feed_dict_and_states(self.batch_size, host_only=True)
        G = self._git
        if G is None:
            G = self._git = _GraphIteration(self, self.replay.images, split=self.graph_split)
        coef = (1.0 - progress) * self.cfg.exploration_penalty
        lrs = (self.agent_optimizer.param_groups[0]["lr"], self.value_optimizer.param_groups[0]["lr"])
        while not G.stage(feed["label"], feed["slots"], feed["state"], feed["z"], coef, *lrs):
            # more matches than the tables hold: larger tables, new capture
            G = self._git = _GraphIteration(self, self.replay.images, cap=2 * G.tables.cap, split=self.graph_split)
        if G.graph is None:
            try:
                G.capture()                                  # (drains the device first: whatever ran before is complete)
            except Exception as e:                           # noqa: BLE001 — nothing has run yet: this iteration and the rest take the ordinary loop
                import warnings
                warnings.warn(f"adaptiveisp_amd.train: the iteration could not be captured as a hipGraph ({type(e).__name__}: "
                              f"{str(e)[:300]}); continuing with the ordinary loop", RuntimeWarning)
                self.graph_mode, self._git = False, None
                for opt in (self.agent_optimizer, self.value_optimizer):
                    opt.zero_grad(set_to_none=True)
                torch.cuda.synchronize(G.dev)
                return self._step_ordinary(self._device_feed(feed))
        G.replay()
        for opt in (self.agent_optimizer, self.value_optimizer):     # what Optimizer.step's wrapper records (LambdaLR looks at it)
            opt._opt_called = True
            if hasattr(opt, "_step_count"):
                opt._step_count += 1
        self.agent_scheduler.step()
        self.value_scheduler.step()
        bad, states_host = G.wait_guard()                    # ~1 ms into the iteration: its backward is still running
        if bad:
            self.replay.drop_batch(feed["records"])
        else:
            self.replay.replace_memory(feed["records"], None, states_host, device_copy=False)    # (the graph scattered the images)
        slot = G.kept_scalars()
        rec = dict(iter=it, agent_loss=slot[0], value_loss=slot[1], reward=slot[2], dropped=bad)
        self.history.append(rec)
        self.iter += 1
        if it % 256 == 255:
            self.materialize()                               # (before the ring of G.scalars comes round)
        if self.save_dir and self.rank == 0 and it % self.cfg.save_model_freq == 0 and it > 0:
            self.save(it)
        return rec

    def materialize(self):
        """History entries as plain floats (one synchronisation for all that are still device tensors)."""
        for rec in self.history:
            for k in ("agent_loss", "value_loss", "reward"):
                if isinstance(rec[k], torch.Tensor):
                    rec[k] = float(rec[k])
        return self.history

    def train(self, iters=None):
        n = self.max_iter_step + 1 if iters is None else iters
        for _ in range(n):
            self.step()
        return self.materialize()

    def save(self, it):
        os.makedirs(self.save_dir, exist_ok=True)
        path = os.path.join(self.save_dir, "ckpt-%d.pth" % it)           # train.py:473 naming
        save_isp_checkpoint(path, it, self.agent, self.value, self.agent_optimizer, self.value_optimizer)
        return path


class _GraphIteration:
    """One RL iteration (rl.train_iteration over the pair engine) captured as a hipGraph. Everything that changes between
    iterations enters through ONE pinned host block, copied by the graph's first node into the buffers its kernels read:
      tables                         the labels' target assignment, fixed row count, padded with rows of no image
                                     (yolo.loss.StaticLabelTables)
      coef (fp32)                    (1 - progress) * exploration_penalty   -> adaisp_policy_tail_args.entropy_coef_dev
      lr[0], lr[1] (fp64)            the two optimizers' learning rates      -> adaisp_clip_adam_step_dev
      slots (int64 [B])              the batch's rows of the replay pool: the graph's second node gathers pool[slots] -> im
      state [B,S], z [B,Z]           the records' state vectors, the noise
    and leaves through pinned host memory the device writes ~1 ms into the iteration, on the side stream, in this order: the
    guard's flag (train.py:374-381), the new state vectors, a sequence number. The host polls the sequence number
    (wait_guard) — an event recorded inside a capture cannot be waited on from the host — reads flag and states, and does the
    pool's bookkeeping and the next batch's label assignment while the iteration's backward runs. The retouched batch re-enters
    the pool INSIDE the graph, behind the guard on the side stream (pool[slots] = flag ? pool[slots] : retouch — replace_memory's
    scatter, train.py:380), and the iteration's three scalars go into a ring indexed by the sequence number: between two replays
    the stream holds nothing but the pool's refills — the gather, the three input copies, the uploads and the scatter of the
    first version sat there with 30-250 us of engine hand-over each, 0.6 ms per iteration in which no kernel ran
    (tools/train_timeline.sh). The detector engines launch their kernels one by one inside the capture (train_engine._graph: no
    nested graph replay); the critic's two calls and the guard fork onto the side stream inside the capture and join before its end.
    Nothing is executed at capture time: the captured iteration runs for the first time at its first replay, so a trainer
    in graph mode makes the same sequence of updates as one in the ordinary loop (tests/test_gpu_train_graph.py)."""

    def __init__(self, tr, pool, cap=512, split=False):
        import numpy as np

        from .yolo.loss import StaticLabelTables
        self.tr, self.split, self.pool = tr, bool(split), pool
        # ADAISP_TRAIN_GRAPH_STREAMS=2 (default): the critic, the guard and the input half's shallow detector layers fork onto a
        # second stream inside the capture, as in the ordinary loop. 1: one chain of launches — on this runtime a fork / join
        # inside a graph costs ~0.3 ms of idle device per replay and ten times the launch work (tools/graph_launch_gap.py), but
        # the small launches of critic and guard then no longer run beside the detector's: 6.7-6.9 ms per iteration against
        # 6.5-6.7, host work 0.7 ms against 1.4
        self.one_stream = os.environ.get("ADAISP_TRAIN_GRAPH_STREAMS", "2") == "1"
        dev = pool.device
        self.dev = dev
        B, S, Z = tr.batch_size, int(tr.cfg.num_state_dim), int(tr.cfg.z_dim)
        extra = 6 + 2 * B + B * S + B * Z
        self.tables = StaticLabelTables(tr.loss_fn, tr.detector.head_shapes(), B, dev, cap=cap, extra_words=extra + (extra & 1))
        ed, eh = self.tables.extra_dev, self.tables.extra_host.numpy()
        o_sl, o_st, o_z = 6, 6 + 2 * B, 6 + 2 * B + B * S
        self.coef = ed[0:1].view(torch.float32)
        self.lr = [ed[2:4].view(torch.float64), ed[4:6].view(torch.float64)]
        self.slots = ed[o_sl:o_st].view(torch.int64)
        self.state = ed[o_st:o_z].view(torch.float32).view(B, S)
        self.z = ed[o_z:o_z + B * Z].view(torch.float32).view(B, Z)
        self._coef_h, self._lr_h = eh[0:1].view(np.float32), [eh[2:4].view(np.float64), eh[4:6].view(np.float64)]
        self._slots_h = eh[o_sl:o_st].view(np.int64)
        self._state_h = eh[o_st:o_z].view(np.float32).reshape(B, S)
        self._z_h = eh[o_z:o_z + B * Z].view(np.float32).reshape(B, Z)
        self.im = torch.empty((B,) + tuple(pool.shape[1:]), dtype=pool.dtype, device=dev)
        self.flag_host = torch.zeros((1,), dtype=torch.bool, pin_memory=True)
        self.states_host = torch.zeros((B, S), dtype=torch.float32, pin_memory=True)
        self.seq_host = torch.zeros((1,), dtype=torch.int32, pin_memory=True)
        self.seq_dev = torch.zeros((1,), dtype=torch.int32, device=dev)
        self._seq_np, self._replays = self.seq_host.numpy(), 0
        self.scalars = torch.zeros((256, 3), dtype=torch.float32, device=dev)
        self.graph = self.graph_step = self.out = None

    def stage(self, labels, slots, states, z, coef, lr_agent, lr_value):
        """The next iteration's inputs into the pinned block (the graph's first copy node reads it). False — nothing usable
        written — when the labels need more rows than the tables hold. Call only after wait_guard() of this graph's previous
        replay (that copy has then long run)."""
        if not self.tables.fill(labels):
            return False
        self._coef_h[0] = coef                               # (rounded to fp32 as the by-value kernel argument is)
        self._lr_h[0][0], self._lr_h[1][0] = lr_agent, lr_value
        self._slots_h[:] = slots
        self._state_h[:] = states
        self._z_h[:] = z
        return True

    def _guard(self, retouch, stats, new_states):
        from .rl import _side_stream
        tr = self.tr
        cur = torch.cuda.current_stream()
        side = cur if self.one_stream else _side_stream(self.dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            mean = stats[:, 0].mean()
            flag = ((stats[:, 1].sum() > 0) | ~torch.isfinite(mean) | (mean < 0.01) | (mean > tr.max_bri)).reshape(1)
            # replace_memory's scatter (a dropped batch leaves the pool as it is) — BEFORE the host is told: once it has read the
            # sequence number it refills released rows of the pool, and nothing of this iteration may touch the pool after that
            keep = self.pool.index_select(0, self.slots)
            self.pool.index_copy_(0, self.slots, torch.where(flag.view(1, 1, 1, 1), keep, retouch.to(self.pool.dtype)))
            self.flag_host.copy_(flag, non_blocking=True)
            self.states_host.copy_(new_states, non_blocking=True)
            self.seq_dev.add_(1)
            self.seq_host.copy_(self.seq_dev, non_blocking=True)
        stats.record_stream(side)
        new_states.record_stream(side)
        retouch.record_stream(side)

    def capture(self):
        from . import dist as adist
        from .rl import _side_stream
        tr = self.tr
        for m in (tr.agent, tr.value):
            if not m.training:
                m.train()
        tr.agent.entropy_coef_dev = self.coef
        getattr(tr.detector, "prepare_capture", lambda: None)()                      # (launch plans: building them probes kernels)
        torch.cuda.synchronize(self.dev)
        from . import optim as aoptim
        g, g2 = torch.cuda.CUDAGraph(), None
        opts = [tr.agent_optimizer, tr.value_optimizer]
        for o in opts:
            aoptim.reserve_capture_table(o)                                          # (pinned: not inside the capture)
        try:
            with torch.cuda.graph(g):
                self.tables.upload()                                                 # the pinned block -> what the kernels read
                torch.index_select(self.pool, 0, self.slots, out=self.im)            # the batch: a gather from the pool
                out = train_iteration(tr.cfg, tr.agent, tr.value, tr.detector, tr.loss_fn, self.im, self.z, self.state, None, 0.0,
                                      opts, buckets=tr.buckets, use_truncated=tr.use_truncated, max_bri=tr.max_bri,
                                      on_retouch=self._guard, assigned=(self.tables.packed, self.tables.packed_pair),
                                      lr_dev=self.lr, step=not self.split, one_stream=self.one_stream)
                vec3 = torch.stack([out["agent_loss"].detach().reshape(()), out["value_loss"].detach().reshape(()),
                                    out["reward"].detach().mean()])
                if not self.one_stream:
                    torch.cuda.current_stream().wait_stream(_side_stream(self.dev))  # (joined whatever the stream switches say)
                self.scalars.index_copy_(0, ((self.seq_dev - 1) % 256).to(torch.int64), vec3.view(1, 3))
            if self.split:
                # the gradients now sit at fixed addresses (every replay of the first graph writes them there); the collective
                # runs on them between the two replays; the second graph clips and steps from the same addresses
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2, pool=g.pool()):
                    adist.synced_step([tr.agent, tr.value], opts, tr.buckets or [], max_grad_norm=1e-5, lr_dev=self.lr,
                                      collective=False, release=False)
        finally:
            tr.agent.entropy_coef_dev = None
        self.graph, self.graph_step, self.out = g, g2, out

    def replay(self):
        self.graph.replay()
        if self.split:
            from . import dist as adist
            adist.reduce_gradients(self.tr.buckets or [])
            self.graph_step.replay()

    def wait_guard(self, timeout=120.0):
        """The guard's verdict and the new states of the replay launched last, as soon as the device has written them."""
        import time
        self._replays += 1
        want, t0, spins = self._replays, time.perf_counter(), 0
        while int(self._seq_np[0]) != want:
            spins += 1
            if spins & 1023 == 0 and time.perf_counter() - t0 > timeout:
                raise RuntimeError(f"graph iteration: the guard's sequence number did not reach {want} within {timeout} s "
                                   f"(it reads {int(self._seq_np[0])})")
        return bool(self.flag_host[0]), self.states_host.numpy().copy()

    def kept_scalars(self):
        """(agent loss, value loss, mean reward) of the replay launched last: a row of the device-side ring."""
        return self.scalars[(self._replays - 1) % 256]


# ---------------------------------------------------------------------------------------------------------------
# Launchable entry (BASELINE config 4): `python -m adaptiveisp_amd.train ...` on one GPU, or one process per GPU under
#   python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m adaptiveisp_amd.train
# The counterpart of train.py:620-660 (`DynamicISP(...).train()`), minus the dataset plumbing: the data source is
# any object with `get_next_batch(n)` (dataset.py:921-931); without a dataset in the container the synthetic source
# stands in.
# ---------------------------------------------------------------------------------------------------------------
def build_trainer(cfg, rank, world, device, batch_size, image_size, lr=3e-5, epochs=800, save_dir=None, sync_bn=False,
                  detector_ckpt=None, isp_ckpt=None, source=None, seed=0, tune_cache=None):
    """Everything one rank owns: its own replay pool in HBM (seeded by rank, so ranks draw different records), the
    frozen detector on the HIP training engine, agent / value / optimizers. Rank 0's weights are broadcast."""
    import random

    from .agent import Agent
    from .replay import DeviceReplayMemory, SyntheticSource
    from .value import Value
    from .yolo import YoloTrainEngine, YoloTrainPairEngine, yolov3
    from .yolo.checkpoint import load_detector_checkpoint, load_isp_checkpoint
    from .yolo.loss import DetectionLoss, default_hyp
    H = W = int(image_size)
    torch.manual_seed(seed)
    agent = Agent(cfg, shape=(6 + len(cfg.filters), 64, 64), device=device).to(device)
    value = Value(cfg, shape=(9 + len(cfg.filters), 64, 64)).to(device)
    if isp_ckpt:
        load_isp_checkpoint(isp_ckpt, agent, value, map_location=device)
    det = load_detector_checkpoint(detector_ckpt) if detector_ckpt else yolov3()
    det = det.to(device).train()
    for m in det.modules():                                   # frozen reward model (train.py:236-243)
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()
    for p in det.parameters():
        p.requires_grad_(False)
    nc = det.model[-1].nc
    loss_fn = DetectionLoss(det.model[-1].anchors, nc=nc, hyp=default_hyp(nc, H), device=device)
    # one 2B-image forward for the input + retouched batches (yolo.YoloTrainPairEngine); ADAYOLO_TRAIN_PAIR=0: two B-image passes
    if os.environ.get("ADAYOLO_TRAIN_PAIR", "1") == "1":
        detector = YoloTrainPairEngine(det, batch_size, H, W, device=device)
    else:
        detector = YoloTrainEngine(det, batch_size, H, W, device=device)
    if tune_cache:
        detector.autotune(cache=tune_cache, write=(rank == 0))
    if source is None:
        source = SyntheticSource((3, H, W), nc=nc, seed=1000 * seed + rank, device=device)
    replay = DeviceReplayMemory(cfg, source, batch_size, device, (3, H, W), rng=random.Random(1000 * seed + rank))
    return Trainer(cfg, agent, value, detector, loss_fn, replay, batch_size=batch_size, lr=lr, epochs=epochs,
                   save_dir=save_dir, rank=rank, world=world, sync_bn=sync_bn)


def main(argv=None):
    import argparse
    import json
    import time

    from .config import cfg
    ap = argparse.ArgumentParser(description="RL training of the ISP policy, data-parallel over the GPUs torchrun gives it")
    ap.add_argument("--iters", type=int, default=None, help="iterations to run (default: the full schedule)")
    ap.add_argument("--warmup", type=int, default=2, help="untimed iterations before the throughput clock starts")
    ap.add_argument("--batch", type=int, default=8, help="per-rank batch (config 4: 8 x 8 GPUs = 64)")
    ap.add_argument("--size", type=int, default=512, help="training image size (train.py --imgsz 512)")
    ap.add_argument("--lr", type=float, default=3e-5)
    ap.add_argument("--epochs", type=int, default=800)
    ap.add_argument("--save-dir", default=None)
    ap.add_argument("--sync-bn", action="store_true")
    ap.add_argument("--detector-ckpt", default=None, help="yolov3.pt (pickled reference module)")
    ap.add_argument("--isp-ckpt", default=None, help="ckpt-*.pth to resume from")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--gpus", type=int, default=None, help="start this many ranks (one per GPU) under torch.distributed.run "
                    "as a child process; without it the process is one rank of whatever torchrun set up")
    a = ap.parse_args(argv)
    if a.gpus and a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import sys
        # the parent never touches the GPU: it starts the ranks as a child and exits with their code
        raise SystemExit(adist.launch_ranks(a.gpus, "adaptiveisp_amd.train", sys.argv[1:] if argv is None else argv, module=True))
    # ADAISP_DP_REHEARSAL=dry: the launch / barrier / max-over-ranks / one-JSON-line harness with no device (an iteration is
    # a host sleep + one real bucketed all-reduce over gloo) — what the CPU test of `--gpus N` runs
    dry = os.environ.get("ADAISP_DP_REHEARSAL") == "dry"
    rank, world, device = adist.init_from_env("gloo" if dry else None)
    if a.gpus and world != a.gpus:
        raise SystemExit(f"adaptiveisp_amd.train: --gpus {a.gpus} but WORLD_SIZE={world}")
    if dry:
        class _Dry:
            max_iter_step, history = 10, []

            def __init__(self):
                self.net = torch.nn.Linear(8, 8)
                self.bucket = adist.GradBucket(self.net)

            def train(self, iters):
                for _ in range(iters):
                    time.sleep(0.002)
                    for p in self.net.parameters():
                        p.grad = torch.ones_like(p)
                    self.bucket.finish(self.bucket.all_reduce_mean())
        tr = _Dry()
    else:
        if device.type != "cuda":
            raise SystemExit("adaptiveisp_amd.train needs a HIP device (the ISP and detector kernels have no CPU path)")
        if os.environ.get("ADAISP_DP_REHEARSAL") != "1" and torch.cuda.device_count() < world:
            raise SystemExit(f"adaptiveisp_amd.train: {world} ranks but only {torch.cuda.device_count()} device(s) visible")
        cache = os.path.join(os.path.dirname(os.path.abspath(__file__)), "yolo", "tuning", "mi355x.json")
        tr = build_trainer(cfg, rank, world, device, a.batch, a.size, lr=a.lr, epochs=a.epochs, save_dir=a.save_dir,
                           sync_bn=a.sync_bn, detector_ckpt=a.detector_ckpt, isp_ckpt=a.isp_ckpt, seed=a.seed, tune_cache=cache)
    n = tr.max_iter_step + 1 if a.iters is None else a.iters
    tr.train(min(a.warmup, n))

    def fence():
        if world > 1:
            torch.distributed.barrier()
        if not dry:
            torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    tr.train(max(0, n - a.warmup))
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    timed = max(0, n - a.warmup)
    # the collective alone: the iteration's ONE flattened gradient bucket all-reduced 5 times back to back, bracketed by a
    # pair of events on the stream it is issued from (host clock without a device) — what the xGMI ring costs per iteration
    bucket = (tr.buckets[0] if hasattr(tr, "buckets") else tr.bucket)
    bucket_bytes = 4 * (bucket.numel + len(bucket.params) + 1)       # gradients + presence mask + one status float
    ar_ms = None
    if world > 1 and bucket.flat is not None:
        reps = 5
        torch.distributed.all_reduce(bucket.flat)                    # warm the communicator
        fence()
        if bucket.flat.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                torch.distributed.all_reduce(bucket.flat)
            e1.record()
            torch.cuda.synchronize()
            ar_ms = e0.elapsed_time(e1) / reps
        else:
            t1 = time.perf_counter()
            for _ in range(reps):
                torch.distributed.all_reduce(bucket.flat)
            ar_ms = (time.perf_counter() - t1) / reps * 1e3
    if rank == 0:
        last = tr.history[-1] if tr.history else {}
        ms_it = dt / max(timed, 1) * 1e3
        # the same one-line shape as bench.py (metric / value / unit / n_gpus / steps / warmup / ms_per_step / scaling / config),
        # so that the first multi-GPU lease yields a config-4 scaling point from one command
        print(json.dumps({"metric": "RL training images/sec", "value": round(world * a.batch * timed / dt, 2) if timed else None,
                          "unit": "images/sec", "n_gpus": world, "steps": timed, "warmup": min(a.warmup, n),
                          "ms_per_step": round(ms_it, 2), "iterations_per_sec": round(1e3 / ms_it, 2) if timed else None,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "bf16 detector / fp32 ISP + heads", "data": "synthetic" + (" (DRY REHEARSAL: no device)" if dry else
                                                 " (REHEARSAL: all ranks share one device over gloo; not a measurement)"
                                                 if os.environ.get("ADAISP_DP_REHEARSAL") == "1" and world > 1 else ""),
                          "iters": timed, "ms_per_iter": round(ms_it, 2),
                          "per_gpu_batch": a.batch, "global_batch": a.batch * world, "image": f"{a.size}x{a.size}",
                          "config": {"workload": f"RL iteration (agent + value + replay + frozen YOLOv3 reward, train.py:234-351) "
                                                 f"batch {a.batch} x {a.size}x{a.size} per GPU", "per_gpu_batch": a.batch,
                                     "global_batch": a.batch * world, "parallelism": f"dp{world}"},
                          # what the N > 1 semantics are: BatchNorm statistics of the agent / value CNNs per rank (False) or
                          # over the global batch (True = the reference's single-GPU batch-64 behaviour); gradients of both
                          # models travel in ONE flattened all-reduce per iteration
                          "graph": (("split" if tr.graph_split else "one") if getattr(tr, "_git", None) is not None
                                    and tr._git.graph is not None else None),
                          "sync_bn": bool(a.sync_bn), "grad_buckets": 1, "grad_bucket_bytes": bucket_bytes,
                          "all_reduce_ms": round(ar_ms, 3) if ar_ms is not None else None,
                          "mask_fetches": getattr(bucket, "mask_fetches", 0),
                          "parallelism": f"dp{world}" + (" (DRY REHEARSAL: no device)" if dry else ""),
                          "last": last}), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
