"""Replay memory of the RL loop with the image pool resident in HBM.

Restates `ReplayMemory` (replay_memory.py:39-231): a pool of `cfg.replay_memory_size` records
(image, label, path, shape, state); a training batch is drawn from the shuffled pool skipping finished records
(`get_next_fake_batch` :208-221); after the step the retouched images re-enter the pool unless their trajectory is
over-long (`replace_memory` :170-181) and fresh records top it up (`fill_pool` :120-133).

The reference keeps the images as host numpy arrays and moves the whole batch H2D before and D2H after every
iteration (train.py:255,380). Here the images live in one device tensor `[slots,3,H,W]`; a record is a slot number
plus small host-side metadata, a batch is one gather, re-insertion one scatter — pixels never cross PCIe. The ORDER
logic (Python `random.shuffle` / `random.random`, list append / slice) is kept literally, so with the same seed the
same records are drawn as in the reference (tests/golden/replay.npz).
"""
import random

import numpy as np
import torch

from .util import STATE_STEP_DIM, STATE_STOPPED_DIM, Dict, to_device_async


class DeviceReplayMemory:
    def __init__(self, cfg, source, batch_size, device, image_shape, rng=None, load=True):
        """`source.get_next_batch(n)` -> (images: list/array/tensor of n [3,H,W], labels: list of [k,6] arrays, paths,
        shapes) — the contract of the reference's `dataset.get_next_batch` (dataset.py:921-931). `rng`: a
        `random.Random` (default: the module-level generator the reference uses)."""
        self.cfg = cfg
        self.dataset = source
        self.batch_size = int(batch_size)
        self.device = torch.device(device)
        self.target_pool_size = int(cfg.replay_memory_size)
        self.rng = rng if rng is not None else random
        C, H, W = image_shape
        nslots = self.target_pool_size + 2 * self.batch_size
        self.images = torch.empty((nslots, C, H, W), dtype=torch.float32, device=self.device)
        self.free = list(range(nslots - 1, -1, -1))        # stack of unused slots
        self.image_pool = []                                # records: Dict(slot, label, path, shape, state[np])
        if load:
            self.fill_pool()

    # ------------------------------------------------------------------------------------------------------
    def get_initial_states(self, batch_size):
        return np.zeros((batch_size, self.cfg.num_state_dim), dtype=np.float32)

    def get_noise(self, batch_size):
        if self.cfg.z_type == "normal":
            return np.random.normal(0, 1, [batch_size, self.cfg.z_dim]).astype(np.float32)
        if self.cfg.z_type == "uniform":
            return np.random.uniform(0, 1, [batch_size, self.cfg.z_dim]).astype(np.float32)
        raise AssertionError("Unknown noise type: %s" % self.cfg.z_type)

    def _store(self, img):
        slot = self.free.pop()
        self.images[slot].copy_(torch.as_tensor(img), non_blocking=True)
        return slot

    def _release(self, rec):
        self.free.append(rec.slot)

    def fill_pool(self):
        while len(self.image_pool) < self.target_pool_size:
            im_list, label_list, path_list, shapes_list = self.dataset.get_next_batch(self.batch_size)
            for i in range(len(im_list)):
                if len(self.free) == 0:                    # more fresh records than the truncation below keeps
                    break
                self.image_pool.append(Dict(slot=self._store(im_list[i]), label=label_list[i], path=path_list[i],
                                            shape=shapes_list[i], state=self.get_initial_states(1)[0]))
        for rec in self.image_pool[self.target_pool_size:]:
            self._release(rec)
        self.image_pool = self.image_pool[:self.target_pool_size]
        assert len(self.image_pool) == self.target_pool_size

    def get_next_fake_batch(self, batch_size):
        self.rng.shuffle(self.image_pool)
        assert batch_size <= len(self.image_pool)
        batch = []
        while len(batch) < batch_size:
            if len(self.image_pool) == 0:
                self.fill_pool()
            record = self.image_pool[0]
            self.image_pool = self.image_pool[1:]
            if record.state[STATE_STOPPED_DIM] != 1:
                batch.append(record)
            else:
                self._release(record)                      # finished images leave the pool here
        return batch

    def get_feed_dict_and_states(self, batch_size, host_only=False):
        """-> dict(im [B,3,H,W] device tensor (a gather, no host copy), label, path, shape, state [B,S] device,
        z [B,z_dim] device, records). `host_only`: no device work — `slots` a list, `state` / `z` numpy arrays, no `im` (a
        caller that stages the batch itself: train.Trainer's graph mode gathers `images[slots]` inside its hipGraph)."""
        batch = self.get_next_fake_batch(batch_size)
        if host_only:
            return dict(label=[r.label for r in batch], path=[r.path for r in batch], shape=[r.shape for r in batch],
                        state=np.stack([r.state for r in batch], 0), z=self.get_noise(batch_size), records=batch,
                        slots=[r.slot for r in batch])
        # pinned staging + asynchronous copies: a pageable source would make each of these wait for the whole stream
        idx = to_device_async([r.slot for r in batch], self.device, dtype=torch.long)
        states = to_device_async(np.stack([r.state for r in batch], 0), self.device)
        z = to_device_async(self.get_noise(batch_size), self.device)
        return dict(im=self.images.index_select(0, idx), label=[r.label for r in batch], path=[r.path for r in batch],
                    shape=[r.shape for r in batch], state=states, z=z, records=batch, slots=idx)

    def replace_memory(self, batch, retouch, new_states, slots=None, device_copy=True):
        """Re-insert the retouched batch: `retouch` [B,3,H,W] device tensor is scattered into the records' own slots,
        `new_states` [B,S] becomes their state — pass it as a HOST array where the loop must not stall (a device tensor is
        read back with a blocking copy, i.e. after everything enqueued so far). `slots`: the records' slot indices on the
        device if the caller still has them (get_feed_dict_and_states()["slots"]). Over-long trajectories are kept with
        probability cfg.over_length_keep_prob; then the pool is topped up with fresh records. `device_copy=False`: the caller
        has written (or will write, before anything reads the pool) the retouched images into the records' slots itself."""
        states = new_states.detach().cpu().numpy() if isinstance(new_states, torch.Tensor) else np.asarray(new_states)
        if device_copy:
            idx = slots if slots is not None else to_device_async([r.slot for r in batch], self.device, dtype=torch.long)
            self.images.index_copy_(0, idx, retouch.detach().to(self.images.dtype))
        self.rng.shuffle(self.image_pool)
        for i, r in enumerate(batch):
            r = Dict(slot=r.slot, label=r.label, path=r.path, shape=r.shape, state=states[i].copy())
            if r.state[STATE_STEP_DIM] < self.cfg.maximum_trajectory_length or \
                    self.rng.random() < self.cfg.over_length_keep_prob:
                self.image_pool.append(r)
            else:
                self._release(r)
        self.fill_pool()
        self.rng.shuffle(self.image_pool)

    def drop_batch(self, batch):
        """The NaN / brightness guard of train.py:374-376: the batch is discarded and the pool refilled."""
        for r in batch:
            self._release(r)
        self.fill_pool()

    def debug(self):
        tot = sum(float(r.state[STATE_STEP_DIM]) for r in self.image_pool)
        return len(self.image_pool), tot / max(len(self.image_pool), 1)


class SyntheticSource:
    """Stand-in for the dataset object (no dataset in the container): seeded images U^2.2*0.5 with a few boxes. With a
    GPU device the pixels are drawn ON the device (its own seeded generator): a real loader prepares batches in worker
    processes beside the training loop (dataset.py:921-931 behind a DataLoader), which a stand-in that draws 6 MB of
    randoms per image on the training thread is not — 13 ms of host time per refill at 8 x 512 x 512."""

    def __init__(self, image_shape, nc=80, seed=0, device="cpu", max_boxes=3):
        self.shape, self.nc, self.count = tuple(image_shape), nc, 0
        self.g = torch.Generator(device="cpu").manual_seed(seed)
        self.device, self.max_boxes = device, max_boxes
        self.gd = None
        if torch.device(device).type == "cuda":
            self.gd = torch.Generator(device=device).manual_seed(seed)

    def get_next_batch(self, n):
        C, H, W = self.shape
        if self.gd is not None:
            ims = torch.rand(n, C, H, W, generator=self.gd, device=self.device) ** 2.2 * 0.5
        else:
            ims = (torch.rand(n, C, H, W, generator=self.g) ** 2.2 * 0.5).to(self.device)
        labels, paths, shapes = [], [], []
        for i in range(n):
            k = int(torch.randint(1, self.max_boxes + 1, (1,), generator=self.g))
            lb = np.zeros((k, 6), np.float32)
            lb[:, 1] = torch.randint(0, self.nc, (k,), generator=self.g).numpy()
            lb[:, 2:4] = (torch.rand(k, 2, generator=self.g) * 0.6 + 0.2).numpy()
            lb[:, 4:6] = (torch.rand(k, 2, generator=self.g) * 0.3 + 0.05).numpy()
            labels.append(lb)
            paths.append(f"synthetic_{self.count:07d}.png")
            shapes.append(((H, W), ((1.0, 1.0), (0.0, 0.0))))
            self.count += 1
        return list(ims), labels, paths, shapes
