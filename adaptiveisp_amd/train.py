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
                 use_truncated=True, max_bri=0.9, rank=0, world=1, sync_bn=False):
        """`detector(x)` returns the three raw head maps with autograd to x (frozen reward model in train mode with
        BN in eval, train.py:236-243). `epochs` -> max_iter_step = epochs*1000//batch_size as train.py:156 (the global
        batch: per-rank batch x world). `sync_bn`: under data parallelism, compute the BatchNorm statistics of the
        agent / value CNNs (agent.py:40,51; value.py:22,34 run in train mode) over the GLOBAL batch with
        torch.nn.SyncBatchNorm — the single-GPU batch-64 semantics of the reference — instead of per rank."""
        if sync_bn and world > 1:
            agent = torch.nn.SyncBatchNorm.convert_sync_batchnorm(agent)
            value = torch.nn.SyncBatchNorm.convert_sync_batchnorm(value)
        self.cfg, self.agent, self.value, self.detector, self.loss_fn = cfg, agent, value, detector, loss_fn
        self.replay, self.batch_size, self.save_dir = replay, int(batch_size), save_dir
        self.use_truncated, self.max_bri, self.rank, self.world = use_truncated, max_bri, rank, world
        self.max_iter_step = max(1, int(epochs * 1000 // (self.batch_size * world)))
        self.agent_optimizer = torch.optim.Adam(agent.parameters(), lr=lr)
        self.value_optimizer = torch.optim.Adam(value.parameters(), lr=lr)
        lf = lr_lambda(self.max_iter_step)
        self.agent_scheduler = torch.optim.lr_scheduler.LambdaLR(self.agent_optimizer, lr_lambda=lf)
        self.value_scheduler = torch.optim.lr_scheduler.LambdaLR(self.value_optimizer, lr_lambda=lf)
        self.buckets = [adist.GradBucket(agent), adist.GradBucket(value)]
        self.iter = 0
        adist.broadcast_parameters([agent, value], src=0)
        self.history = []

    def step(self):
        it = self.iter
        self.agent.train()
        self.value.train()
        progress = float(it) / self.max_iter_step
        feed = self.replay.get_feed_dict_and_states(self.batch_size)
        labels = [torch.as_tensor(lb) for lb in feed["label"]]
        out = train_iteration(self.cfg, self.agent, self.value, self.detector, self.loss_fn, feed["im"], feed["z"],
                              feed["state"], labels, progress, [self.agent_optimizer, self.value_optimizer],
                              buckets=self.buckets, use_truncated=self.use_truncated, max_bri=self.max_bri)
        self.agent_scheduler.step()
        self.value_scheduler.step()
        retouch = out["retouch"]
        mean = torch.mean(retouch)
        bad = bool((~torch.isfinite(retouch)).any() | (mean < 0.01) | (mean > self.max_bri))   # one host sync, as the reference
        if bad:
            self.replay.drop_batch(feed["records"])
        else:
            self.replay.replace_memory(feed["records"], retouch, out["new_states"])
        rec = dict(iter=it, agent_loss=float(out["agent_loss"].detach()), value_loss=float(out["value_loss"].detach()),
                   reward=float(out["reward"].detach().mean()), dropped=bad)
        self.history.append(rec)
        self.iter += 1
        if self.save_dir and self.rank == 0 and it % self.cfg.save_model_freq == 0 and it > 0:
            self.save(it)
        return rec

    def train(self, iters=None):
        n = self.max_iter_step + 1 if iters is None else iters
        for _ in range(n):
            self.step()
        return self.history

    def save(self, it):
        os.makedirs(self.save_dir, exist_ok=True)
        path = os.path.join(self.save_dir, "ckpt-%d.pth" % it)           # train.py:473 naming
        save_isp_checkpoint(path, it, self.agent, self.value, self.agent_optimizer, self.value_optimizer)
        return path
