#!/usr/bin/env python3
"""GPU time of the phases of one RL iteration (8 x 512 x 512, default configuration) from events recorded between them in an
UNPROFILED run: where the iteration's wall time goes beyond the summed kernel durations of tools/train_trace.sh.
usage: train_phase_events.py [iters=30]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adaptiveisp_amd import dist as adist, rl
from adaptiveisp_amd.config import cfg
from adaptiveisp_amd.train import build_trainer
from adaptiveisp_amd.yolo.loss import assign_labels_packed

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
tr = build_trainer(cfg, 0, 1, dev, 8, 512, tune_cache=os.path.join(ROOT, "adaptiveisp_amd", "yolo", "tuning", "mi355x.json"))
tr.train(3)
torch.cuda.synchronize()
names = ["feed+labels", "agent forward + filters", "detector forward + loss", "critic x2 + TD", "backward", "clip + Adam", "replay write"]
acc = np.zeros(len(names))
all_ev = []
for it in range(iters):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
    ev[0].record()
    feed = tr.replay.get_feed_dict_and_states(tr.batch_size)
    labels = [torch.as_tensor(lb) for lb in feed["label"]]
    imgs, z, states = feed["im"], feed["z"], feed["state"]
    with torch.no_grad():
        packed, packed_pair = assign_labels_packed(tr.loss_fn, tr.detector.head_shapes(), labels, dev, pair=True)
    ev[1].record()
    (retouch, new_states, surrogate, penalty), _, _ = tr.agent((imgs, z, states), 0.1)
    stats = rl.retouch_stats(retouch)
    cur, side = torch.cuda.current_stream(), rl._side_stream(dev)       # the trainer's early read-back of the states (train.py)
    sh = torch.empty(new_states.shape, dtype=new_states.dtype, pin_memory=True)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        sh.copy_(new_states.detach(), non_blocking=True)
        got = torch.cuda.Event()
        got.record(side)
    new_states.record_stream(side)
    ev[2].record()
    l_in, l_re = tr.detector.per_sample_loss_pair(tr.loss_fn, imgs, retouch, packed, packed_pair)
    ev[3].record()
    old_value, new_value = tr.value.forward_pair(imgs, states, retouch, new_states)
    out = rl.td_losses(cfg, l_in, l_re, penalty, surrogate, new_states, old_value, new_value, stats[:, 0:1], True, 0.9)
    ev[4].record()
    torch.autograd.backward([out["value_loss"], out["agent_loss"]])
    ev[5].record()
    adist.synced_step([tr.agent, tr.value], [tr.agent_optimizer, tr.value_optimizer], tr.buckets, max_grad_norm=1e-5)
    ev[6].record()
    got.synchronize()
    tr.replay.replace_memory(feed["records"], retouch.detach(), sh.numpy().copy(), slots=feed["slots"])
    ev[7].record()
    all_ev.append(ev)
    if os.environ.get("PHASE_SYNC") == "1":
        torch.cuda.synchronize()                              # each iteration from an idle GPU: the host's enqueue work exposed
torch.cuda.synchronize()
for ev in all_ev[10:]:                                        # (the host is ahead of the GPU by then: pure GPU-side durations)
    acc += np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(len(names))])
n = len(all_ev) - 10
for k, v in zip(names, acc / n):
    print(f"{v:7.3f} ms  {k}")
print(f"{acc.sum() / n:7.3f} ms  sum; wall per iteration {all_ev[10][0].elapsed_time(all_ev[-1][-1]) / n:.3f} ms")
