#!/usr/bin/env python3
"""Soak of the shipped training loop (config 4 per rank, 8 x 512 x 512, train.build_trainer): N iterations in one process; reports
ms per iteration per block of 1000, every iteration with a non-finite loss, dropped batches, whether every parameter is finite at the
end and the device memory high-water mark at the end of each block (growth = a leak).
usage (GPU box): [ADAISP_TRAIN_GRAPH=0|1] [ADAISP_TRAIN_GRAPH_STREAMS=1|2] python tools/train_soak.py [iters=10000]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptiveisp_amd import train as atrain  # noqa: E402
from adaptiveisp_amd.config import cfg  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
dev = torch.device("cuda:0")
cache = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "adaptiveisp_amd", "yolo", "tuning", "mi355x.json")
tr = atrain.build_trainer(cfg, 0, 1, dev, 8, 512, tune_cache=cache, seed=0)
print(f"graph mode {tr.graph_mode}, streams inside the capture {os.environ.get('ADAISP_TRAIN_GRAPH_STREAMS', '2')}", flush=True)
done, bad_total = 0, []
while done < N:
    n = min(1000, N - done)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    h = tr.train(n)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    blk = h[done:done + n]
    bad = [r["iter"] for r in blk if not np.isfinite([r["agent_loss"], r["value_loss"], r["reward"]]).all()]
    bad_total += bad
    print(f"iterations {done:6d}..{done + n - 1:6d}: {dt:.3f} ms each, non-finite {len(bad)}{(' first at ' + str(bad[0])) if bad else ''}, "
          f"dropped {sum(r['dropped'] for r in blk)}, max memory {torch.cuda.max_memory_allocated() / 2 ** 30:.2f} GiB", flush=True)
    done += n
ok = all(torch.isfinite(p).all().item() for m in (tr.agent, tr.value) for p in m.parameters())
print(f"{N} iterations: {len(bad_total)} with a non-finite loss, every parameter finite: {ok}")
