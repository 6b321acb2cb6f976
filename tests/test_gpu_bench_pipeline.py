"""bench.py's two-stream software pipeline (ISP episode of batch i+1 beside the detector of batch i) must compute
exactly what the sequential step computes."""
import argparse
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pipelined_replay_equals_sequential_step():
    sys.path.insert(0, ROOT)
    import bench
    a = argparse.Namespace(batch=2, height=96, width=128, schedule="mixed", retune=False)
    bench.TUNE_CACHE = None                                   # tiny shapes: library-default kernels, nothing written
    step, engine, agent, x0, sched = bench.build_workload(a, torch.device("cuda:0"))
    ref = step().clone()                                      # eager: ISP episode then detector
    torch.cuda.synchronize()
    prime, run = bench.build_pipeline(step, engine, x0)
    prime()
    for _ in range(3):                                        # even and odd graphs, steady state
        run()
        torch.cuda.synchronize()
        assert torch.equal(engine.pred, ref)
    assert torch.isfinite(ref).all() and ref.shape[0] == 2
