"""bench.py's two-stream software pipeline (ISP episode of batch i+1 beside the detector of batch i) must compute
exactly what the sequential step computes."""
import argparse
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("cut,gate", [(None, None), (0, 0), (0, 3), (2, 0), (5, 0), (5, 1), (4, 3), (9, 1), (1, 9)])
def test_pipelined_replay_equals_sequential_step(cut, gate):
    """cut: the half-step at which the episode is split between two replays; gate: half-steps that run before the
    detector stream is released (None, None = bench.py's default: cut in front of the NLM launch, detector after it)."""
    sys.path.insert(0, ROOT)
    import bench
    a = argparse.Namespace(batch=2, height=96, width=128, schedule="mixed", retune=False)
    bench.TUNE_CACHE = None                                   # tiny shapes: library-default kernels, nothing written
    step, engine, agent, x0, sched = bench.build_workload(a, torch.device("cuda:0"))
    ref = step().clone()                                      # eager: ISP episode then detector
    torch.cuda.synchronize()
    prime, run = bench.build_pipeline(step, engine, x0, cut=cut, gate=gate)
    prime()
    for _ in range(4):                                        # even and odd graphs, steady state
        run()
        torch.cuda.synchronize()
        assert torch.equal(engine.pred, ref)
    assert torch.isfinite(ref).all() and ref.shape[0] == 2


@pytest.mark.parametrize("points", [None, [1, 2, 3, 4, 5], [0, 10, 20, 30, 60], [60, 61, 62, 63, 64]])
def test_interleaved_replay_equals_sequential_step(points):
    """bench.py's default arrangement (round 4): the next batch's ISP filters are issued BETWEEN the detector's layers on the
    detector's stream (YoloEngine.hook), its policy launches on a second stream. Same predictions as the sequential step for
    every placement of the five filters, as a graph replay and launched eagerly."""
    sys.path.insert(0, ROOT)
    import bench
    a = argparse.Namespace(batch=2, height=96, width=128, schedule="mixed", retune=False)
    bench.TUNE_CACHE = None
    step, engine, agent, x0, sched = bench.build_workload(a, torch.device("cuda:0"))
    ref = step().clone()
    torch.cuda.synchronize()
    assert engine.num_launches() > 64
    for eager in (False, True):
        prime, run = bench.build_interleaved(step, engine, x0, points=points, eager=eager)
        prime()
        for _ in range(4):
            run()
            torch.cuda.synchronize()
            assert torch.equal(engine.pred, ref)
        assert engine.hook is None
    with pytest.raises(ValueError):
        bench.build_interleaved(step, engine, x0, points=[5, 4, 3, 2, 1])


def test_full_size_chain_is_bit_reproducible_beside_the_detector():
    """Regression: with the detector's big-LDS workgroups running on a second stream, an earlier LDS + barrier form of
    the policy's fc1 kernel produced run-to-run different sums (tools/chain_stress5.py). The ISP episode must come out
    bit-identical every time, alone or beside the detector, and so must the detector."""
    sys.path.insert(0, ROOT)
    import bench
    bench.TUNE_CACHE = os.path.join(ROOT, "adaptiveisp_amd", "yolo", "tuning", "mi355x.json")
    a = argparse.Namespace(batch=8, height=720, width=1280, schedule="mixed", retune=False)
    step, engine, agent, x0, sched = bench.build_workload(a, torch.device("cuda:0"))
    xref = step.isp_chain().clone()
    pref = step().clone()
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    for i in range(25):
        with torch.cuda.stream(side), torch.no_grad():
            pred = engine(xref)
        x = step.isp_chain()
        torch.cuda.synchronize()
        assert torch.equal(x, xref), f"ISP episode differs in run {i}"
        assert torch.equal(pred, pref), f"detector output differs in run {i}"


def test_raw_bayer_start_pipelined_equals_sequential():
    """bench.py --raw: every step starts from a uint16 RGGB plane (adaisp_demosaic at the top of the episode). The pipelined
    replay must equal the sequential step, and the demosaiced batch must be what the ISP chain starts from."""
    sys.path.insert(0, ROOT)
    import bench
    from adaptiveisp_amd import _lib
    a = argparse.Namespace(batch=2, height=96, width=128, schedule="mixed", retune=False, raw=True)
    bench.TUNE_CACHE = None
    step, engine, agent, x0, sched = bench.build_workload(a, torch.device("cuda:0"))
    ref = step().clone()
    first = x0.clone()                                         # what the demosaic wrote
    assert torch.isfinite(first).all() and 0.0 <= float(first.min()) and float(first.max()) <= 1.0
    x0.fill_(float("nan"))                                     # the next step must rebuild it from the Bayer plane
    assert torch.equal(step(), ref) and torch.equal(x0, first)
    for build in (bench.build_pipeline, bench.build_interleaved):
        prime, run = build(step, engine, x0)
        prime()
        for _ in range(4):
            x0.fill_(float("nan"))
            run()
            torch.cuda.synchronize()
            assert torch.equal(engine.pred, ref)


def test_headline_pipeline_with_the_persistent_chain_is_bit_reproducible():
    """The headline's arrangement at the headline's size with the backbone's C = 256 stage as a persistent chain (round 5): the
    chain's workgroups hold their CUs for ~0.8 ms while the ISP stream's kernels (NLM's 48.5 KB workgroups, the policy's small
    launches) take whatever CUs are free — uneven residency, graph replay on two streams. Every replay must equal the eager
    sequential step bit for bit, and no dependency wait of the chain may have given up."""
    sys.path.insert(0, ROOT)
    import bench
    a = argparse.Namespace(batch=8, height=720, width=1280, schedule="mixed", retune=False)
    step, engine, agent, x0, sched = bench.build_workload(a, torch.device("cuda:0"))
    assert engine.chains and max(c["layers"] for c in engine.chains) >= 8
    ref = step().clone()
    xref = step.isp_chain().clone()
    torch.cuda.synchronize()
    for two in (False, True):                                # one graph with the detector forked inside / two one-stream graphs + events
        prime, run = bench.build_pipeline(step, engine, x0, two_graphs=two)
        prime()
        for i in range(24):
            run()
            torch.cuda.synchronize()
            assert torch.equal(run.xbuf[i & 1], xref), (two, i)
            assert torch.equal(engine.pred, ref), (two, i)
        # ... and without a host synchronisation between the steps (what the timed loop does)
        prime()
        for i in range(12):
            run()
        torch.cuda.synchronize()
        assert torch.equal(run.xbuf[11 & 1], xref) and torch.equal(engine.pred, ref), two
    assert engine.chain_status() == 0
