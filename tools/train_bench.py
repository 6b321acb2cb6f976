#!/usr/bin/env python3
"""RL training iterations per second on one GPU (BASELINE config 4 per-rank shape: batch 8 x 512 x 512), detector on
the HIP training engine vs the PyTorch module tree. usage: train_bench.py [iters=6] [B=8] [HW=512]"""
import os, random, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptiveisp_amd.agent import Agent
from adaptiveisp_amd.config import cfg
from adaptiveisp_amd.replay import DeviceReplayMemory, SyntheticSource
from adaptiveisp_amd.train import Trainer
from adaptiveisp_amd.value import Value
from adaptiveisp_amd.yolo import YoloTrainEngine, yolov3
from adaptiveisp_amd.yolo.loss import DetectionLoss, default_hyp

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
HW = int(sys.argv[3]) if len(sys.argv) > 3 else 512
DEV = "cuda:0"
if os.environ.get("CUDNN_BENCHMARK"):
    torch.backends.cudnn.benchmark = os.environ["CUDNN_BENCHMARK"] == "1"
torch.manual_seed(0); np.random.seed(0)
det = yolov3().to(DEV).train()
for m in det.modules():
    if isinstance(m, torch.nn.BatchNorm2d):
        m.eval()
for p in det.parameters():
    p.requires_grad_(False)
for name in ([os.environ["TRAIN_BENCH_ONLY"]] if os.environ.get("TRAIN_BENCH_ONLY") else ["hip", "torch"]):
    agent = Agent(cfg, shape=(16, 64, 64), device=DEV).to(DEV)
    value = Value(cfg, shape=(19, 64, 64)).to(DEV)
    if os.environ.get("TRAIN_BENCH_CHANNELS_LAST") == "1":           # experiment: NHWC weights for the policy / critic trunks
        agent = agent.to(memory_format=torch.channels_last)
        value = value.to(memory_format=torch.channels_last)
    loss_fn = DetectionLoss(det.model[-1].anchors, nc=80, hyp=default_hyp(80, HW), device=DEV)
    replay = DeviceReplayMemory(cfg, SyntheticSource((3, HW, HW), seed=1, device=DEV), B, DEV, (3, HW, HW), rng=random.Random(1))
    if name == "hip":
        from adaptiveisp_amd.yolo import YoloTrainPairEngine
        detector = (YoloTrainPairEngine if os.environ.get("ADAYOLO_TRAIN_PAIR", "1") == "1" else YoloTrainEngine)(det, B, HW, HW, device=DEV)
    else:
        detector = det
    if name == "hip":
        detector.autotune(cache=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "adaptiveisp_amd", "yolo", "tuning", "mi355x.json"))
    tr = Trainer(cfg, agent, value, detector, loss_fn, replay, batch_size=B)
    tr.train(2)
    torch.cuda.synchronize()
    waited = [0.0]
    _ev_sync, _cpu = torch.cuda.Event.synchronize, torch.Tensor.cpu

    def _timed(fn):
        def call(*a, **k):
            t = time.perf_counter()
            r = fn(*a, **k)
            waited[0] += time.perf_counter() - t
            return r
        return call
    torch.cuda.Event.synchronize, torch.Tensor.cpu = _timed(_ev_sync), _timed(_cpu)   # the host's waits for the GPU
    prof = None
    if os.environ.get("TRAIN_BENCH_PROFILE"):                     # host profile of the steady state only (tools/train_host_profile.sh)
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    tr.train(iters)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    if prof is not None:
        prof.disable()
        prof.dump_stats(os.environ["TRAIN_BENCH_PROFILE"])
    torch.cuda.Event.synchronize, torch.Tensor.cpu = _ev_sync, _cpu
    print(f"      host: {t_host / iters * 1e3:.1f} ms / iteration, of which {waited[0] / iters * 1e3:.1f} ms waiting for the GPU (guard event, "
          f".cpu() copies) -> {(t_host - waited[0]) / iters * 1e3:.1f} ms of enqueue work")
    # detector forward+backward alone
    x = torch.rand(B, 3, HW, HW, device=DEV, requires_grad=True)
    if hasattr(detector, "half"):                                 # pair engine: its B-image member (own forward here)
        detector = YoloTrainEngine(det, B, HW, HW, device=DEV)
        detector.autotune(cache=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "adaptiveisp_amd", "yolo", "tuning", "mi355x.json"), write=False)
    for _ in range(2):
        sum(r.float().sum() for r in detector(x)).backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        sum(r.float().sum() for r in detector(x)).backward()
    torch.cuda.synchronize()
    dd = (time.perf_counter() - t0) / 3
    print(f"{name:5s} detector: {dt * 1e3:8.1f} ms / iteration ({B / dt:6.1f} images/s), detector fwd+bwd alone {dd * 1e3:7.1f} ms")
