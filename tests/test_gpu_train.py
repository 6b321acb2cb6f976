"""RL training loop on the HIP path: replay pool in HBM -> agent (HIP ISP + parameter-gradient kernels) -> frozen
detector (autograd to the retouched image) -> TD losses -> clip/Adam/LR schedule -> records back into the pool."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _build(width=0.0625, nc=5, hw=(64, 96), pool=16, bs=4):
    from _synth import synth_state_dict, synth_yolo_state_dict
    from adaptiveisp_amd.agent import Agent
    from adaptiveisp_amd.config import cfg
    from adaptiveisp_amd.replay import DeviceReplayMemory, SyntheticSource
    from adaptiveisp_amd.util import Dict
    from adaptiveisp_amd.value import Value
    from adaptiveisp_amd.yolo.loss import DetectionLoss, default_hyp
    from adaptiveisp_amd.yolo.model import DetectionModel
    c = Dict(cfg)
    c.replay_memory_size = pool
    c.save_model_freq = 2
    agent = Agent(c, shape=(16, 64, 64), device=DEV)
    agent.load_state_dict(synth_state_dict(agent, seed=0))
    agent = agent.to(DEV)
    value = Value(c, shape=(19, 64, 64))
    value.load_state_dict(synth_state_dict(value, seed=1))
    value = value.to(DEV)
    det = DetectionModel(nc=nc, width=width)
    det.load_state_dict(synth_yolo_state_dict(det))
    det = det.to(DEV).train()
    for m in det.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()
    for p in det.parameters():
        p.requires_grad_(False)
    loss_fn = DetectionLoss(det.model[-1].anchors, nc=nc, hyp=default_hyp(nc, hw[1]), device=DEV)
    replay = DeviceReplayMemory(c, SyntheticSource((3,) + hw, nc=nc, seed=2), bs, DEV, (3,) + hw, rng=random.Random(5))
    return c, agent, value, det, loss_fn, replay


def test_trainer_runs_and_updates(tmp_path):
    from adaptiveisp_amd.train import Trainer
    from adaptiveisp_amd.yolo.checkpoint import ISP_KEYS, load_isp_checkpoint
    np.random.seed(0)
    torch.manual_seed(0)
    c, agent, value, det, loss_fn, replay = _build()
    tr = Trainer(c, agent, value, det, loss_fn, replay, batch_size=4, lr=3e-5, epochs=1, save_dir=str(tmp_path))
    assert tr.max_iter_step == 250                                       # epochs*1000//batch (train.py:156)
    before = torch.cat([p.detach().reshape(-1) for p in agent.parameters()]).clone()
    hist = tr.train(iters=5)
    torch.cuda.synchronize()
    assert len(hist) == 5 and all(np.isfinite([h["agent_loss"], h["value_loss"], h["reward"]]).all() for h in hist)
    after = torch.cat([p.detach().reshape(-1) for p in agent.parameters()])
    assert (after != before).any()
    assert len(replay.image_pool) == 16 and len(replay.image_pool) + len(replay.free) == replay.images.shape[0]
    assert replay.images.device.type == "cuda"                           # the pool never left HBM
    assert max(float(r.state[2]) for r in replay.image_pool) >= 1.0      # retouched records re-entered the pool
    # LR schedule 0.1^(3 it / max_it) (train.py:206-218)
    assert abs(tr.agent_scheduler.get_last_lr()[0] - 3e-5 * 0.1 ** (3 * 5 / 250)) < 1e-12
    ck = sorted(os.listdir(tmp_path))
    assert ck == ["ckpt-2.pth", "ckpt-4.pth"]
    raw = load_isp_checkpoint(str(tmp_path / "ckpt-4.pth"))
    assert tuple(raw) == ISP_KEYS and raw["iter"] == 4


def test_train_iteration_with_hip_detector_matches_torch_detector():
    """Same state, same batch: the iteration run with the HIP training engine (bf16 MFMA forward + HIP data-gradient
    path) gives the same reward / losses as with the fp32 PyTorch module tree, and a parallel gradient on the heads."""
    from _synth import synth_state_dict, synth_yolo_state_dict, test_image
    from adaptiveisp_amd import dist as adist
    from adaptiveisp_amd.agent import Agent
    from adaptiveisp_amd.config import cfg
    from adaptiveisp_amd.rl import train_iteration
    from adaptiveisp_amd.value import Value
    from adaptiveisp_amd.yolo import YoloTrainEngine, yolov3
    from adaptiveisp_amd.yolo.loss import DetectionLoss, default_hyp
    B, H, W = 4, 64, 96
    det = yolov3()
    det.load_state_dict(synth_yolo_state_dict(det))
    det = det.to(DEV).train()
    for m in det.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()
    for p in det.parameters():
        p.requires_grad_(False)
    Hp = (H + 31) // 32 * 32

    def torch_detector(x):                                   # letterbox like the engine's stem does
        top = (Hp - x.shape[2]) // 2
        pad = torch.full((x.shape[0], 3, Hp, x.shape[3]), 114.0 / 255.0, device=x.device)
        return det(torch.cat([pad[:, :, :top], x, pad[:, :, top + x.shape[2]:]], 2) if Hp != x.shape[2] else x)

    eng = YoloTrainEngine(det, B, H, W, device=DEV)
    loss_fn = DetectionLoss(det.model[-1].anchors, nc=80, hyp=default_hyp(80, W), device=DEV)
    imgs = torch.from_numpy(test_image(B, H, W, seed=3, special=False)).to(DEV)
    labels = [torch.tensor([[0, 1 + b, 0.5, 0.5, 0.3, 0.4]]) for b in range(B)]
    outs, grads = [], []
    for detector in (torch_detector, eng):
        torch.manual_seed(0)
        agent = Agent(cfg, shape=(16, 64, 64), device=DEV)
        agent.load_state_dict(synth_state_dict(agent, seed=0))
        agent = agent.to(DEV).train()
        value = Value(cfg, shape=(19, 64, 64))
        value.load_state_dict(synth_state_dict(value, seed=1))
        value = value.to(DEV).train()
        z = torch.full((B, cfg.z_dim), 0.37, device=DEV)
        states = torch.zeros(B, cfg.num_state_dim, device=DEV)
        opts = [torch.optim.SGD(agent.parameters(), lr=0.0), torch.optim.SGD(value.parameters(), lr=0.0)]
        captured = {}

        class Bucket(adist.GradBucket):                      # capture the pre-clip gradient
            def finish(self, work=None):
                super().finish(work)
                captured[id(self)] = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
                                                for p in self.params]).clone()

        bk = [Bucket(agent), Bucket(value)]
        torch.manual_seed(7)                                 # same dropout masks / sampling in both runs
        out = train_iteration(cfg, agent, value, detector, loss_fn, imgs, z, states, labels, 0.1, opts, buckets=bk)
        torch.cuda.synchronize()
        outs.append(out)
        grads.append(captured[id(bk[0])])
    a, b = outs
    # measured on MI355X: reward differs by <= 2.2e-3 (it is 100 x a DIFFERENCE of two detection losses, each carrying
    # the bf16 detector's ~1e-3 relative noise), value loss by 0.2 %, pre-clip head gradient: cosine 0.9999992,
    # relative error 0.33 %
    assert torch.allclose(a["reward"], b["reward"], rtol=0.02, atol=6e-3), (a["reward"], b["reward"])
    assert abs(float(a["value_loss"].detach()) - float(b["value_loss"].detach())) <= 0.01 * abs(float(a["value_loss"].detach()))
    cos = torch.nn.functional.cosine_similarity(grads[0].reshape(1, -1), grads[1].reshape(1, -1)).item()
    assert cos > 0.9995, cos
    rel = ((grads[0] - grads[1]).norm() / grads[0].norm()).item()
    assert rel < 0.02, rel


def test_train_iteration_at_the_config4_per_rank_shape():
    """BASELINE config 4, what ONE rank does per iteration: batch 8 x 512 x 512 through `build_trainer` (the entry
    `python -m adaptiveisp_amd.train` uses: full-width detector on the HIP training engine with the tuning table,
    replay pool in HBM) — two iterations, finite losses, heads move, records re-enter the pool."""
    from adaptiveisp_amd.config import cfg
    from adaptiveisp_amd.train import build_trainer
    from adaptiveisp_amd.util import Dict
    c = Dict(cfg)
    c.replay_memory_size = 16
    np.random.seed(0)
    cache = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "adaptiveisp_amd", "yolo", "tuning",
                         "mi355x.json")
    tr = build_trainer(c, 0, 1, torch.device(DEV), batch_size=8, image_size=512, tune_cache=cache)
    assert tr.max_iter_step == 800 * 1000 // 8
    before = torch.cat([p.detach().reshape(-1) for p in tr.agent.parameters()]).clone()
    hist = tr.train(iters=2)
    torch.cuda.synchronize()
    assert all(np.isfinite([h["agent_loss"], h["value_loss"], h["reward"]]).all() for h in hist)
    after = torch.cat([p.detach().reshape(-1) for p in tr.agent.parameters()])
    assert (after != before).any()
    assert tr.replay.images.shape[1:] == (3, 512, 512) and tr.replay.images.device.type == "cuda"
    # as in the reference: Adam holds state for every parameter that enters the loss and for none of the fc_mask heads
    # (gradients are released after the step — zero_grad(set_to_none=True) — so the optimizer state is what shows it)
    dead = [n for n, p in tr.agent.named_parameters() if p not in tr.agent_optimizer.state]
    assert dead and all("fc_mask" in n for n in dead)
    assert all(p.grad is None for p in tr.agent.parameters())


def test_input_batch_loss_on_a_second_stream_changes_nothing():
    """rl.train_iteration runs the detection loss of the INPUT batch on a second stream beside the agent's forward (the
    retouched batch's forward, on the same engine buffers, waits for it). Both per-image losses the iteration used must be
    exactly what the engine gives for those batches on one stream afterwards (same kernels, same inputs: bit for bit), over
    three iterations on changing inputs; and the one-stream order (ADAISP_TRAIN_OVERLAP=0) passes the same check."""
    from _synth import synth_state_dict, synth_yolo_state_dict, test_image
    from adaptiveisp_amd.agent import Agent
    from adaptiveisp_amd.config import cfg
    from adaptiveisp_amd.rl import train_iteration
    from adaptiveisp_amd.value import Value
    from adaptiveisp_amd.yolo import YoloTrainEngine, yolov3
    from adaptiveisp_amd.yolo.loss import DetectionLoss, assign_labels_packed, default_hyp
    B, H, W = 4, 64, 96
    det = yolov3()
    det.load_state_dict(synth_yolo_state_dict(det))
    det = det.to(DEV).train()
    for m in det.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()
    for p in det.parameters():
        p.requires_grad_(False)
    eng = YoloTrainEngine(det, B, H, W, device=DEV)
    loss_fn = DetectionLoss(det.model[-1].anchors, nc=80, hyp=default_hyp(80, W), device=DEV)
    for overlap in ("1", "0"):
        os.environ["ADAISP_TRAIN_OVERLAP"] = overlap
        try:
            torch.manual_seed(0)
            agent = Agent(cfg, shape=(16, 64, 64), device=DEV)
            agent.load_state_dict(synth_state_dict(agent, seed=0))
            agent = agent.to(DEV).train()
            value = Value(cfg, shape=(19, 64, 64))
            value.load_state_dict(synth_state_dict(value, seed=1))
            value = value.to(DEV).train()
            opts = [torch.optim.Adam(agent.parameters(), lr=1e-3), torch.optim.Adam(value.parameters(), lr=1e-3)]
            for i in range(3):
                imgs = torch.from_numpy(test_image(B, H, W, seed=30 + i, special=False)).to(DEV)
                z = torch.full((B, cfg.z_dim), 0.2 + 0.25 * i, device=DEV)
                states = torch.zeros(B, cfg.num_state_dim, device=DEV)
                labels = [torch.tensor([[0, 1 + b + i, 0.5, 0.5, 0.3, 0.4]]) for b in range(B)]
                out = train_iteration(cfg, agent, value, eng, loss_fn, imgs, z, states, labels, 0.1, opts)
                torch.cuda.synchronize()
                with torch.no_grad():
                    packed = assign_labels_packed(loss_fn, eng.head_shapes(), labels, imgs.device)
                    l_in = eng.per_sample_loss(loss_fn, imgs, packed).clone()
                    l_re = eng.per_sample_loss(loss_fn, out["retouch"], packed).clone()
                assert torch.equal(out["detect_loss_input"], l_in), (overlap, i)
                assert torch.equal(out["detect_loss_retouch"], l_re), (overlap, i)
                assert float((l_in - l_re).abs().max()) > 0          # the two batches do differ: a mixed-up pass would show
                assert torch.isfinite(out["reward"]).all()
        finally:
            del os.environ["ADAISP_TRAIN_OVERLAP"]


def test_critic_on_a_second_stream_changes_nothing():
    """rl.train_iteration with the pair engine runs the critic's two calls (and, through autograd's stream rule, their backward)
    on a second stream beside the detector (ADAISP_CRITIC_STREAM, default on). Three iterations from the same state with the
    switch on and off: the same losses, rewards and retouched images bit for bit, the same updated parameters (gradient sums
    reach the shared leaves in another order: fp32 rounding)."""
    from _margins import close_scaled
    from _synth import synth_state_dict, synth_yolo_state_dict, test_image
    from adaptiveisp_amd.agent import Agent
    from adaptiveisp_amd.config import cfg
    from adaptiveisp_amd.rl import train_iteration
    from adaptiveisp_amd.value import Value
    from adaptiveisp_amd.yolo import YoloTrainPairEngine, yolov3
    from adaptiveisp_amd.yolo.loss import DetectionLoss, default_hyp
    B, H, W = 4, 64, 96
    det = yolov3()
    det.load_state_dict(synth_yolo_state_dict(det))
    det = det.to(DEV).train()
    for p in det.parameters():
        p.requires_grad_(False)
    eng = YoloTrainPairEngine(det, B, H, W, device=DEV)
    loss_fn = DetectionLoss(det.model[-1].anchors, nc=80, hyp=default_hyp(80, W), device=DEV)
    runs = {}
    for sw in ("1", "0"):
        os.environ["ADAISP_CRITIC_STREAM"] = sw
        try:
            torch.manual_seed(0)
            agent = Agent(cfg, shape=(16, 64, 64), device=DEV)
            agent.load_state_dict(synth_state_dict(agent, seed=0))
            agent = agent.to(DEV).train()
            agent.feature_extractor.droupout.p = agent.action_selection.droupout.p = 0.0
            value = Value(cfg, shape=(19, 64, 64))
            value.load_state_dict(synth_state_dict(value, seed=1))
            value = value.to(DEV).train()
            # plain SGD: the update is proportional to the (clipped) gradient, so a rounding-level difference between the two
            # arrangements stays one (Adam's first steps are lr * sign(g): the sign of a conv bias's rounding-noise gradient in
            # front of a batch-statistics BatchNorm would decide a 1e-3 move)
            opts = [torch.optim.SGD(agent.parameters(), lr=10.0), torch.optim.SGD(value.parameters(), lr=10.0)]
            outs = []
            for i in range(3):
                imgs = torch.from_numpy(test_image(B, H, W, seed=30 + i, special=False)).to(DEV)
                z = torch.full((B, cfg.z_dim), 0.2 + 0.25 * i, device=DEV)
                states = torch.zeros(B, cfg.num_state_dim, device=DEV)
                labels = [torch.tensor([[0, 1 + b + i, 0.5, 0.5, 0.3, 0.4]]) for b in range(B)]
                out = train_iteration(cfg, agent, value, eng, loss_fn, imgs, z, states, labels, 0.1, opts)
                torch.cuda.synchronize()
                outs.append({k: out[k].detach().clone() for k in ("retouch", "reward", "value_loss", "agent_loss", "detect_loss_input",
                                                                   "detect_loss_retouch")})
            runs[sw] = (outs, [p.detach().clone() for p in list(agent.parameters()) + list(value.parameters())])
        finally:
            del os.environ["ADAISP_CRITIC_STREAM"]
    for a, b in zip(runs["1"][0][:1], runs["0"][0][:1]):                    # first iteration: identical state, identical forward
        for k in a:
            assert torch.equal(a[k], b[k]), k
    for i, (a, b) in enumerate(zip(runs["1"][0], runs["0"][0])):
        for k in a:
            close_scaled("train.critic_stream." + k, a[k], b[k], 2e-4, err_msg=f"iteration {i}")
    for pa, pb in zip(runs["1"][1], runs["0"][1]):
        close_scaled("train.critic_stream.params", pa, pb, 2e-4)
