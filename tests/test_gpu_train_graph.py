"""The RL iteration as one hipGraph (train.Trainer graph mode, train._GraphIteration): what enters the captured launches through
device memory — the padded label tables, the entropy coefficient, the learning rates — gives the results of the by-value
arguments bit for bit, and a replayed iteration is the ordinary iteration (train.py:234-351)."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _detector(B, H, W, nc=80):
    from _synth import synth_yolo_state_dict
    from adaptiveisp_amd.yolo import YoloTrainPairEngine, yolov3
    from adaptiveisp_amd.yolo.loss import DetectionLoss, default_hyp
    det = yolov3()
    det.load_state_dict(synth_yolo_state_dict(det))
    det = det.to(DEV).train()
    for m in det.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()
    for p in det.parameters():
        p.requires_grad_(False)
    eng = YoloTrainPairEngine(det, B, H, W, device=DEV)
    return eng, DetectionLoss(det.model[-1].anchors, nc=nc, hyp=default_hyp(nc, W), device=DEV)


def _labels(B, i, many=False):
    g = torch.Generator().manual_seed(100 + i)
    out = []
    for b in range(B):
        k = 1 + (b + i) % 3 + (4 if many else 0)
        lb = torch.zeros(k, 6)
        lb[:, 1] = torch.randint(0, 80, (k,), generator=g).float()
        lb[:, 2:4] = torch.rand(k, 2, generator=g) * 0.6 + 0.2
        lb[:, 4:6] = torch.rand(k, 2, generator=g) * 0.3 + 0.05
        out.append(lb)
    return out


def test_padded_label_tables_give_the_exact_tables_results():
    """yolo.loss.StaticLabelTables (fixed row count, rows of image -1 behind the iteration's own) against the exact-size tables
    of assign_labels_packed through the pair engine: both per-image losses and the image gradient, bit for bit — also with a
    label centred on the same cell twice (the same-cell scans walk the padded rows too)."""
    from _synth import test_image
    from adaptiveisp_amd.yolo.loss import StaticLabelTables, assign_labels_packed
    B, H, W = 4, 64, 96
    eng, loss_fn = _detector(B, H, W)
    tables = StaticLabelTables(loss_fn, eng.head_shapes(), B, DEV, cap=256)
    for i in range(3):
        labels = _labels(B, i)
        if i == 2:
            labels[1] = torch.cat([labels[1], labels[1][:1]], 0)            # two labels on one cell
        imgs = torch.from_numpy(test_image(B, H, W, seed=40 + i, special=False)).to(DEV)
        res = []
        for which in ("exact", "padded"):
            if which == "exact":
                packed, pair = assign_labels_packed(loss_fn, eng.head_shapes(), labels, DEV, pair=True)
            else:
                assert tables.fill(labels)
                tables.upload()
                packed, pair = tables.packed, tables.packed_pair
            ret = (imgs * 0.9 + 0.01).requires_grad_(True)
            l_in, l_re = eng.per_sample_loss_pair(loss_fn, imgs, ret, packed, pair)
            (l_re * torch.arange(1, B + 1, device=DEV).view(B, 1)).sum().backward()
            torch.cuda.synchronize()
            res.append((l_in.detach().clone(), l_re.detach().clone(), ret.grad.clone()))
        for a, b, name in zip(res[0], res[1], ("l_in", "l_re", "d image")):
            assert torch.equal(a, b), (i, name)
        assert float(res[0][2].abs().max()) > 0
    assert not tables.fill(_labels(B, 0, many=True) * 8)                     # more rows than the tables hold: refused, nothing written


def test_device_scalars_give_the_by_value_results():
    """adaisp_policy_tail_args.entropy_coef_dev and adaisp_clip_adam_step_dev against their by-value forms: bit for bit."""
    from _synth import synth_state_dict
    from adaptiveisp_amd import optim as aoptim
    from adaptiveisp_amd.agent import Agent
    from adaptiveisp_amd.config import cfg
    from adaptiveisp_amd import policy_train
    B = 4
    torch.manual_seed(0)
    agent = Agent(cfg, shape=(16, 64, 64), device=DEV)
    agent.load_state_dict(synth_state_dict(agent, seed=0))
    agent = agent.to(DEV).train()
    F, pw = len(agent.filters), agent._param_width
    g = torch.Generator(device=DEV).manual_seed(3)
    x0 = torch.randn(B, F, pw, generator=g, device=DEV)
    lg0 = torch.randn(B, F, generator=g, device=DEV)
    noise = torch.rand(B, 1, generator=g, device=DEV)
    st = torch.zeros(B, cfg.num_state_dim, device=DEV)
    coef = 0.3 * cfg.exploration_penalty
    outs = []
    for c in (coef, torch.tensor([coef], dtype=torch.float32, device=DEV)):
        x, lg = x0.clone().requires_grad_(True), lg0.clone().requires_grad_(True)
        assert policy_train.serves(agent, x, lg, c)
        packed, op_ids, selected, sur, pen, ns, pdf, table = policy_train.policy_tail(agent, x, lg, noise, st, c)
        (pen.sum() * 1.5 + sur.sum() + (packed * packed).sum()).backward()
        outs.append([t.detach().clone() for t in (packed, op_ids, selected, sur, pen, ns, pdf, table, x.grad, lg.grad)])
    for k, (a, b) in enumerate(zip(*outs)):
        assert torch.equal(a, b), k
    assert float(outs[0][4].abs().max()) > 0 and float(outs[0][9].abs().max()) > 0
    # ... and through Agent.forward: the device scalar replaces (1 - progress) * exploration_penalty
    agent.feature_extractor.droupout.p = agent.action_selection.droupout.p = 0.0
    img = torch.rand(B, 3, 64, 64, device=DEV)
    z = torch.full((B, cfg.z_dim), 0.37, device=DEV)
    fw = []
    for dev_scalar in (False, True):
        agent.entropy_coef_dev = torch.tensor([coef], dtype=torch.float32, device=DEV) if dev_scalar else None
        try:
            with torch.no_grad():
                (ret, ns, sur, pen), _, _ = agent((img, z, st), 0.0 if dev_scalar else 0.7)     # (progress is ignored with the scalar)
        finally:
            agent.entropy_coef_dev = None
        fw.append([t.clone() for t in (ret, ns, sur, pen)])
    for k, (a, b) in enumerate(zip(*fw)):
        assert torch.equal(a, b), k
    # clip + Adam: two copies of one model, same gradients, lr by value / from the device
    nets = [torch.nn.Sequential(torch.nn.Linear(300, 70), torch.nn.Linear(70, 5)).to(DEV) for _ in range(2)]
    nets[1].load_state_dict(nets[0].state_dict())
    opts = [torch.optim.Adam(n.parameters(), lr=1e-3, fused=True) for n in nets]
    lr_dev = torch.zeros(1, dtype=torch.float64, device=DEV)
    for it in range(4):
        g = torch.Generator(device=DEV).manual_seed(it)
        grads = [torch.randn(p.shape, generator=g, device=DEV) for p in nets[0].parameters()]
        lr = 1e-3 * 0.5 ** it
        for k, (n, o) in enumerate(zip(nets, opts)):
            for p, gr in zip(n.parameters(), grads):
                p.grad = gr.clone()
            o.param_groups[0]["lr"] = lr if k == 0 else 123.0           # (the device form must not look at the group's rate)
            if it == 0:
                o.param_groups[0]["lr"] = lr
                torch.nn.utils.clip_grad_norm_(list(n.parameters()), 1e-5)
                o.step()                                                    # creates the state (the kernels need it)
            else:
                lr_dev.fill_(lr)
                assert aoptim.clip_adam_step(o, 1e-5, lr_dev=lr_dev if k == 1 else None)
    torch.cuda.synchronize()
    for a, b in zip(nets[0].parameters(), nets[1].parameters()):
        assert torch.equal(a, b)


def _fresh(B, dropout=0.0):
    from _synth import synth_state_dict
    from adaptiveisp_amd.agent import Agent
    from adaptiveisp_amd.config import cfg
    from adaptiveisp_amd.value import Value
    torch.manual_seed(0)
    agent = Agent(cfg, shape=(16, 64, 64), device=DEV)
    agent.load_state_dict(synth_state_dict(agent, seed=0))
    agent = agent.to(DEV).train()
    agent.feature_extractor.droupout.p = agent.action_selection.droupout.p = dropout
    value = Value(cfg, shape=(19, 64, 64))
    value.load_state_dict(synth_state_dict(value, seed=1))
    value = value.to(DEV).train()
    return cfg, agent, value


def test_a_replayed_iteration_is_the_ordinary_iteration():
    """From ONE state (two ordinary iterations, then a snapshot of models and optimizers): two more iterations through
    rl.train_iteration, and the same two as replays of ONE capture (train._GraphIteration) with other images, labels, noise,
    entropy coefficient and learning rates each time. The first replay starts from the identical state: everything its forward
    computes is bit-identical; the detection loss of the INPUT batch — labels and pixels only — also at the second. What passes
    through the backward's atomics (the updated parameters, hence the second iteration's forward) agrees to rounding. And the
    scalars are live: an ordinary second iteration with the FIRST iteration's coefficient is far off."""
    import copy
    import types

    from _margins import close_scaled
    from _synth import test_image
    from adaptiveisp_amd.rl import train_iteration
    from adaptiveisp_amd.train import _GraphIteration
    B, H, W = 4, 64, 96
    eng, loss_fn = _detector(B, H, W)
    cfg, agent, value = _fresh(B)
    opts = [torch.optim.Adam(agent.parameters(), lr=3e-5, fused=True), torch.optim.Adam(value.parameters(), lr=3e-5, fused=True)]

    def feed(i):
        return dict(im=torch.from_numpy(test_image(B, H, W, seed=30 + i, special=False)).to(DEV),
                    z=torch.full((B, cfg.z_dim), 0.2 + 0.2 * i, device=DEV), state=torch.zeros(B, cfg.num_state_dim, device=DEV),
                    label=_labels(B, i))
    for i in range(2):                                           # Adam's state exists, every kernel has run
        f = feed(i)
        train_iteration(cfg, agent, value, eng, loss_fn, f["im"], f["z"], f["state"], f["label"], 0.1, opts)
    torch.cuda.synchronize()
    snap = copy.deepcopy((agent.state_dict(), value.state_dict(), opts[0].state_dict(), opts[1].state_dict()))

    def restore():
        agent.load_state_dict(snap[0])
        value.load_state_dict(snap[1])
        opts[0].load_state_dict(copy.deepcopy(snap[2]))
        opts[1].load_state_dict(copy.deepcopy(snap[3]))
    KEYS = ("retouch", "new_states", "reward", "value_loss", "agent_loss", "detect_loss_input", "detect_loss_retouch")
    sched = [(0.25, 3e-5, 3e-4), (0.75, 1e-5, 1e-4)]             # (progress, agent lr, critic lr) of the two iterations

    def ordinary(stale_coef=False):
        restore()
        outs = []
        for i, (prog, lra, lrv) in enumerate(sched):
            f = feed(2 + i)
            opts[0].param_groups[0]["lr"], opts[1].param_groups[0]["lr"] = lra, lrv
            out = train_iteration(cfg, agent, value, eng, loss_fn, f["im"], f["z"], f["state"], f["label"],
                                  sched[0][0] if stale_coef else prog, opts)
            torch.cuda.synchronize()
            outs.append({k: out[k].detach().clone() for k in KEYS})
        return outs, [p.detach().clone() for p in list(agent.parameters()) + list(value.parameters())]
    ref, ref_params = ordinary()
    stale, _ = ordinary(stale_coef=True)
    restore()
    tr = types.SimpleNamespace(cfg=cfg, agent=agent, value=value, detector=eng, loss_fn=loss_fn, batch_size=B, max_bri=0.9,
                               use_truncated=True, agent_optimizer=opts[0], value_optimizer=opts[1], buckets=None)
    pool = torch.cat([feed(2)["im"], feed(3)["im"]], 0)          # the "replay pool": the graph gathers its batch from it by row
    pool0 = pool.clone()
    G = _GraphIteration(tr, pool, cap=256)
    got = []
    for i, (prog, lra, lrv) in enumerate(sched):
        f = feed(2 + i)
        rows = list(range(i * B, (i + 1) * B))
        assert G.stage(f["label"], rows, f["state"].cpu().numpy(), f["z"].cpu().numpy(), (1.0 - prog) * cfg.exploration_penalty, lra, lrv)
        if G.graph is None:
            G.capture()
        G.replay()
        bad, states_host = G.wait_guard(timeout=30.0)
        torch.cuda.synchronize()
        got.append({k: G.out[k].detach().clone() for k in KEYS})
        assert not bad
        assert np.array_equal(states_host, got[-1]["new_states"].cpu().numpy())          # what the host read mid-iteration
        assert torch.equal(pool[rows], got[-1]["retouch"])                               # replace_memory's scatter, inside the graph
        kept = G.kept_scalars()
        assert float(kept[0]) == float(got[-1]["agent_loss"]) and float(kept[1]) == float(got[-1]["value_loss"])
        assert float(kept[2]) == float(got[-1]["reward"].mean())
    assert torch.equal(pool[B:], got[1]["retouch"]) and not torch.equal(pool[:B], pool0[:B])
    params = [p.detach().clone() for p in list(agent.parameters()) + list(value.parameters())]
    for k in KEYS:
        assert torch.equal(got[0][k], ref[0][k]), k
    assert torch.equal(got[1]["detect_loss_input"], ref[1]["detect_loss_input"])
    assert not torch.equal(ref[0]["detect_loss_input"], ref[1]["detect_loss_input"])
    for k in KEYS:
        close_scaled("train.graph.second_iteration." + k, got[1][k], ref[1][k], 2e-3)
    for j, (a, b) in enumerate(zip(params, ref_params)):
        close_scaled("train.graph.parameters", a, b, 2e-3, err_msg=f"parameter {j}")
    # the scalars are read at every replay: with the first iteration's coefficient the second agent loss is somewhere else
    gap_stale = abs(float(stale[1]["agent_loss"]) - float(ref[1]["agent_loss"]))
    gap_graph = abs(float(got[1]["agent_loss"]) - float(ref[1]["agent_loss"]))
    assert gap_stale > 100 * max(gap_graph, 1e-7), (gap_stale, gap_graph)


def test_a_dropped_batch_leaves_the_pool_as_it_is():
    """The guard of train.py:374-381 inside the captured iteration: with a brightness bound every retouched batch violates, the
    host reads `bad`, and the scatter behind the guard keeps the pool's rows (pool[slots] = flag ? pool[slots] : retouch)."""
    import types

    from _synth import test_image
    from adaptiveisp_amd.rl import train_iteration
    from adaptiveisp_amd.train import _GraphIteration
    B, H, W = 4, 64, 96
    eng, loss_fn = _detector(B, H, W)
    cfg, agent, value = _fresh(B)
    opts = [torch.optim.Adam(agent.parameters(), lr=3e-5, fused=True), torch.optim.Adam(value.parameters(), lr=3e-5, fused=True)]
    im = torch.from_numpy(test_image(B, H, W, seed=31, special=False)).to(DEV)
    z, st = torch.full((B, cfg.z_dim), 0.4, device=DEV), torch.zeros(B, cfg.num_state_dim, device=DEV)
    train_iteration(cfg, agent, value, eng, loss_fn, im, z, st, _labels(B, 0), 0.1, opts)      # Adam's state exists
    torch.cuda.synchronize()
    tr = types.SimpleNamespace(cfg=cfg, agent=agent, value=value, detector=eng, loss_fn=loss_fn, batch_size=B, max_bri=-1.0,
                               use_truncated=True, agent_optimizer=opts[0], value_optimizer=opts[1], buckets=None)
    pool = torch.cat([im, im * 0.5], 0)
    pool0 = pool.clone()
    G = _GraphIteration(tr, pool, cap=256)
    assert G.stage(_labels(B, 1), [4, 5, 6, 7], st.cpu().numpy(), z.cpu().numpy(), 0.01, 3e-5, 3e-5)
    G.capture()
    G.replay()
    bad, _ = G.wait_guard(timeout=30.0)
    torch.cuda.synchronize()
    assert bad and torch.equal(pool, pool0)
    assert torch.equal(G.im, pool0[4:]) and not torch.equal(G.out["retouch"], pool0[4:])


def test_graph_trainer_runs_the_schedule_and_keeps_the_pool():
    """train.Trainer in graph mode over the pair engine: three ordinary iterations, the capture, nine replays — finite losses,
    the LambdaLR rates of train.py:206-218 on the device scalars, records re-entering the pool, Adam's step counts at 12 — as one
    graph per iteration and in the form data parallelism uses (graph="split": forward + backward | the collective, outside |
    clip + Adam). Against a trainer of the same seed in the ordinary loop the history agrees as far as two ordinary runs agree
    with each other: one fp32 ulp in a filter parameter flips bf16 roundings in the detector, and the per-image losses move in
    the fourth digit (tools/train_graph_hist.py: ordinary vs ordinary 1e-4 .. 3e-4 from the fourth iteration on, graph vs
    ordinary the same); the tolerance is measured (tests/_margins.py). That a replay IS the ordinary iteration, bit for bit, is
    the test above."""
    from _margins import close_scaled
    from adaptiveisp_amd.replay import DeviceReplayMemory, SyntheticSource
    from adaptiveisp_amd.train import Trainer
    from adaptiveisp_amd.util import Dict
    B, H, W, N = 4, 64, 96, 12
    eng, loss_fn = _detector(B, H, W)
    hist = {}
    for mode in (False, True, "split"):
        cfg, agent, value = _fresh(B)
        c = Dict(cfg)
        c.replay_memory_size = 16
        np.random.seed(0)
        replay = DeviceReplayMemory(c, SyntheticSource((3, H, W), nc=80, seed=2), B, DEV, (3, H, W), rng=random.Random(5))
        tr = Trainer(c, agent, value, eng, loss_fn, replay, batch_size=B, lr=3e-5, epochs=1, graph=mode)
        assert tr.graph_mode is bool(mode) and tr.graph_split is (mode == "split")
        h = tr.train(iters=N)
        torch.cuda.synchronize()
        assert len(h) == N and all(np.isfinite([r["agent_loss"], r["value_loss"], r["reward"]]).all() for r in h)
        assert abs(tr.agent_scheduler.get_last_lr()[0] - 3e-5 * 0.1 ** (3 * N / 250)) < 1e-12
        assert len(replay.image_pool) == 16 and len(replay.image_pool) + len(replay.free) == replay.images.shape[0]
        assert max(float(r.state[2]) for r in replay.image_pool) >= 1.0
        steps = {float(s["step"]) for s in tr.agent_optimizer.state.values()}
        assert steps == {float(N)}, steps
        if mode:
            G = tr._git
            assert G is not None and G.graph is not None and G._replays == N - tr.graph_warmup
            assert (G.graph_step is not None) is (mode == "split")
            lr_last = 3e-5 * 0.1 ** (3 * (N - 1) / 250)                      # the rate the LAST iteration stepped with
            assert abs(float(G.lr[0]) - lr_last) < 1e-18 and abs(float(G.lr[1]) - lr_last * float(c.value_lr_mul)) < 1e-17
            assert abs(float(G.coef) - np.float32((1.0 - (N - 1) / 250) * c.exploration_penalty)) == 0.0
        hist[mode] = h
    for mode in (True, "split"):
        for k in ("agent_loss", "value_loss", "reward"):
            a = torch.tensor([r[k] for r in hist[mode]])
            b = torch.tensor([r[k] for r in hist[False]])
            close_scaled("train.graph.history." + k, a, b, 5e-2)
        assert [r["dropped"] for r in hist[mode]] == [r["dropped"] for r in hist[False]]
