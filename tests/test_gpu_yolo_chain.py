"""Persistent chain of 256 x 256-tile conv layers (adayolo_conv_chain_*, csrc/yolo_conv_pp.hip: k_conv_chain) against the SAME
layers launched one by one (adayolo_conv_fwd_variant 50 / adayolo_conv_fused1x1_fwd — themselves pinned against fp32 F.conv2d
in test_gpu_yolo_variants.py, i.e. Conv / Bottleneck of yolov3/models/common.py:45-59,110-120): the tile code is the same, so
the results must be BIT-identical; what is tested is the chain's own machinery — work counter, arrival counters over halo and
residual tiles, written-through stores + L1 invalidate, counter reset per launch, graph replay — under conditions that
expose a missing dependency or a stale cache line: outputs poisoned before every launch, inputs changed between launches,
a stride-2 consumer at the end, partial last tiles, and a second stream keeping the CUs unevenly busy."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _bf(*shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(torch.bfloat16).to(DEV)


def _pack_w2(w2):                       # [128][256] -> fragment-major (include/adayolo.h)
    return w2.reshape(4, 32, 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous()


def _check(net, run, want, tag):
    torch.cuda.synchronize()
    st = run.status()
    head = run.keep[1][:64].view(torch.int32).tolist()
    assert st == 0, (tag, st, head)
    for i, (got, ref) in enumerate(zip(net.outputs(), want)):
        assert torch.equal(got, ref), (tag, i, head)


class _Net:
    """x -(3x3 s1 c0->256, +1x1)-> x0,h0 -(3x3 128->256 + x0, +1x1)-> x1,h1 ... -(3x3 128->256 + x_{n-1})-> x_n -(3x3 s2 256->512)-> y"""

    def __init__(self, B, H, W, c0=64, blocks=3, tail=True, seed=0):
        self.B, self.H, self.W = B, H, W
        self.x = _bf(B, H, W, c0, seed=seed)
        self.layers = []                # dicts: in, w, b, res, out, k, s, (w2, w2p, b2, out2)
        cur_in, cur_res, cin = self.x, None, c0
        for i in range(blocks + 1):
            last = i == blocks
            ly = dict(inp=cur_in, w=_bf(256, 3, 3, cin, seed=seed + 10 + i, scale=(9 * cin) ** -0.5),
                      b=torch.randn(256, generator=torch.Generator().manual_seed(seed + 30 + i)).to(DEV) * 0.1, res=cur_res,
                      out=torch.empty((B, H, W, 256), dtype=torch.bfloat16, device=DEV), k=3, s=1, cin=cin, cout=256)
            if not last:
                ly["w2"] = _bf(128, 256, seed=seed + 50 + i, scale=256 ** -0.5)
                ly["w2p"] = _pack_w2(ly["w2"])
                ly["b2"] = torch.randn(128, generator=torch.Generator().manual_seed(seed + 70 + i)).to(DEV) * 0.1
                ly["out2"] = torch.empty((B, H, W, 128), dtype=torch.bfloat16, device=DEV)
                cur_in, cin = ly["out2"], 128
            cur_res = ly["out"]
            self.layers.append(ly)
        if tail:
            Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
            self.layers.append(dict(inp=cur_res, w=_bf(512, 3, 3, 256, seed=seed + 90, scale=(9 * 256) ** -0.5),
                                    b=torch.zeros(512, device=DEV), res=None,
                                    out=torch.empty((B, Ho, Wo, 512), dtype=torch.bfloat16, device=DEV), k=3, s=2, cin=256, cout=512))

    def outputs(self):
        return [t for ly in self.layers for t in ([ly["out"]] + ([ly["out2"]] if "out2" in ly else []))]

    def poison(self):
        for t in self.outputs():
            t.fill_(float("nan"))

    def run_separately(self):
        from adaptiveisp_amd.yolo import _lib
        L, st, vp = _lib.load(), _lib.stream_ptr(), ctypes.c_void_p
        for ly in self.layers:
            Bq, Hq, Wq, _ = ly["inp"].shape
            common = (vp(ly["inp"].data_ptr()), ly["cin"], vp(ly["w"].data_ptr()), vp(ly["b"].data_ptr()),
                      vp(ly["res"].data_ptr()) if ly["res"] is not None else None, ly["cout"] if ly["res"] is not None else 0,
                      vp(ly["out"].data_ptr()), ly["cout"], Bq, Hq, Wq, ly["cin"], ly["cout"], ly["k"], ly["s"], _lib.ACT_SILU)
            if "w2" in ly:
                rc = L.adayolo_conv_fused1x1_fwd(*common, vp(ly["w2p"].data_ptr()), vp(ly["b2"].data_ptr()), vp(ly["out2"].data_ptr()), 128, 128, st)
            else:
                rc = L.adayolo_conv_fwd_variant(*common, 60 if ly.get("tile") else 50, st)
            _lib.check(rc, "separate launch")

    def chain(self):
        from adaptiveisp_amd.yolo import _lib
        L, vp = _lib.load(), ctypes.c_void_p
        n = len(self.layers)
        arr = (_lib.ChainLayer * n)()
        for c, ly in zip(arr, self.layers):
            Bq, Hq, Wq, _ = ly["inp"].shape
            c.in_, c.in_cstride, c.weight, c.bias = ly["inp"].data_ptr(), ly["cin"], ly["w"].data_ptr(), ly["b"].data_ptr()
            c.residual, c.res_cstride = (ly["res"].data_ptr(), ly["cout"]) if ly["res"] is not None else (None, 0)
            c.out, c.out_cstride = ly["out"].data_ptr(), ly["cout"]
            c.B, c.H, c.W, c.Cin, c.Cout, c.ksize, c.stride, c.act = Bq, Hq, Wq, ly["cin"], ly["cout"], ly["k"], ly["s"], _lib.ACT_SILU
            c.tile = ly.get("tile", 0)
            if "w2" in ly:
                c.weight2, c.bias2, c.out2, c.out2_cstride, c.Cout2 = ly["w2p"].data_ptr(), ly["b2"].data_ptr(), ly["out2"].data_ptr(), 128, 128
        nbytes = int(L.adayolo_conv_chain_workspace_bytes(arr, n))
        assert nbytes > 0
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=DEV)
        _lib.check(L.adayolo_conv_chain_prepare(arr, n, vp(ws.data_ptr()), nbytes), "prepare")

        def run():
            _lib.check(L.adayolo_conv_chain_fwd(arr, n, vp(ws.data_ptr()), nbytes, _lib.stream_ptr()), "chain")
        run.status = lambda: int(L.adayolo_conv_chain_status(vp(ws.data_ptr())))
        run.keep = (arr, ws)
        return run


class _Net512(_Net):
    """The detector's C = 512 stage in small: x -(3x3 s2 c0->512, tile 0)-> y0, then blocks of
    [1x1 512->256 (256 x 128 tile), 3x3 256->512 + shortcut (256 x 256 tile, two n-tiles)], then 3x3 s2 512->1024 on 256 x 128 tiles."""

    def __init__(self, B, H, W, c0=256, blocks=2, seed=0):
        self.B, self.H, self.W = B, H, W
        self.x = _bf(B, H, W, c0, seed=seed)
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        rnd = lambda n, sd: torch.randn(n, generator=torch.Generator().manual_seed(sd)).to(DEV) * 0.1     # noqa: E731
        mk = lambda shape: torch.empty(shape, dtype=torch.bfloat16, device=DEV)                            # noqa: E731
        self.layers = [dict(inp=self.x, w=_bf(512, 3, 3, c0, seed=seed + 1, scale=(9 * c0) ** -0.5), b=rnd(512, seed + 2), res=None,
                            out=mk((B, Ho, Wo, 512)), k=3, s=2, cin=c0, cout=512, tile=0)]
        cur = self.layers[0]["out"]
        for i in range(blocks):
            hid = dict(inp=cur, w=_bf(256, 1, 1, 512, seed=seed + 10 + i, scale=512 ** -0.5), b=rnd(256, seed + 20 + i), res=None,
                       out=mk((B, Ho, Wo, 256)), k=1, s=1, cin=512, cout=256, tile=1)
            blk = dict(inp=hid["out"], w=_bf(512, 3, 3, 256, seed=seed + 30 + i, scale=(9 * 256) ** -0.5), b=rnd(512, seed + 40 + i),
                       res=cur, out=mk((B, Ho, Wo, 512)), k=3, s=1, cin=256, cout=512, tile=0)
            self.layers += [hid, blk]
            cur = blk["out"]
        H2, W2 = (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1
        self.layers.append(dict(inp=cur, w=_bf(1024, 3, 3, 512, seed=seed + 60, scale=(9 * 512) ** -0.5), b=rnd(1024, seed + 61), res=None,
                                out=mk((B, H2, W2, 1024)), k=3, s=2, cin=512, cout=1024, tile=1))


# (B, H, W, first Cin, blocks): one tile per layer; partial last tile + several images per tile; many tiles (two rounds of 256
# CUs, the layers overlap in flight); the benchmark's stage shape
@pytest.mark.parametrize("B,H,W,c0,blocks", [(1, 16, 16, 64, 2), (3, 9, 11, 128, 3), (2, 92, 160, 128, 4), (8, 92, 160, 128, 8)])
def test_chain_is_bit_identical_to_separate_launches(B, H, W, c0, blocks):
    net = _Net(B, H, W, c0, blocks, seed=B * 100 + H)
    net.poison()
    net.run_separately()
    torch.cuda.synchronize()
    want = [t.clone() for t in net.outputs()]
    assert all(torch.isfinite(t.float()).all() for t in want)
    run = net.chain()
    for rep in range(4):                                 # every launch starts from zeroed counters and poisoned outputs
        net.poison()
        run()
        torch.cuda.synchronize()
        assert run.status() == 0
        for i, (got, ref) in enumerate(zip(net.outputs(), want)):
            assert torch.equal(got, ref), (rep, i, (got.float() - ref.float()).abs().max().item())


@pytest.mark.parametrize("B,H,W,blocks", [(1, 20, 24, 1), (3, 30, 22, 2), (8, 92, 160, 3)])
def test_mixed_tile_chain_is_bit_identical_to_separate_launches(B, H, W, blocks):
    """256 x 256 and 256 x 128 tiles in one chain (the C = 512 stage's shape: 1x1 on the narrow tile, 3x3 + shortcut on the wide
    one with two n-tiles per m-tile, stride-2 convs at both ends) against variants 50 / 60 launched one by one."""
    net = _Net512(B, H, W, 256, blocks, seed=B * 10 + H)
    net.poison()
    net.run_separately()
    torch.cuda.synchronize()
    want = [t.clone() for t in net.outputs()]
    assert all(torch.isfinite(t.float()).all() for t in want)
    run = net.chain()
    for rep in range(4):
        net.poison()
        run()
        _check(net, run, want, f"mixed tiles, rep {rep}")
    side = torch.cuda.Stream()
    junk = torch.randn(4096, 4096, device=DEV)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    for rep in range(4):
        net.poison()
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(1 + rep % 3):
                junk = torch.tanh(junk @ junk * 1e-3)
        g.replay()
        _check(net, run, want, f"mixed tiles, replay under load, rep {rep}")


def test_chain_sees_fresh_data_when_the_input_changes():
    """The same buffers, new contents, no poison in between: a CU (or an XCD's L2) that served a line of an intermediate tensor
    during the previous launch must not serve it again — the arrival counter + the L1 invalidate are the only things between a
    consumer and the previous launch's values."""
    net = _Net(4, 46, 80, 128, 4, seed=5)
    run = net.chain()
    for rep in range(4):
        net.x.copy_(_bf(*net.x.shape, seed=100 + rep))
        net.run_separately()
        torch.cuda.synchronize()
        want = [t.clone() for t in net.outputs()]
        net.x.copy_(_bf(*net.x.shape, seed=200 + rep))   # leave OTHER values behind in every intermediate tensor
        net.run_separately()
        net.x.copy_(_bf(*net.x.shape, seed=100 + rep))
        run()
        torch.cuda.synchronize()
        assert run.status() == 0
        for i, (got, ref) in enumerate(zip(net.outputs(), want)):
            assert torch.equal(got, ref), (rep, i)


def test_chain_under_uneven_load():
    """A second stream keeps part of the chip busy with unrelated work of varying length while the chain runs (fewer resident
    workgroups, uneven progress — the conditions under which a placement- or timing-dependent hand-off fails)."""
    net = _Net(8, 92, 160, 128, 5, seed=9)
    net.run_separately()
    torch.cuda.synchronize()
    want = [t.clone() for t in net.outputs()]
    run = net.chain()
    side = torch.cuda.Stream()
    junk = torch.randn(4096, 4096, device=DEV)
    for rep in range(6):
        net.poison()
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(1 + rep % 3):
                junk = torch.tanh(junk @ junk * 1e-3)
        run()
        _check(net, run, want, f"eager under load, rep {rep}")


def test_chain_graph_replay():
    """The chain as a replayed hipGraph: the counter reset is a memset node that must replay, ahead of the kernel, every time;
    then the same under a loaded second stream, alternating with eager launches."""
    net = _Net(8, 92, 160, 128, 5, seed=9)
    net.run_separately()
    torch.cuda.synchronize()
    want = [t.clone() for t in net.outputs()]
    run = net.chain()
    run()
    _check(net, run, want, "eager warm-up")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    for rep in range(4):
        net.poison()
        g.replay()
        _check(net, run, want, f"replay, idle chip, rep {rep}")
    side = torch.cuda.Stream()
    junk = torch.randn(4096, 4096, device=DEV)
    for rep in range(6):
        net.poison()
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(1 + rep % 3):
                junk = torch.tanh(junk @ junk * 1e-3)
        if rep % 2:
            g.replay()
        else:
            run()
        _check(net, run, want, f"{'replay' if rep % 2 else 'eager'} under load, rep {rep}")


def test_chain_in_a_graph_is_ordered_behind_its_producers():
    """[copy a new input into the chain's input tensor; the chain] captured into ONE graph: every replay must see the input the
    copy node of THAT replay wrote (round 5: with the counter reset as a memset node in front of the kernel, the captured chain
    lost its dependency on the nodes before it and ran beside them — results of the PREVIOUS replay's input)."""
    net = _Net(4, 46, 80, 128, 3, seed=21)
    src = torch.empty_like(net.x)
    run = net.chain()
    src.copy_(_bf(*net.x.shape, seed=300))
    net.x.copy_(src)
    run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        net.x.copy_(src)
        run()
    for rep in range(5):
        src.copy_(_bf(*net.x.shape, seed=301 + rep))
        net.x.copy_(src)
        net.run_separately()
        torch.cuda.synchronize()
        want = [t.clone() for t in net.outputs()]
        net.poison()
        net.x.fill_(0)
        g.replay()
        _check(net, run, want, f"graph [copy, chain], rep {rep}")


def test_a_wait_that_can_never_be_satisfied_gives_up_and_reports():
    """Every spin in the chain is bounded: with one item's dependency made unsatisfiable (its `need` poked to 32767 in the
    device-side table) the launch still ENDS (~1 s), `adayolo_conv_chain_status` names the item, the launch after it waits
    normally again, and after a fresh `prepare` the chain is clean and correct."""
    import numpy as np
    from adaptiveisp_amd.yolo import _lib
    L = _lib.load()
    net = _Net(2, 40, 48, 128, 2, seed=31)
    net.run_separately()
    torch.cuda.synchronize()
    want = [t.clone() for t in net.outputs()]
    run = net.chain()
    arr, ws = run.keep
    n = len(net.layers)
    img = np.zeros(ws.numel(), np.uint8)
    info = (ctypes.c_int32 * 6)()
    assert L.adayolo_conv_chain_tables(arr, n, img.ctypes.data_as(ctypes.c_void_p), img.size, info) == 0
    total, ndone, off_layers, off_items, off_deps, _ = list(info)
    deps = img[off_deps:off_deps + 16 * total].view(np.int32).reshape(total, 4)
    victim = int(np.nonzero(deps[:, 1] >> 16)[0][3])                 # some item that waits for an input window
    wsi = ws.view(torch.int32)
    word = off_deps // 4 + 4 * victim + 1
    good = int(wsi[word])
    wsi[word] = (good & ~0xFFFF) | 0x7FFF
    net.poison()
    run()
    torch.cuda.synchronize()                                         # must return: the spin is bounded
    first = run.status()
    assert first >= victim + 1                                       # the victim (or, in a photo finish, an item waiting for IT)
    # the production path's check: the pinned host word, read without any device call (the give-up must not be
    # visible through a blocking test hook only)
    assert int(L.adayolo_conv_chain_poll(ctypes.c_void_p(ws.data_ptr()))) == first
    wsi[word] = good                                                 # the table is whole again: this launch waits normally
    net.poison()
    run()
    torch.cuda.synchronize()
    for got, ref in zip(net.outputs(), want):
        assert torch.equal(got, ref)
    assert run.status() == first                                     # (sticky until the next prepare)
    assert int(L.adayolo_conv_chain_poll(ctypes.c_void_p(ws.data_ptr()))) == first
    _lib.check(L.adayolo_conv_chain_prepare(arr, n, ctypes.c_void_p(ws.data_ptr()), ws.numel()), "prepare")
    assert int(L.adayolo_conv_chain_poll(ctypes.c_void_p(ws.data_ptr()))) == 0
    net.poison()
    run()
    _check(net, run, want, "after a fresh prepare")
    assert int(L.adayolo_conv_chain_poll(ctypes.c_void_p(ws.data_ptr()))) == 0


def test_engine_raises_when_a_chain_wait_gave_up(monkeypatch):
    """YoloEngine polls the host word at the top of every forward and offers the blocking check for sync points: a forward whose
    chain gave up must not go unnoticed (rc 0 + wrong detections)."""
    import os
    import numpy as np
    from adaptiveisp_amd.yolo import YoloEngine, _lib, yolov3
    tune = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "adaptiveisp_amd", "yolo", "tuning", "mi355x.json")
    torch.manual_seed(0)
    monkeypatch.setenv("ADAYOLO_CHAIN", "1")
    eng = YoloEngine(yolov3().eval(), 8, 720, 1280, device=DEV)      # the benchmark's size: the C = 256 stage is a chain by default
    eng.autotune(cache=tune, write=False)
    assert eng.chains
    x = torch.rand(8, 3, 720, 1280, device=DEV)
    eng(x)
    torch.cuda.synchronize()
    eng.check_chains(sync=True)                                      # clean so far
    L = _lib.load()
    ws = eng.chains[0]["ws"]
    img = np.zeros(ws.numel(), np.uint8)
    info = (ctypes.c_int32 * 6)()
    kind, fn, args = next(p for p in eng.plan if p[0] == "chain" and p[2][2].value == ws.data_ptr())
    assert L.adayolo_conv_chain_tables(args[0], args[1], img.ctypes.data_as(ctypes.c_void_p), img.size, info) == 0
    total, _, _, _, off_deps, _ = list(info)
    deps = img[off_deps:off_deps + 16 * total].view(np.int32).reshape(total, 4)
    victim = int(np.nonzero(deps[:, 1] >> 16)[0][0])
    wsi = ws.view(torch.int32)
    word = off_deps // 4 + 4 * victim + 1
    good = int(wsi[word])
    wsi[word] = (good & ~0xFFFF) | 0x7FFF
    eng(x)                                                           # this forward is wrong, and returns normally
    torch.cuda.synchronize()
    wsi[word] = good
    with pytest.raises(_lib.AdayoloError, match="gave up"):
        eng(x)                                                       # ... the next one says so before it launches anything
    with pytest.raises(_lib.AdayoloError, match="gave up"):
        eng.check_chains(sync=True)


def test_chain_fwd_refuses_a_workspace_prepared_for_another_list():
    from adaptiveisp_amd.yolo import _lib
    L = _lib.load()
    net = _Net(2, 40, 48, 128, 2, seed=5)
    run = net.chain()
    arr, ws = run.keep
    n = len(net.layers)
    vp = ctypes.c_void_p
    assert L.adayolo_conv_chain_fwd(arr, n, vp(ws.data_ptr()), ws.numel(), _lib.stream_ptr()) == 0
    assert L.adayolo_conv_chain_fwd(arr, n - 1, vp(ws.data_ptr()), ws.numel(), _lib.stream_ptr()) == -1
    other = torch.empty_like(ws)
    assert L.adayolo_conv_chain_fwd(arr, n, vp(other.data_ptr()), other.numel(), _lib.stream_ptr()) == -1     # never prepared
    torch.cuda.synchronize()


def test_chain_refuses_what_it_does_not_serve():
    from adaptiveisp_amd.yolo import _lib
    L = _lib.load()
    net = _Net(1, 16, 16, 64, 1, tail=False)
    n = len(net.layers)

    def arr_of(mut=None):
        arr = (_lib.ChainLayer * n)()
        for c, ly in zip(arr, net.layers):
            c.in_, c.in_cstride, c.weight, c.bias = ly["inp"].data_ptr(), ly["cin"], ly["w"].data_ptr(), ly["b"].data_ptr()
            c.residual, c.res_cstride = (ly["res"].data_ptr(), 256) if ly["res"] is not None else (None, 0)
            c.out, c.out_cstride = ly["out"].data_ptr(), ly["cout"]
            c.B, c.H, c.W, c.Cin, c.Cout, c.ksize, c.stride, c.act = 1, 16, 16, ly["cin"], ly["cout"], 3, 1, _lib.ACT_SILU
            if "w2" in ly:
                c.weight2, c.bias2, c.out2, c.out2_cstride, c.Cout2 = ly["w2p"].data_ptr(), ly["b2"].data_ptr(), ly["out2"].data_ptr(), 128, 128
        if mut:
            mut(arr)
        return arr
    assert L.adayolo_conv_chain_workspace_bytes(arr_of(), n) > 0
    assert L.adayolo_conv_chain_workspace_bytes(arr_of(), 0) == 0
    # an output that is also the chain's input; a Cin the tile does not serve; an input INSIDE an earlier output (not exactly it)
    assert L.adayolo_conv_chain_workspace_bytes(arr_of(lambda a: setattr(a[1], "out", a[0].in_)), n) == 0
    assert L.adayolo_conv_chain_workspace_bytes(arr_of(lambda a: setattr(a[0], "Cin", 40)), n) == 0
    assert L.adayolo_conv_chain_workspace_bytes(arr_of(lambda a: setattr(a[1], "in_", a[0].out2 + 256)), n) == 0


def test_engine_with_chains_equals_engine_without(monkeypatch):
    """The tuned detector at the benchmark's size: ADAYOLO_CHAIN=1 (default) against ADAYOLO_CHAIN=0 — the prediction is
    bit-identical, the plan holds at least the backbone's C = 256 stage as one launch, eagerly and from a replayed graph."""
    import os
    from _synth import synth_yolo_state_dict, test_image
    from adaptiveisp_amd.yolo import YoloEngine, yolov3
    tune = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "adaptiveisp_amd", "yolo", "tuning", "mi355x.json")
    det = yolov3()
    det.load_state_dict(synth_yolo_state_dict(det, seed=2))
    det = det.eval()
    x = torch.from_numpy(test_image(8, 720, 1280, seed=3, special=False)).to(DEV)
    monkeypatch.setenv("ADAYOLO_CHAIN", "0")
    plain = YoloEngine(det, 8, 720, 1280, device=DEV)
    plain.autotune(cache=tune, write=False)
    assert not plain.chains
    want = plain(x).clone()
    monkeypatch.setenv("ADAYOLO_CHAIN", "1")
    eng = YoloEngine(det, 8, 720, 1280, device=DEV)
    eng.autotune(cache=tune, write=False)
    assert eng.chains and max(c["layers"] for c in eng.chains) >= 8, [c["layers"] for c in eng.chains]
    assert len(eng.plan) < len(plain.plan)
    got = eng(x)
    torch.cuda.synchronize()
    assert eng.chain_status() == 0
    assert torch.equal(got, want)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        eng(x)
    for _ in range(3):
        eng.pred.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        assert eng.chain_status() == 0 and torch.equal(eng.pred, want)


def test_engine_with_chains_equals_engine_without_at_4k(monkeypatch):
    """Config 5's size (4 x 2160 x 3840): every stage of the backbone has more tiles than CUs there, so the engine's chain runs
    through the C = 256, C = 512 and C = 1024 stages with BOTH tile kinds (a mixed chain of two dozen layers) — bit-identical
    to one launch per layer."""
    import os
    from _synth import synth_yolo_state_dict, test_image
    from adaptiveisp_amd.yolo import YoloEngine, yolov3
    tune = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "adaptiveisp_amd", "yolo", "tuning", "mi355x.json")
    det = yolov3()
    det.load_state_dict(synth_yolo_state_dict(det, seed=2))
    det = det.eval()
    x = torch.from_numpy(test_image(4, 2160, 3840, seed=5, special=False)).to(DEV)
    monkeypatch.setenv("ADAYOLO_CHAIN", "0")
    plain = YoloEngine(det, 4, 2160, 3840, device=DEV)
    plain.autotune(cache=tune, write=False)
    want = plain(x).clone()
    del plain
    torch.cuda.empty_cache()
    monkeypatch.setenv("ADAYOLO_CHAIN", "1")
    eng = YoloEngine(det, 4, 2160, 3840, device=DEV)
    eng.autotune(cache=tune, write=False)
    assert eng.chains and max(c["layers"] for c in eng.chains) >= 16, [c["layers"] for c in eng.chains]
    for _ in range(2):
        got = eng(x)
        torch.cuda.synchronize()
        assert eng.chain_status() == 0
        assert torch.equal(got, want)
