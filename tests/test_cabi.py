"""The C-ABI library loads and exports every symbol include/adaisp.h declares (no compute: no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ada(?:isp|yolo)_\w+)\s*\(", text)))


def test_libadaisp_exports_header_symbols():
    from adaptiveisp_amd import _lib
    L = _lib.load()
    names = _declared("adaisp.h")
    assert set(_lib.EXPORTS) <= set(names)
    for n in names:
        assert hasattr(L, n), f"libadaisp.so does not export {n}"
    assert L.adaisp_abi_version() == _lib.ABI_VERSION
    assert L.adaisp_strerror(0) == b"ok" and b"alias" in L.adaisp_strerror(-3)
    expect = {-1: 0, 0: 1, 1: 1, 2: 9, 3: 1, 4: 1, 5: 8, 6: 1, 7: 1, 8: 1, 9: 3, 10: 2, 11: 1, 12: 24, 13: -1}
    for op, n in expect.items():
        assert L.adaisp_num_params(op) == n


def test_argument_checks_without_gpu():
    from adaptiveisp_amd import _lib
    L = _lib.load()
    # null pointers / bad sizes are rejected before anything touches a device
    assert L.adaisp_process(0, None, None, None, 1, 1, 8, 8, 0, None) == -1
    assert L.adaisp_forward(None, None, None, None, None, 1, 1, 8, 8, 0, None) == -1
    assert L.adaisp_pool64(None, None, 1, 8, 8, None) == -1
    buf = (ctypes.c_float * 16)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert L.adaisp_process(99, p, p, p, 1, 1, 2, 2, 0, None) == -2          # unknown op
    assert L.adaisp_process(2, p, p, p, 1, 1, 2, 2, 0, None) == -1           # CCM needs 9 params, stride 1
    assert L.adaisp_process(3, p, p, p, 1, 1, 2, 2, 0, None) == -3           # stencil in place
    # the host-known-op form of the RL step: same checks as adaisp_process, and never in place (it may be a stencil step)
    q = ctypes.cast((ctypes.c_float * 16)(), ctypes.c_void_p)
    assert L.adaisp_forward_uniform(0, None, None, None, None, 1, 1, 8, 8, 0, None) == -1
    assert L.adaisp_forward_uniform(99, p, q, None, p, 1, 1, 2, 2, 0, None) == -2
    assert L.adaisp_forward_uniform(2, p, q, None, p, 1, 1, 2, 2, 0, None) == -1
    assert L.adaisp_forward_uniform(0, p, p, None, p, 1, 1, 2, 2, 0, None) == -3
    assert L.adaisp_forward_uniform(0, p, q, None, p, 1, 1, 2, 2, 0, None) == -4          # H, W < 3


def test_fused_pooling_geometry_is_host_side_and_total():
    """The cut of the fused-pooling kernels (isp_internal.h: PoolGeom), restated: every pixel column is owned by exactly one
    strip, every strip's read span covers what it owns and fits 64 lanes x 4 px, for the BASELINE sizes and awkward widths."""
    def win_lo(o, n): return (o * n) // 64
    def win_hi(o, n): return ((o + 1) * n + 63) // 64
    for W in (64, 68, 72, 100, 128, 200, 512, 640, 1000, 1280, 1920, 3840, 4096):
        geom = None
        for strips in range(1, 17):
            cps = (64 + strips - 1) // strips
            if (64 + cps - 1) // cps != strips:
                continue
            c0 = lambda s: min(s * cps, 64)                                            # noqa: E731
            lo = lambda s: W if c0(s) >= 64 else win_lo(c0(s), W) & ~3                  # noqa: E731
            end = lambda s: (win_hi(c0(s + 1) - 1, W) + 3) & ~3                         # noqa: E731
            if all(end(s) - lo(s) <= 256 for s in range(strips)):
                geom = (cps, strips, lo, end)
                break
        assert geom is not None, W
        cps, strips, lo, end = geom
        owned = []
        for s in range(strips):
            assert lo(s) % 4 == 0 and end(s) % 4 == 0 and lo(s) < end(s) <= W and end(s) >= lo(s + 1)
            for ox in range(min(s * cps, 64), min((s + 1) * cps, 64)):                 # its pool columns lie inside its span
                assert lo(s) <= win_lo(ox, W) and win_hi(ox, W) <= end(s)
            owned += list(range(lo(s), lo(s + 1)))
        assert owned == list(range(W)), W


def test_libadayolo_exports_header_symbols():
    from adaptiveisp_amd.yolo import _lib
    L = _lib.load()
    names = _declared("adayolo.h")
    assert set(_lib.EXPORTS) <= set(names)
    for n in names:
        assert hasattr(L, n), f"libadayolo.so does not export {n}"
    assert L.adayolo_abi_version() == _lib.ABI_VERSION
    assert not hasattr(L, "adayolo_debug_stamps")            # measurement builds are not in the shipped library


def test_adayolo_argument_checks_without_gpu():
    """Rejected before anything touches a device: null pointers, bad channel counts, kernel numbers that are not in the
    library (only the variants the tuning table may name exist: include/adayolo.h)."""
    from adaptiveisp_amd.yolo import _lib
    L = _lib.load()
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    conv = L.adayolo_conv_fwd_variant
    assert conv(None, 8, p, p, None, 0, p, 8, 1, 4, 4, 8, 8, 3, 1, 1, 0, None) == -1
    assert conv(p, 8, p, p, None, 0, p, 8, 1, 4, 4, 7, 8, 3, 1, 1, 0, None) == -2            # Cin % 8
    assert conv(p, 8, p, p, None, 0, p, 8, 1, 4, 4, 8, 8, 5, 1, 1, 0, None) == -2            # ksize 5
    for bogus in (1, 3, 13, 33, 41, 57, 66, 99):
        assert conv(p, 8, p, p, None, 0, p, 8, 1, 4, 4, 8, 8, 3, 1, 1, bogus, None) == -1, bogus
    assert L.adayolo_letterbox_pack(None, p, 8, 1, 4, 4, 4, 0, 0.5, None) == -1
    assert L.adayolo_letterbox_pack(p, p, 6, 1, 4, 4, 4, 0, 0.5, None) == -2


def test_detloss_argument_checks_without_gpu():
    """adayolo_detloss_fwd / _bwd (include/adayolo.h) refuse malformed argument blocks before any launch, and the
    ctypes mirror of the structs has the header's layout (88 / 424 bytes on LP64)."""
    from adaptiveisp_amd.yolo import _lib
    L = _lib.load()
    assert ctypes.sizeof(_lib.LossLayer) == 88 and ctypes.sizeof(_lib.LossArgs) == 424
    assert L.adayolo_detloss_fwd(None, None) == -1
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p).value
    a = _lib.LossArgs()
    a.nl, a.B, a.na, a.nc, a.no, a.loss, a.ticket = 1, 1, 3, 80, 85, p, p
    lay = a.layer[0]
    lay.raw, lay.cs, lay.ny, lay.nx, lay.tobj, lay.cnt, lay.part, lay.n = p, 256, 2, 2, p, p, p, 0
    lay.cs = 248                                               # narrower than na * no
    assert L.adayolo_detloss_fwd(ctypes.byref(a), None) == -2
    lay.cs = 256
    a.no = 84                                                  # no != nc + 5
    assert L.adayolo_detloss_fwd(ctypes.byref(a), None) == -1
    a.no = 85
    lay.n = 3                                                  # matches announced, no arrays
    assert L.adayolo_detloss_fwd(ctypes.byref(a), None) == -1
    lay.n = 0
    a.nc, a.no = 200, 205                                      # more than two classes per lane
    assert L.adayolo_detloss_fwd(ctypes.byref(a), None) == -2
    a.nc, a.no = 80, 85
    assert L.adayolo_detloss_bwd(ctypes.byref(a), None) == -1  # no upstream gradient
    a.grad_loss = p
    assert L.adayolo_detloss_bwd(ctypes.byref(a), None) == -2  # no gradient map
    lay.grad, lay.grad_cs = p, 252                             # not a multiple of 8
    assert L.adayolo_detloss_bwd(ctypes.byref(a), None) == -2


def test_chain_tables_without_gpu():
    """adayolo_conv_chain_tables (the workspace image adayolo_conv_chain_prepare uploads) for a 4-layer chain with fake device
    addresses: layer-major item order, one arrival counter per (layer, m-tile), and for EVERY tile the dependency window checked
    against a brute-force model — the set of producer m-tiles that hold a pixel one of the tile's taps reads (3x3 halo across
    image rows and image boundaries, stride 2 at the end) must be inside [in_lo, in_lo + in_n), and the residual rows' tile too."""
    import ctypes
    import numpy as np
    from adaptiveisp_amd.yolo import _lib
    L = _lib.load()
    B, H, W = 3, 21, 13                                       # M = 819: 4 m-tiles, images straddle tiles
    base = 0x10000000
    MB = 1 << 24
    x, w, b = base, base + 1 * MB, base + 2 * MB
    x0, h0, x1, h1, x2, y = (base + k * MB for k in range(3, 9))
    w2 = base + 10 * MB

    def layer(inp, cin, res, out, cout, s, out2=None, Hi=H, Wi=W):
        c = _lib.ChainLayer()
        c.in_, c.in_cstride, c.weight, c.bias, c.residual, c.res_cstride = inp, cin, w, b, res, 256 if res else 0
        c.out, c.out_cstride, c.B, c.H, c.W, c.Cin, c.Cout, c.ksize, c.stride, c.act = out, cout, B, Hi, Wi, cin, cout, 3, s, 1
        if out2:
            c.weight2, c.bias2, c.out2, c.out2_cstride, c.Cout2 = w2, b, out2, 128, 128
        return c
    specs = [layer(x, 64, None, x0, 256, 1, h0), layer(h0, 128, x0, x1, 256, 1, h1), layer(h1, 128, x1, x2, 256, 1),
             layer(x2, 256, None, y, 512, 2)]
    arr = (_lib.ChainLayer * 4)(*specs)
    nbytes = L.adayolo_conv_chain_workspace_bytes(arr, 4)
    assert nbytes > 0
    img = np.zeros(nbytes, np.uint8)
    info = (ctypes.c_int32 * 6)()
    assert L.adayolo_conv_chain_tables(arr, 4, img.ctypes.data_as(ctypes.c_void_p), nbytes, info) == 0
    total, ndone, off_layers, off_items, off_deps, _ = list(info)
    M = B * H * W
    mt = (M + 255) // 256
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    M3 = B * Ho * Wo
    mt3 = (M3 + 255) // 256
    assert total == 3 * mt + mt3 * 2 and ndone == 3 * mt + mt3
    items = img[off_items:off_items + 16 * total].view(np.int32).reshape(total, 4)
    deps = img[off_deps:off_deps + 16 * total].view(np.int32).reshape(total, 4)
    assert (img[:64 + 4 * ndone] == 0).all()
    # layer-major order, tiles ascending; counters: one per (layer, m-tile)
    want_items = [(l, t, l * mt + t) for l in range(3) for t in range(mt)] + [(3, t, 3 * mt + t // 2) for t in range(mt3 * 2)]
    assert [tuple(r[:3]) for r in items.tolist()] == want_items
    src_in, src_res = [None, 0, 1, 2], [None, 0, 1, None]
    for (l, lid, flag, _), (in_lo, inn, res_lo, rnn) in zip(items.tolist(), deps.tolist()):
        s, (Hq, Wq, Mq) = (2, (Ho, Wo, M3)) if l == 3 else (1, (H, W, M))
        m_tile = lid // (2 if l == 3 else 1)
        need = set()
        for m in range(m_tile * 256, min(m_tile * 256 + 256, Mq)):
            bb, rem = divmod(m, Hq * Wq)
            ho, wo = divmod(rem, Wq)
            for kh in range(3):
                for kw in range(3):
                    hi, wi = ho * s - 1 + kh, wo * s - 1 + kw
                    if 0 <= hi < H and 0 <= wi < W:
                        need.add(((bb * H + hi) * W + wi) // 256)
        if src_in[l] is None:
            assert inn == 0
        else:
            n_in, need_in = inn >> 16, inn & 0xFFFF
            lo = in_lo - src_in[l] * mt
            assert need_in == 1 and 1 <= n_in <= 32 and 0 <= lo and lo + n_in <= mt
            assert need <= set(range(lo, lo + n_in)), (l, lid, sorted(need), lo, n_in)
            assert n_in <= len(need) + 2                     # a window, not "everything"
        if src_res[l] is None:
            assert rnn == 0
        else:
            assert res_lo == src_res[l] * mt + m_tile and rnn == (1 << 16) | 1
    # an item only ever waits for counters of EARLIER layers (deadlock freedom of the hand-out order)
    for (l, _, flag, _), (in_lo, inn, res_lo, rnn) in zip(items.tolist(), deps.tolist()):
        if inn:
            assert in_lo + (inn >> 16) <= l * mt
        if rnn:
            assert res_lo < l * mt
    # argument checks
    assert L.adayolo_conv_chain_fwd(arr, 4, None, 0, None) == -1
    assert L.adayolo_conv_chain_prepare(arr, 4, None, 0) == -1
    bad = (_lib.ChainLayer * 4)(*specs)
    bad[2].out = x0                                           # writes what layer 1 reads
    assert L.adayolo_conv_chain_workspace_bytes(bad, 4) == 0


def test_chain_tables_mixed_tiles_without_gpu():
    """A chain that mixes the two tile kinds, as the detector's C = 512 stage does: 3x3 128 -> 512 on 256 x 256 tiles (two n-tiles
    per m-tile), Bottleneck.cv1 1x1 512 -> 256 on 256 x 128 tiles (two n-tiles), Bottleneck.cv2 3x3 256 -> 512 + shortcut on
    256 x 256 tiles. Counters are per (layer, m-tile) and count n-tiles: a consumer needs ALL n-tiles of every producer m-tile
    in its window (need = the producer's n-tile count); a 1x1 layer's window is its own pixels."""
    import ctypes
    import numpy as np
    from adaptiveisp_amd.yolo import _lib
    L = _lib.load()
    B, H, W = 2, 19, 27                                       # M = 1026: 5 m-tiles, the last one 2 pixels
    base, MB = 0x20000000, 1 << 24
    x, w, b, y0, h, y1 = (base + k * MB for k in range(6))

    def layer(inp, cin, res, out, cout, k, tile):
        c = _lib.ChainLayer()
        c.in_, c.in_cstride, c.weight, c.bias, c.residual, c.res_cstride = inp, cin, w, b, res, cout if res else 0
        c.out, c.out_cstride, c.B, c.H, c.W, c.Cin, c.Cout, c.ksize, c.stride, c.act, c.tile = out, cout, B, H, W, cin, cout, k, 1, 1, tile
        return c
    specs = [layer(x, 128, None, y0, 512, 3, 0), layer(y0, 512, None, h, 256, 1, 1), layer(h, 256, y0, y1, 512, 3, 0)]
    arr = (_lib.ChainLayer * 3)(*specs)
    nbytes = L.adayolo_conv_chain_workspace_bytes(arr, 3)
    assert nbytes > 0
    img = np.zeros(nbytes, np.uint8)
    info = (ctypes.c_int32 * 6)()
    assert L.adayolo_conv_chain_tables(arr, 3, img.ctypes.data_as(ctypes.c_void_p), nbytes, info) == 0
    total, ndone, off_layers, off_items, off_deps, _ = list(info)
    mt = (B * H * W + 255) // 256
    assert (total, ndone) == (mt * 2 * 3, mt * 3)
    items = img[off_items:off_items + 16 * total].view(np.int32).reshape(total, 4).tolist()
    deps = img[off_deps:off_deps + 16 * total].view(np.int32).reshape(total, 4).tolist()
    assert [tuple(r[:3]) for r in items] == [(l, t, l * mt + t // 2) for l in range(3) for t in range(2 * mt)]
    for (l, lid, flag, _), (in_lo, inn, res_lo, rnn) in zip(items, deps):
        m_tile = lid // 2
        if l == 0:
            assert inn == 0 and rnn == 0
            continue
        n_in, need_in = inn >> 16, inn & 0xFFFF
        assert need_in == 2                                   # both n-tiles of every producer m-tile
        lo = in_lo - (l - 1) * mt
        if l == 1:                                            # 1x1: exactly its own m-tile
            assert (lo, n_in) == (m_tile, 1) and rnn == 0
        else:
            need = set()
            for m in range(m_tile * 256, min(m_tile * 256 + 256, B * H * W)):
                bb, rem = divmod(m, H * W)
                ho, wo = divmod(rem, W)
                need |= {((bb * H + hi) * W + wi) // 256 for hi in range(ho - 1, ho + 2) for wi in range(wo - 1, wo + 2)
                         if 0 <= hi < H and 0 <= wi < W}
            assert need <= set(range(lo, lo + n_in)) and n_in <= len(need) + 2
            assert (res_lo, rnn) == (m_tile, (1 << 16) | 2)    # the shortcut: layer 0's m-tile, both its n-tiles
    # a fused layer must be on the 256 x 256 tile; the 256 x 128 tile takes Cout % 128
    bad = (_lib.ChainLayer * 3)(*specs)
    bad[1].tile = 0                                           # Cout 256 on 256 x 256 tiles is fine ...
    assert L.adayolo_conv_chain_workspace_bytes(bad, 3) > 0
    bad[1].tile = 2
    assert L.adayolo_conv_chain_workspace_bytes(bad, 3) == 0
    bad[1].tile, bad[1].Cout, bad[1].out_cstride = 1, 192, 192
    assert L.adayolo_conv_chain_workspace_bytes(bad, 3) == 0
