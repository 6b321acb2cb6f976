"""Pins the CPU oracle (oracle/isp_oracle.c) against golden vectors produced by the reference itself
(tests/golden/gen_golden.py imports /root/reference in the build container)."""
import numpy as np
import pytest

OPS = {"E": 0, "G": 1, "CCM": 2, "Shr": 3, "NLM": 4, "T": 5, "Ct": 6, "Sp": 7, "BW": 8, "W": 9, "USM": 10,
       "ShrV2": 11, "C": 12}
# filters whose arithmetic is only +,-,*,/,min,max,floor,compare must match the reference bit for bit
BIT_EXACT = {"E", "CCM", "Shr", "T", "Sp", "BW", "W", "ShrV2"}
ABS_TOL = 5e-7      # transcendental filters: libm vs ATen/Sleef differ by an ulp or two


@pytest.mark.parametrize("name", sorted(OPS))
@pytest.mark.parametrize("mode", ["process", "forward"])
def test_filter_matches_reference(golden, oracle_mod, name, mode):
    g = golden("filters")
    out = oracle_mod.forward(g["img"], OPS[name], g[f"{name}.param"], clip=(mode == "forward"))
    ref = g[f"{name}.{mode}"]
    if name in BIT_EXACT:
        assert np.array_equal(out, ref), f"{name}: {np.abs(out - ref).max()}"
    else:
        assert np.abs(out - ref).max() <= ABS_TOL


@pytest.mark.parametrize("tag", ["a", "tiny", "odd"])
def test_nlm_wraparound(golden, oracle_mod, tag):
    g = golden("nlm")
    out = oracle_mod.forward(g[f"{tag}.img"], OPS["NLM"], g[f"{tag}.h"], clip=True)
    assert np.abs(out - g[f"{tag}.out"]).max() <= ABS_TOL


@pytest.mark.parametrize("tag", ["s3p1", "s5p5", "s7p3", "s9p3_out_of_range", "s11p5", "s21p7"])
def test_nlm_general_window_sizes(golden, oracle_mod, tag):
    """oracle_nlm_general against the reference's NonLocalMeansGray(search, patch) for sizes other than the ISP's 11 / 5
    (class default 21 / 7), incl. an image outside [0, 1]: only the luminance is clipped (isp/denoise.py:11-17,93-119)."""
    g = golden("nlm_general")
    search, patch = (int(v) for v in g[f"{tag}.sizes"])
    out = oracle_mod.nlm_general(g[f"{tag}.img"], g[f"{tag}.h"], search, patch)
    assert np.abs(out - g[f"{tag}.out"]).max() <= ABS_TOL
    if tag == "s11p5":                                   # in range: the DenoiseFilter path computes the same thing
        via_op = oracle_mod.forward(g[f"{tag}.img"], OPS["NLM"], g[f"{tag}.h"].reshape(-1, 1), clip=True)
        assert np.array_equal(via_op, out)
    with pytest.raises(ValueError):
        oracle_mod.nlm_general(g[f"{tag}.img"], g[f"{tag}.h"], 4, 3)


@pytest.mark.parametrize("tag", ["a", "small", "exact", "hd"])
def test_pool64_bit_exact(golden, oracle_mod, tag):
    g = golden("pool64")
    assert np.array_equal(oracle_mod.pool64(g[f"{tag}.img"]), g[f"{tag}.out"])


def test_pdf_sample_and_one_hot_bit_exact(golden, oracle_mod):
    g = golden("select")
    sel, ns = oracle_mod.select_and_update(g["pdf"], g["u"], np.zeros((24, 13), np.float32), train=True)
    assert np.array_equal(sel, g["idx"])
    assert sel[0] == -1                                            # u == 0 -> all-zero one-hot
    assert np.array_equal(ns[:, 3:], g["one_hot"].astype(np.float32))


def test_state_update_matches_agent_fixture(golden, oracle_mod):
    g = golden("agent")
    for tag in ("s0", "s1"):
        logits = g[f"{tag}.logits"].astype(np.float64)
        # the selection only needs the ordering of the pdf; eval mode = argmax
        sel, ns = oracle_mod.select_and_update(np.exp(logits - logits.max(1, keepdims=True)).astype(np.float32),
                                               g["z"][:, 0], g[tag], train=False)
        assert np.array_equal(sel, g[f"{tag}.selected"])
        assert np.array_equal(ns, g[f"{tag}.new_states"])


# ---- the torch-CPU op-for-op restatement (oracle/torch_ref.py: what bench.py's cpu_baseline times) -------------

@pytest.mark.parametrize("name", sorted(OPS))
@pytest.mark.parametrize("mode", ["process", "forward"])
def test_torch_restatement_matches_reference(golden, name, mode):
    """Same formulation as the reference (whole-tensor ops, masked HSV, 8-pass curves): same ATen kernels, so the
    result is bit-identical except where a reduction's order is ATen's choice (conv) — there 1e-6."""
    import torch
    from oracle import torch_ref
    g = golden("filters")
    img, p = torch.from_numpy(g["img"]), torch.from_numpy(g[f"{name}.param"])
    out = (torch_ref.process if mode == "process" else torch_ref.forward)(OPS[name], img, p).numpy()
    ref = g[f"{name}.{mode}"]
    if name in ("NLM", "USM", "Shr", "ShrV2"):
        np.testing.assert_allclose(out, ref, rtol=0, atol=1e-6)
    else:
        assert np.array_equal(out, ref), f"{name}: {np.abs(out - ref).max()}"


@pytest.mark.parametrize("tag", ["a", "tiny", "odd"])
def test_torch_restatement_nlm_wraparound(golden, tag):
    import torch
    from oracle import torch_ref
    g = golden("nlm")
    out = torch_ref.forward(torch_ref.NLM, torch.from_numpy(g[f"{tag}.img"]), torch.from_numpy(g[f"{tag}.h"])).numpy()
    np.testing.assert_allclose(out, g[f"{tag}.out"], rtol=0, atol=1e-6)


def test_torch_policy_step_all_filters_vs_selected_only(golden, oracle_mod):
    """The reference's stack + one-hot select (agent.py:103-116,154) equals running only the selected filter per image,
    equals the C oracle's mixed-id call; id -1 (u = 0) gives the zero image in all three."""
    import torch
    from oracle import torch_ref
    g = golden("filters")
    img = torch.from_numpy(np.concatenate([g["img"], g["img"][::-1]], 0).copy())      # 4 images
    names = ["E", "G", "CCM", "Shr", "NLM", "T", "Ct", "Sp", "BW", "W"]
    params = [torch.from_numpy(np.concatenate([g[f"{n}.param"], g[f"{n}.param"][::-1]], 0).copy()) for n in names]
    sel = torch.tensor([5, -1, 2, 7])
    full = torch_ref.policy_step(img, params, sel)
    only = torch_ref.policy_step(img, params, sel, selected_only=True)
    np.testing.assert_allclose(full.numpy(), only.numpy(), rtol=0, atol=1e-7)
    assert not full[1].any()
    packed = np.zeros((4, 24), np.float32)
    for b, j in enumerate(sel.tolist()):
        if j >= 0:
            q = params[j][b].reshape(-1).numpy()
            packed[b, :q.size] = q
    ids = np.array([j if j >= 0 else -1 for j in sel.tolist()], np.int32)
    ref = oracle_mod.forward(img.numpy(), ids, packed, clip=True)
    np.testing.assert_allclose(full.numpy(), ref, rtol=0, atol=5e-7)


# ---- mid-size fixtures (1 x 3 x 96 x 160: an interior larger than an NLM tile / a conv strip) ------------

@pytest.mark.parametrize("name", ["NLM", "USM", "Shr", "ShrV2"])
def test_midsize_stencils_match_reference(golden, oracle_mod, name):
    g = golden("midsize")
    out = oracle_mod.forward(g["img"], OPS[name], g[f"{name}.param"], clip=True)
    ref = g[f"{name}.forward"]
    if name in BIT_EXACT:
        assert np.array_equal(out, ref), f"{name}: {np.abs(out - ref).max()}"
    else:
        assert np.abs(out - ref).max() <= ABS_TOL, np.abs(out - ref).max()
    # and the 64 x 64 pooling of the reference's result, bit for bit (windows of 1-2 rows x 2-3 columns)
    assert np.array_equal(oracle_mod.pool64(ref), g[f"{name}.pooled"])


def test_midsize_pooling_bit_exact(golden, oracle_mod):
    g = golden("midsize")
    assert np.array_equal(oracle_mod.pool64(g["pool.img"]), g["pool.out"])       # 180 x 160: 3-4 row overlapping windows


@pytest.mark.parametrize("name", ["NLM", "USM", "Shr", "ShrV2"])
def test_torch_restatement_midsize(golden, name):
    import torch
    from oracle import torch_ref
    g = golden("midsize")
    out = torch_ref.forward(OPS[name], torch.from_numpy(g["img"]), torch.from_numpy(g[f"{name}.param"])).numpy()
    np.testing.assert_allclose(out, g[f"{name}.forward"], rtol=0, atol=1e-6)
