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
