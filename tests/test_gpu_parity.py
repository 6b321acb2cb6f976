"""HIP kernels vs the oracle / the reference's golden vectors, through the C-ABI (libadaisp.so).
Tolerances: integer/selection stages exact; fp32 filters 1e-5 relative (north_star), with a small
absolute floor for values near zero."""
import zlib

import numpy as np
import pytest
import torch

from _margins import close

pytestmark = pytest.mark.gpu

OPS = {"E": 0, "G": 1, "CCM": 2, "Shr": 3, "NLM": 4, "T": 5, "Ct": 6, "Sp": 7, "BW": 8, "W": 9, "USM": 10,
       "ShrV2": 11, "C": 12}
RTOL, ATOL = 1e-5, 2e-6


def dev():
    assert torch.cuda.is_available(), "run with -m gpu on the MI355X box"
    return torch.device("cuda:0")


def gpu_process(op, img, params, clip):
    from adaptiveisp_amd import _lib
    out = _lib.process(op, torch.from_numpy(img).to(dev()), torch.from_numpy(np.asarray(params, np.float32)).to(dev()),
                       clip=clip)
    torch.cuda.synchronize()
    return out.cpu().numpy()


def rand_params(op, B, rng):
    """Regressed-range parameters for op."""
    if op in (OPS["E"],): return rng.uniform(-3.0, 3.0, (B, 1))
    if op == OPS["G"]: return rng.uniform(0.4, 2.8, (B, 1))
    if op == OPS["CCM"]: return rng.uniform(0.2, 2.0, (B, 9)) * np.where(rng.random((B, 9)) < 0.2, -0.3, 1.0)
    if op in (OPS["Shr"], OPS["ShrV2"]): return rng.uniform(0.0, 10.0, (B, 1))
    if op == OPS["NLM"]: return rng.uniform(0.02, 0.9, (B, 1))
    if op == OPS["T"]: return rng.uniform(0.5, 2.0, (B, 8))
    if op == OPS["C"]: return rng.uniform(0.9, 1.1, (B, 24))
    if op == OPS["Ct"]: return rng.uniform(-0.95, 0.95, (B, 1))
    if op in (OPS["Sp"], OPS["BW"]): return rng.uniform(0.02, 0.98, (B, 1))
    if op == OPS["W"]: return rng.uniform(0.6, 1.6, (B, 3))
    if op == OPS["USM"]: return np.stack([rng.uniform(0.3, 2.0, B), rng.uniform(0.0, 2.0, B)], 1)
    raise KeyError(op)


# Stages whose arithmetic is +,-,*,/,min,max,floor,compare only: the kernels are built -ffp-contract=off in the
# reference's operation order, so the HIP result must be BIT-IDENTICAL to the reference's (and to the oracle's).
# The rest go through a transcendental (exp/log/pow/cos/sqrt: device unit vs ATen/libm) or — NLM, USM — a long sum
# whose association differs from ATen's: 1e-5 relative (north_star).
BIT_EXACT = ("CCM", "Shr", "ShrV2", "T", "Sp", "BW", "W")


@pytest.mark.parametrize("name", sorted(OPS))
@pytest.mark.parametrize("mode", ["process", "forward"])
def test_golden_vectors(golden, name, mode):
    """Same inputs as the reference run in the build container -> same outputs."""
    g = golden("filters")
    out = gpu_process(OPS[name], g["img"], g[f"{name}.param"], clip=(mode == "forward"))
    if name in BIT_EXACT:
        np.testing.assert_array_equal(out, g[f"{name}.{mode}"])
    else:
        close(f"golden_vectors:{name}", out, g[f"{name}.{mode}"], rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("tag", ["a", "tiny", "odd"])
def test_golden_nlm_wrap(golden, tag):
    g = golden("nlm")
    out = gpu_process(OPS["NLM"], g[f"{tag}.img"], g[f"{tag}.h"], clip=True)
    close("golden_nlm_wrap#1", out, g[f"{tag}.out"], rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("tag", ["s3p1", "s5p5", "s7p3", "s9p3_out_of_range", "s11p5", "s21p7"])
def test_nlm_general_window_sizes(golden, oracle_mod, tag):
    """NonLocalMeansGray(search, patch) for window sizes other than the ISP's 11 / 5 (adaisp_nlm_general, the class API of
    isp/denoise.py:93-119) against the reference's outputs and the oracle; 11 / 5 goes through the tuned kernel."""
    from adaptiveisp_amd import _lib
    from adaptiveisp_amd.isp.denoise import NonLocalMeansGray
    g = golden("nlm_general")
    search, patch = (int(v) for v in g[f"{tag}.sizes"])
    img, h = torch.from_numpy(g[f"{tag}.img"]).to(dev()), torch.from_numpy(g[f"{tag}.h"]).to(dev())
    with torch.no_grad():
        out = NonLocalMeansGray(search, patch)(img, h.reshape(-1, 1, 1, 1))
    torch.cuda.synchronize()
    close("nlm_general_vs_reference", out.cpu().numpy(), g[f"{tag}.out"], rtol=RTOL, atol=ATOL)
    gen = _lib.nlm_general(img, h, search, patch)                       # the gather kernel itself, also for 11 / 5
    close("nlm_general_vs_oracle", gen.cpu().numpy(), oracle_mod.nlm_general(g[f"{tag}.img"], g[f"{tag}.h"], search, patch),
          rtol=RTOL, atol=ATOL)
    with pytest.raises(ValueError):
        NonLocalMeansGray(4, 3)
    with pytest.raises(_lib.AdaispError):
        _lib.nlm_general(img, h, 5, 2)
    with pytest.raises(_lib.AdaispError):
        _lib.nlm_general(img, h, search, patch, out=img)                 # in place


@pytest.mark.parametrize("tag", ["a", "tiny", "odd"])
def test_nlm_reference_order_kernel(golden, oracle_mod, tag):
    """ADAISP_NLM_EXACT: the 25 patch terms in the reference's running-sum order; and how far the default
    (separable association) is from it."""
    from adaptiveisp_amd import _lib
    g = golden("nlm")
    img, h = torch.from_numpy(g[f"{tag}.img"]).to(dev()), torch.from_numpy(g[f"{tag}.h"]).to(dev())
    exact = _lib.process(OPS["NLM"], img, h, clip=True, nlm_exact=True).cpu().numpy()
    fast = _lib.process(OPS["NLM"], img, h, clip=True).cpu().numpy()
    close("nlm_reference_order_kernel#1", exact, g[f"{tag}.out"], rtol=RTOL, atol=ATOL)
    close("nlm_reference_order_kernel#2", fast, g[f"{tag}.out"], rtol=RTOL, atol=ATOL)
    assert np.abs(fast - exact).max() < 2e-6


@pytest.mark.parametrize("shape", [(2, 64, 128), (1, 37, 53), (3, 5, 7), (1, 33, 260), (2, 96, 200)])
def test_nlm_hand_scheduled_kernel_matches_compiled_form(oracle_mod, shape):
    """The default forward kernel (packed row pairs, v_add_f32_dpp row sums in inline asm; both tile heights) against the compiler-
    scheduled form of the same scheme (ADAISP_NLM_SEP_V1) and the oracle, on ragged sizes; repeated launches must be
    bit-identical (a missing wait state around the inline asm shows up as run-to-run garbage)."""
    from adaptiveisp_amd import _lib
    from _synth import test_image
    B, H, W = shape
    img = torch.from_numpy(test_image(B, H, W, seed=5)).to(dev())
    h = torch.linspace(0.02, 0.3, B, device=dev()).reshape(B, 1)
    fast = _lib.process(OPS["NLM"], img, h, clip=True)
    v1 = _lib.process(OPS["NLM"], img, h, clip=True, nlm_v1=True)
    t32 = _lib.process(OPS["NLM"], img, h, clip=True, nlm_tile32=True)
    for _ in range(5):
        assert torch.equal(_lib.process(OPS["NLM"], img, h, clip=True), fast)
        assert torch.equal(_lib.process(OPS["NLM"], img, h, clip=True, nlm_tile32=True), t32)
    ref = oracle_mod.forward(img.cpu().numpy(), OPS["NLM"], h.cpu().numpy(), clip=True)
    close("nlm_hand_scheduled_kernel_matches_compiled_form#1", fast.cpu().numpy(), ref, rtol=RTOL, atol=ATOL)
    assert (fast - v1).abs().max().item() < 2e-6
    # the 24-row (3 workgroups per CU, default) and the 32-row tile run the same per-pixel instruction sequence
    assert torch.equal(fast, t32)


@pytest.mark.parametrize("tag", ["a", "small", "exact", "hd"])
def test_golden_pool64(golden, tag):
    from adaptiveisp_amd import _lib
    g = golden("pool64")
    out = _lib.pool64(torch.from_numpy(g[f"{tag}.img"]).to(dev())).cpu().numpy()
    # window sums in a fixed (but not ATen's) order, then one division: a few ulp, never more
    close("golden_pool64#1", out, g[f"{tag}.out"], rtol=2e-6, atol=0)


SHAPES = [(2, 64, 128), (1, 37, 53), (3, 5, 7), (1, 3, 3), (1, 33, 260), (2, 96, 64)]


@pytest.mark.parametrize("name", sorted(OPS))
@pytest.mark.parametrize("shape", SHAPES)
def test_vs_oracle_shapes(oracle_mod, name, shape):
    """Vector and scalar paths, ragged tiles, minimum sizes; inputs include out-of-range values."""
    from _synth import test_image
    B, H, W = shape
    rng = np.random.default_rng(zlib.crc32(repr((name, shape)).encode()))      # (hash() of a str is salted per process)
    img = test_image(B, H, W, seed=H * 1000 + W)
    img += rng.normal(0, 0.01, img.shape).astype(np.float32)
    if name not in ("NLM",):
        img[:, :, H // 2] *= 3.0                       # a bright row that leaves [0,1]
    p = rand_params(OPS[name], B, rng).astype(np.float32)
    for clip in (False, True):
        out = gpu_process(OPS[name], img, p, clip)
        ref = oracle_mod.forward(img, OPS[name], p, clip=clip)
        if name in BIT_EXACT:
            np.testing.assert_array_equal(out, ref, err_msg=f"{name} {shape} clip={clip}")
        else:
            close(f"vs_oracle_shapes:{name}", out, ref, rtol=RTOL, atol=ATOL, err_msg=f"{name} {shape} clip={clip}")


def test_mixed_ids_one_call(oracle_mod):
    """adaisp_forward: every op (and the zero image) in one batch, ids on the device, fused pooling."""
    from _synth import test_image
    from adaptiveisp_amd import _lib
    ids = np.array([-1] + list(range(13)), np.int32)
    B = len(ids)
    rng = np.random.default_rng(3)
    img = test_image(B, 72, 100, seed=9)
    params = np.zeros((B, 24), np.float32)
    for b, op in enumerate(ids):
        if op >= 0:
            p = rand_params(int(op), 1, rng)
            params[b, :p.shape[1]] = p[0]
    pooled = torch.empty(B, 3, 64, 64, device=dev())
    out = _lib.forward(torch.from_numpy(img).to(dev()), torch.from_numpy(ids).to(dev()),
                       torch.from_numpy(params).to(dev()), clip=True, pooled=pooled)
    torch.cuda.synchronize()
    ref = oracle_mod.forward(img, ids, params, clip=True)
    close("mixed_ids_one_call#1", out.cpu().numpy(), ref, rtol=RTOL, atol=ATOL)
    assert not out[0].any()
    close("mixed_ids_one_call#2", pooled.cpu().numpy(), oracle_mod.pool64(ref), rtol=RTOL, atol=1e-6)
    # the per-op entry point runs the same kernels: bit-identical to the mixed call
    for b, op in enumerate(ids):
        one = _lib.process(int(op), torch.from_numpy(img[b:b + 1]).to(dev()), torch.from_numpy(params[b:b + 1]).to(dev()),
                           clip=True)
        assert torch.equal(one[0], out[b]), f"op {op}"
    # an id outside enum adaisp_op (the ids are device data: no host-side rejection is possible): zero image, not garbage
    bad = torch.tensor([99, 0, -7], dtype=torch.int32, device=dev())
    x3 = torch.from_numpy(img[:3]).to(dev())
    o3 = torch.full_like(x3, float("nan"))
    _lib.forward(x3, bad, torch.from_numpy(params[:3]).to(dev()), clip=True, out=o3)
    assert not o3[0].any() and not o3[2].any() and torch.isfinite(o3[1]).all()


def test_error_paths():
    from adaptiveisp_amd import _lib
    x = torch.rand(1, 3, 8, 8, device=dev())
    p = torch.ones(1, 9, device=dev())
    with pytest.raises(_lib.AdaispError, match="alias"):
        _lib.process(OPS["Shr"], x, p, out=x)
    with pytest.raises(_lib.AdaispError, match="shape"):
        _lib.process(OPS["USM"], torch.rand(1, 3, 2, 8, device=dev()), p)
    with pytest.raises(_lib.AdaispError, match="unknown op"):
        _lib.process(77, x, p)
    y = _lib.process(OPS["E"], x.clone(), p, out=None)
    z = x.clone()
    _lib.process(OPS["E"], z, p, out=z)                      # pointwise ops may run in place
    assert torch.equal(y, z)


# ---- BASELINE.json sizes: properties that need no oracle run ---------------------------------------------

def _full(B=8, H=720, W=1280, seed=1235):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.rand(B, 3, H, W, generator=g) ** 2.2 * 0.5).to(dev())


def test_fullsize_identities():
    from adaptiveisp_amd import _lib
    x = _full()
    B = x.shape[0]
    one = torch.ones(B, 1, device=dev())
    # identity parameterisations are exact
    assert torch.equal(_lib.process(OPS["W"], x, torch.ones(B, 3, device=dev())), x)
    eye = torch.eye(3, device=dev()).reshape(1, 9).repeat(B, 1)
    assert torch.equal(_lib.process(OPS["CCM"], x, eye * 2.0), x)              # rows are re-normalised
    assert torch.equal(_lib.process(OPS["E"], x, one * 0.0), x)
    close("fullsize_identities#1", _lib.process(OPS["G"], x, one), x.clamp_min(0.001), rtol=1e-6, atol=0)   # gamma 1
    assert torch.equal(_lib.process(OPS["BW"], x, one * 0.0), x)
    assert torch.equal(_lib.process(OPS["Sp"], x, one * 0.0), x)
    assert torch.equal(_lib.process(OPS["Shr"], x, one), x)                      # factor 1 keeps the image
    assert torch.equal(_lib.process(OPS["ShrV2"], x, one * 0.0), x)
    # exposure: +1 EV then -1 EV is the identity up to two roundings of exp()
    y = _lib.process(OPS["E"], _lib.process(OPS["E"], x, one), x.new_full((B, 1), -1.0))
    close("fullsize_identities#2", y, x, rtol=1e-6, atol=0)
    # a flat tone curve is the identity on [0,1] up to rounding
    t = _lib.process(OPS["T"], x, torch.full((B, 8), 1.3, device=dev()))
    close("fullsize_identities#3", t, x, rtol=2e-6, atol=1e-7)
    # clip is idempotent and bounds the output
    c = _lib.process(OPS["E"], x, one * 3.0, clip=True)
    assert float(c.max()) <= 1.0 and float(c.min()) >= 0.0
    assert torch.equal(_lib.process(OPS["W"], c, torch.ones(B, 3, device=dev()), clip=True), c)


def test_fullsize_nlm_properties():
    from adaptiveisp_amd import _lib
    x = _full(B=2)
    # h -> 0: only the zero shift keeps weight 1, the filter returns the (clamped) image exactly
    y = _lib.process(OPS["NLM"], x, torch.zeros(2, 1, device=dev()))
    assert torch.equal(y, x)
    # a constant image is a fixed point for any h, and the output is a convex combination of inputs
    c = torch.full_like(x, 0.375)
    close("fullsize_nlm_properties#1", _lib.process(OPS["NLM"], c, torch.full((2, 1), 0.5, device=dev())), c, rtol=1e-6, atol=0)
    z = _lib.process(OPS["NLM"], x, torch.full((2, 1), 0.3, device=dev()))
    assert float(z.max()) <= float(x.max()) + 1e-6 and float(z.min()) >= float(x.min()) - 1e-6
    # circular boundary: rolling the input rolls the output (torch.roll semantics of the reference)
    xr = torch.roll(x, shifts=(17, -29), dims=(2, 3)).contiguous()
    zr = _lib.process(OPS["NLM"], xr, torch.full((2, 1), 0.3, device=dev()))
    assert torch.equal(zr, torch.roll(z, shifts=(17, -29), dims=(2, 3)))


def test_fullsize_pool_checksum():
    from adaptiveisp_amd import _lib
    x = _full(B=8)
    p = _lib.pool64(x)
    # 1280 = 64*20 columns tile exactly; rows overlap (720/64 = 11.25) -> compare against torch's own pooling
    ref = torch.nn.functional.adaptive_avg_pool2d(x, (64, 64))
    close("fullsize_pool_checksum#1", p, ref, rtol=1e-5, atol=1e-7)


# ---- parameter gradients (adaisp_backward_params) vs the reference's autograd ----------------------------

@pytest.mark.parametrize("name", sorted(OPS))
@pytest.mark.parametrize("mode", ["process", "forward"])
def test_param_gradients_match_reference_autograd(golden, name, mode):
    from adaptiveisp_amd import _lib
    g, gg = golden("filters"), golden("filters_grad")
    img = torch.from_numpy(g["img"]).to(dev())
    p = torch.from_numpy(g[f"{name}.param"]).to(dev())
    ids = torch.full((2,), OPS[name], dtype=torch.int32, device=dev())
    grad = _lib.backward_params(img, torch.from_numpy(gg["grad_out"]).to(dev()), ids, p, clip=(mode == "forward"))
    ref = gg[f"{name}.{mode}"]
    # sums of ~3k signed terms in a different order than autograd's reductions: compare to the gradient scale
    close(f"param_gradients:{name}", grad.cpu().numpy()[:, :ref.shape[1]], ref, rtol=2e-4, atol=2e-4 * max(1.0, np.abs(ref).max()))


def test_autograd_through_filter_modules(golden):
    """Filter.forward on a parameter that requires grad: autograd reaches the heads through the HIP backward."""
    from adaptiveisp_amd.config import cfg
    from adaptiveisp_amd.isp import filters as F
    g, gg = golden("filters"), golden("filters_grad")
    img = torch.from_numpy(g["img"]).to(dev())
    G = torch.from_numpy(gg["grad_out"]).to(dev())
    for name, cls in (("CCM", F.CCMFilter), ("T", F.ToneFilter), ("NLM", F.DenoiseFilter), ("W", F.ImprovedWhiteBalanceFilter)):
        f = cls(cfg, predict=False)
        feat = torch.from_numpy(g[f"{name}.feat"]).to(dev()).requires_grad_(True)
        low, _, _ = f.forward(img, specified_parameter=f.filter_param_regressor(feat))
        (low * G).sum().backward()
        assert feat.grad is not None and torch.isfinite(feat.grad).all() and feat.grad.abs().sum() > 0
    with pytest.raises(NotImplementedError):
        x = img.clone().requires_grad_(True)
        F.ExposureFilter(cfg).forward(x, specified_parameter=torch.zeros(2, 1, device=dev()))[0].sum().backward()


def test_fullsize_4k_properties(oracle_mod):
    """BASELINE config 5 sizes (4 x 2160 x 3840, denoise + sharpen heavy): size-independent properties, plus an exact
    oracle comparison on a crop that contains a tile corner of every kernel."""
    from adaptiveisp_amd import _lib
    torch.manual_seed(5)
    B, H, W = 2, 2160, 3840                                   # two of the four images: 0.8 GB of fp32 per tensor pair
    x = (torch.rand(B, 3, H, W, device=dev()) ** 2.2 * 0.5)
    one = torch.ones(B, 1, device=dev())
    # sharpen with factor 1 is the identity (adjust_sharpness: img*1 + blur*0), NLM of a constant image is that constant
    close("fullsize_4k_properties#1", _lib.process(OPS["Shr"], x, one), x, rtol=0, atol=0)
    c = torch.full((1, 3, H, W), 0.25, device=dev())
    close("fullsize_4k_properties#2", _lib.process(OPS["NLM"], c, torch.full((1, 1), 0.4, device=dev())), c, rtol=1e-6, atol=1e-7)
    # sharpen(f) on the whole frame vs the oracle on a crop whose stencil footprint lies inside the crop
    f = torch.tensor([[2.5], [0.3]], device=dev())
    y = _lib.process(OPS["Shr"], x, f)
    y0, x0, ch, cw = 1000, 2000, 96, 160
    crop = x[:, :, y0 - 1:y0 + ch + 1, x0 - 1:x0 + cw + 1].contiguous().cpu().numpy()
    ref = oracle_mod.forward(crop, oracle_mod.OPS["SHARPEN"], f.cpu().numpy(), clip=True)[:, :, 1:-1, 1:-1]
    close("fullsize_4k_properties#3", y[:, :, y0:y0 + ch, x0:x0 + cw].cpu().numpy(), ref, rtol=1e-5, atol=2e-6)
    assert float(y.min()) >= 0.0 and float(y.max()) <= 1.0
    p = _lib.pool64(x)
    close("fullsize_4k_properties#4", p, torch.nn.functional.adaptive_avg_pool2d(x, 64), rtol=1e-5, atol=1e-6)


# ---- BASELINE.json sizes against the oracle itself ---------------------------------------------------------

SCHEDULES = {"S_mixed": ["E", "CCM", "NLM", "Shr", "T"], "S_point": ["E", "W", "CCM", "T", "G"]}


@pytest.mark.parametrize("sched", sorted(SCHEDULES))
def test_fullsize_episode_vs_oracle(oracle_mod, sched):
    """Config 2 (8 x 3 x 720 x 1280), five RL steps through adaisp_forward with MIXED op ids in every call (image b runs
    the schedule rotated by b, so each launch carries all five kernels' work) and fixed regressed-range parameters.
    Every step is compared with the oracle on all 8 images, the oracle being fed the same step input (rtol 1e-5;
    bit-exact for the BIT_EXACT stages); the fused 64x64 pooling of every step likewise; and the oracle's own chained
    episode (its outputs fed back to it) must agree with the device's final image."""
    from adaptiveisp_amd import _lib
    names = SCHEDULES[sched]
    B, H, W = 8, 720, 1280
    rng = np.random.default_rng(2024)
    x = _full(B, H, W, seed=1235)
    chain = x.cpu().numpy()
    for k in range(5):
        step_names = [names[(k + b) % 5] for b in range(B)]
        ids = np.array([OPS[n] for n in step_names], np.int32)
        params = np.zeros((B, 24), np.float32)
        for b, n in enumerate(step_names):
            p = rand_params(OPS[n], 1, rng)
            if n == "E":
                p = np.clip(p, -1.0, 1.5)                 # keep the episode inside [0,1] so later steps see content
            if n == "NLM":
                p = rng.uniform(0.02, 0.3, (1, 1))
            params[b, :p.shape[1]] = p[0]
        pooled = torch.empty(B, 3, 64, 64, device=dev())
        y = _lib.forward(x, torch.from_numpy(ids).to(dev()), torch.from_numpy(params).to(dev()), clip=True, pooled=pooled)
        torch.cuda.synchronize()
        ref = oracle_mod.forward(x.cpu().numpy(), ids, params, clip=True)
        out = y.cpu().numpy()
        for b, n in enumerate(step_names):
            if n in BIT_EXACT:
                np.testing.assert_array_equal(out[b], ref[b], err_msg=f"step {k} image {b} {n}")
            else:
                close(f"fullsize_episode_vs_oracle:{n}", out[b], ref[b], rtol=RTOL, atol=ATOL, err_msg=f"step {k} image {b} {n}")
        close("fullsize_episode_vs_oracle#2", pooled.cpu().numpy(), oracle_mod.pool64(out), rtol=2e-6, atol=0)
        chain = oracle_mod.forward(chain, ids, params, clip=True)
        x = y
    # end to end: five chained steps on each side. Differences of 1e-5 relative per step pass through sharpen's
    # (1 + 2f) gain and the tone curve's slopes, hence the wider absolute bound; most pixels stay within 1e-5.
    d = np.abs(x.cpu().numpy() - chain)
    assert d.max() < 2e-4, d.max()
    assert np.mean(d > 1e-5 * np.abs(chain) + 2e-6) < 1e-3


def test_4k_denoise_sharpen_vs_oracle(oracle_mod):
    """Config 5 (2160 x 3840, S_heavy = [NLM, Shr]) on one real-content frame, the whole frame against the oracle."""
    from adaptiveisp_amd import _lib
    x = _full(1, 2160, 3840, seed=77)
    h = torch.tensor([[0.12]], device=dev())
    f = torch.tensor([[3.5]], device=dev())
    y = _lib.process(OPS["NLM"], x, h, clip=True)
    z = _lib.process(OPS["Shr"], y, f, clip=True)
    torch.cuda.synchronize()
    ref_y = oracle_mod.forward(x.cpu().numpy(), OPS["NLM"], h.cpu().numpy(), clip=True)
    close("4k_denoise_sharpen_vs_oracle#1", y.cpu().numpy(), ref_y, rtol=RTOL, atol=ATOL)
    ref_z = oracle_mod.forward(y.cpu().numpy(), OPS["Shr"], f.cpu().numpy(), clip=True)
    np.testing.assert_array_equal(z.cpu().numpy(), ref_z)


# ---- the next step's 64x64 pooling out of the filter launch (row a-fuse) --------------------------------------

FUSE_SHAPES = [(2, 720, 1280), (1, 512, 512), (1, 640, 640), (1, 2160, 3840), (2, 100, 72), (1, 96, 200), (3, 64, 64),
               (1, 65, 4100), (2, 40, 56), (1, 70, 102), (1, 360, 1920), (1, 128, 1000), (1, 1080, 1916)]


@pytest.mark.parametrize("shape", FUSE_SHAPES)
def test_fused_pooling_is_bit_identical_to_the_pooling_launch(shape):
    """adaisp_forward / adaisp_forward_uniform with pooled64_next: for every op (device ids and host-known op) the image
    equals adaisp_process's and the pooled planes equal adaisp_pool64 of that image BIT FOR BIT — the policy's selection
    must not depend on which launch produced its input. Shapes: the BASELINE sizes, window-aligned and ragged sizes,
    and sizes the fused cut does not serve (H < 64, W % 4 != 0, pool columns wider than 64 px: the call falls back)."""
    from _synth import test_image
    from adaptiveisp_amd import _lib
    B, H, W = shape
    big = H * W > 2_000_000
    rng = np.random.default_rng(H * 7 + W)
    img = torch.from_numpy(test_image(B, H, W, seed=H + W, special=W >= 16)).to(dev())
    names = ["E", "Shr", "NLM", "T"] if big else sorted(OPS)
    for name in names:
        op = OPS[name]
        p = torch.from_numpy(rand_params(op, B, rng).astype(np.float32)).to(dev())
        if name == "NLM":
            p = p * 0.3
        pad = torch.zeros(B, 24, device=dev())
        pad[:, :p.shape[1]] = p
        want = _lib.process(op, img, pad, clip=True)
        want_pool = _lib.pool64(want)
        ids = torch.full((B,), op, dtype=torch.int32, device=dev())
        for host_op in (None, op):
            pooled = torch.full((B, 3, 64, 64), float("nan"), device=dev())
            out = torch.full_like(img, float("nan"))
            _lib.forward(img, ids, pad, clip=True, pooled=pooled, out=out, host_op=host_op)
            assert torch.equal(out, want), f"{name} {shape} host_op={host_op}: image differs"
            assert torch.equal(pooled, want_pool), f"{name} {shape} host_op={host_op}: pooled planes differ"
    # mixed ids (every kernel family in one call, plus the zero image and an id no family owns)
    if not big:
        all_ops = [-1, 99] + [OPS[n] for n in sorted(OPS)]
        ids_np = np.array([all_ops[(3 * b + H) % len(all_ops)] for b in range(B)], np.int32)
        params = np.zeros((B, 24), np.float32)
        for b, op in enumerate(ids_np):
            if 0 <= op < 13:
                q = rand_params(int(op), 1, rng)
                params[b, :q.shape[1]] = q[0] * (0.3 if op == OPS["NLM"] else 1.0)
        pooled = torch.full((B, 3, 64, 64), float("nan"), device=dev())
        out = _lib.forward(img, torch.from_numpy(ids_np).to(dev()), torch.from_numpy(params).to(dev()), clip=True, pooled=pooled)
        plain = _lib.forward(img, torch.from_numpy(ids_np).to(dev()), torch.from_numpy(params).to(dev()), clip=True)
        assert torch.equal(out, plain) and torch.equal(pooled, _lib.pool64(out))


@pytest.mark.parametrize("name", ["NLM", "USM", "Shr", "ShrV2"])
def test_midsize_golden_stencils(golden, name):
    """The reference's own output at 1 x 3 x 96 x 160 (interior larger than an NLM tile and a conv strip) — image and the
    64 x 64 pooling of it out of the same launch."""
    from adaptiveisp_amd import _lib
    g = golden("midsize")
    img = torch.from_numpy(g["img"]).to(dev())
    p = torch.zeros(1, 24, device=dev())
    q = torch.from_numpy(g[f"{name}.param"]).to(dev()).reshape(1, -1)
    p[:, :q.shape[1]] = q
    pooled = torch.empty(1, 3, 64, 64, device=dev())
    out = _lib.forward(img, None, p, clip=True, pooled=pooled, host_op=OPS[name])
    ref = g[f"{name}.forward"]
    if name in BIT_EXACT:
        np.testing.assert_array_equal(out.cpu().numpy(), ref)
    else:
        close(f"midsize_golden:{name}", out, ref, rtol=RTOL, atol=ATOL)
    close(f"midsize_golden_pooled:{name}", pooled, g[f"{name}.pooled"], rtol=2e-6, atol=1e-7)


def test_midsize_golden_pooling(golden):
    from adaptiveisp_amd import _lib
    g = golden("midsize")
    close("midsize_golden_pool", _lib.pool64(torch.from_numpy(g["pool.img"]).to(dev())), g["pool.out"], rtol=2e-6, atol=0)
