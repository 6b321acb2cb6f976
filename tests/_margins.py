"""Float assertions of the GPU parity tests, with the measured error kept and the tolerance taken from it.

`close(label, got, ref, rtol, atol)` asserts |got - ref| <= atol + rtol * |ref| element-wise (numpy's allclose rule).
The (rtol, atol) at the call site are CAPS (north_star's 1e-5 for the fp32 filters, the gradient-scale bounds for the
gradient checks); the tolerance actually asserted comes from `tests/parity_tolerances.json` when the label is listed
there — written by `tools/set_tolerances.py` from the errors MEASURED on the MI355X (`profiles/round3_parity_margins.txt`),
each entry <= 4x the measured value and never looser than the cap.

Per label a GPU session records: the largest absolute error, the largest relative error on elements with
|ref| >= 1e-3, `r4` = min(cap, 4 x that relative error) and `need_atol` = the largest |d| - r4 * |ref| (what the absolute
term has to cover once the relative term is r4 — the dark pixels and the cancellation-limited stencil outputs), and the
share of the ASSERTED tolerance the worst element used. `tests/conftest.py` writes the table to
`gpurun_out/parity_margins.txt` at the end of the session. Test infrastructure only."""
import json
import os

import numpy as np

RECORDS = {}
NOTES = []          # free-form measured figures (one line each), written below the table
_TABLE_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "parity_tolerances.json")
try:          # PARITY_MEASURE=1: assert the call-site caps only (the run that (re)measures the margins the table is made from)
    TABLE = {} if os.environ.get("PARITY_MEASURE") == "1" else json.load(open(_TABLE_PATH))
except Exception:
    TABLE = {}


def _np(a):
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return np.asarray(a, dtype=np.float64)


def close(label, got, ref, rtol, atol=0.0, err_msg=""):
    g, r = _np(got), _np(ref)
    assert g.shape == r.shape, f"{label}: shape {g.shape} vs {r.shape}"
    if g.size == 0:
        return
    cap_r, cap_a = float(rtol), float(atol)
    if label in TABLE:
        rtol, atol = min(cap_r, float(TABLE[label]["rtol"])), min(cap_a, float(TABLE[label]["atol"]))
    fin = np.isfinite(r)
    assert np.array_equal(g[~fin], r[~fin], equal_nan=True), f"{label}: non-finite pattern differs"
    assert np.isfinite(g[fin]).all(), f"{label}: non-finite output"
    d = np.abs(g[fin] - r[fin])
    a = np.abs(r[fin])
    allowed = atol + rtol * a
    with np.errstate(divide="ignore", invalid="ignore"):
        used = np.where(d == 0, 0.0, d / allowed)
    big = a >= 1e-3
    rec = RECORDS.setdefault(label, {"n": 0, "max_abs": 0.0, "max_rel": 0.0, "r4": 0.0, "need_atol": 0.0, "used": 0.0,
                                     "rtol": 0.0, "atol": 0.0, "cap_r": cap_r, "cap_a": cap_a})
    rec["n"] += 1
    rel = 0.0
    if d.size:
        rec["max_abs"] = max(rec["max_abs"], float(d.max()))
        if big.any():
            rel = float((d[big] / a[big]).max())
            rec["max_rel"] = max(rec["max_rel"], rel)
        r4 = min(cap_r, max(4.0 * rel, 2.4e-7)) if cap_r > 0 else 0.0
        rec["r4"] = max(rec["r4"], r4)
        rec["need_atol"] = max(rec["need_atol"], float(np.maximum(d - r4 * a, 0.0).max()))
        rec["used"] = max(rec["used"], float(used.max()))
    rec["rtol"], rec["atol"] = max(rec["rtol"], rtol), max(rec["atol"], atol)
    rec["cap_r"], rec["cap_a"] = max(rec["cap_r"], cap_r), max(rec["cap_a"], cap_a)
    worst = float(used.max()) if d.size else 0.0
    assert worst <= 1.0, (f"{label}: error uses {worst:.3g}x the tolerance (rtol {rtol:g}, atol {atol:g}); "
                          f"max abs {float(d.max()):.3g} {err_msg}")


def close_scaled(label, got, ref, frac, floor=1.0, err_msg=""):
    """max |got - ref| <= frac * max(floor, max |ref|): the detector's rule (bf16 kernels against an fp32 reference: the error
    is a fraction of the tensor's scale, not of each element). Recorded like `close` on the scale-normalised tensors, so the
    measured share of the bound lands in the same table."""
    g, r = _np(got), _np(ref)
    scale = max(float(floor), float(np.abs(r[np.isfinite(r)]).max()) if r.size else 0.0)
    close(label, g / scale, r / scale, rtol=0.0, atol=float(frac), err_msg=err_msg)


VECTORS = {}


def vector_close(label, got, ref, max_rel, min_cos):
    """Whole-tensor agreement of a bf16 result with its fp32 reference: relative L2 distance and cosine (what a gradient
    through 75 bf16 layers can be held to; element-wise bounds say nothing there). Largest rel / smallest cosine per label are
    written below the table."""
    g, r = _np(got).ravel(), _np(ref).ravel()
    assert g.shape == r.shape and np.isfinite(g).all(), label
    nr = float(np.linalg.norm(r))
    rel = float(np.linalg.norm(g - r)) / max(nr, 1e-30)
    cos = float(g @ r) / max(float(np.linalg.norm(g)) * nr, 1e-30)
    rec = VECTORS.setdefault(label, {"n": 0, "rel": 0.0, "cos": 1.0, "max_rel": max_rel, "min_cos": min_cos})
    rec["n"] += 1
    rec["rel"], rec["cos"] = max(rec["rel"], rel), min(rec["cos"], cos)
    assert rel <= max_rel and cos >= min_cos, f"{label}: rel L2 {rel:.5f} (bound {max_rel}), cosine {cos:.6f} (bound {min_cos})"
    return rel, cos


def dump(path):
    if not RECORDS and not NOTES and not VECTORS:
        return
    lines = ["# label | assertions | max abs err | max rel err (|ref| >= 1e-3) | r4 | need_atol | share of asserted tolerance used "
             "| asserted rtol | asserted atol | cap rtol | cap atol"]
    for k in sorted(RECORDS):
        r = RECORDS[k]
        lines.append(f"{k} | {r['n']} | {r['max_abs']:.3e} | {r['max_rel']:.3e} | {r['r4']:.3e} | {r['need_atol']:.3e} | "
                     f"{r['used']:.3f} | {r['rtol']:g} | {r['atol']:g} | {r['cap_r']:g} | {r['cap_a']:g}")
    lines += [f"# vector {k} | assertions {v['n']} | rel L2 {v['rel']:.5f} (bound {v['max_rel']}) | cosine {v['cos']:.7f} "
              f"(bound {v['min_cos']})" for k, v in sorted(VECTORS.items())]
    lines += ["# " + n for n in NOTES]
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
