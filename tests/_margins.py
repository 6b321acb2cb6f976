"""Float assertions of the GPU parity tests, with the measured error kept.

`close(label, got, ref, rtol, atol)` asserts |got - ref| <= atol + rtol * |ref| element-wise (numpy's allclose rule) and
records, per label, the largest absolute error, the largest relative error on elements with |ref| >= 1e-3, and the
largest share of the allowed error any element used. At the end of a GPU session `tests/conftest.py` writes the table
to `gpurun_out/parity_margins.txt`; the copy under `profiles/` is what the tolerances in the tests were set from
(each <= 4x the measured value). Test infrastructure only."""
import numpy as np

RECORDS = {}


def _np(a):
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return np.asarray(a, dtype=np.float64)


def close(label, got, ref, rtol, atol=0.0, err_msg=""):
    g, r = _np(got), _np(ref)
    assert g.shape == r.shape, f"{label}: shape {g.shape} vs {r.shape}"
    if g.size == 0:
        return
    assert np.isfinite(g).all() or not np.isfinite(r).all(), f"{label}: non-finite output"
    fin = np.isfinite(r)
    assert np.array_equal(g[~fin], r[~fin], equal_nan=True), f"{label}: non-finite pattern differs"
    d = np.abs(g[fin] - r[fin])
    a = np.abs(r[fin])
    allowed = atol + rtol * a
    with np.errstate(divide="ignore", invalid="ignore"):
        used = np.where(d == 0, 0.0, d / allowed)
    big = a >= 1e-3
    rec = RECORDS.setdefault(label, {"n": 0, "max_abs": 0.0, "max_rel": 0.0, "used": 0.0, "rtol": rtol, "atol": atol})
    rec["n"] += 1
    if d.size:
        rec["max_abs"] = max(rec["max_abs"], float(d.max()))
        if big.any():
            rec["max_rel"] = max(rec["max_rel"], float((d[big] / a[big]).max()))
        rec["used"] = max(rec["used"], float(used.max()))
    rec["rtol"], rec["atol"] = max(rec["rtol"], rtol), max(rec["atol"], atol)
    worst = float(used.max()) if d.size else 0.0
    assert worst <= 1.0, (f"{label}: error uses {worst:.3g}x the tolerance (rtol {rtol:g}, atol {atol:g}); "
                          f"max abs {float(d.max()):.3g} {err_msg}")


def dump(path):
    if not RECORDS:
        return
    lines = ["# label | assertions | max abs err | max rel err (|ref| >= 1e-3) | share of tolerance used | rtol | atol"]
    for k in sorted(RECORDS):
        r = RECORDS[k]
        lines.append(f"{k} | {r['n']} | {r['max_abs']:.3e} | {r['max_rel']:.3e} | {r['used']:.3f} | {r['rtol']:g} | {r['atol']:g}")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
