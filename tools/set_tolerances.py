#!/usr/bin/env python3
"""tests/parity_tolerances.json from the errors measured on the MI355X (profiles/round3_parity_margins*.txt, each written by
a `pytest -m gpu` session: tests/_margins.py). Per label, over the measured runs:
  deterministic labels (the HIP kernels against goldens / the oracle: every run measures the same error)
      rtol = r4 = min(cap, 4 x the relative error on elements with |ref| >= 1e-3)
      atol = min(cap, 4 x need_atol)      (what the absolute term had to cover with that rtol)
  labels whose error differs from run to run (they pass through MIOpen / rocBLAS in the PyTorch head, critic and autograd
  paths, whose reductions are not run-to-run reproducible)
      rtol = min(cap, 16 x the LARGEST relative error of any run), atol = min(cap, 16 x the LARGEST absolute error of any
      run) — their error is a draw from the libraries' reduction order (atomics in the backward kernels) and was seen to vary
      up to 9x between runs (critic_to_actor_gradient:f9: 2.6e-9, 4.2e-9, 2.3e-8), so 4x one sample is not a bound; 16x the
      worst of three (and never below 1/50 of the cap) still leaves these assertions 4 - 50x tighter than in round 2
both rounded UP to two significant digits; floors of 2.4e-7 (two fp32 ulp) on rtol and 1e-9 on atol so that an exact match
on one box does not make a one-ulp difference on the next a failure. A label whose cap is 0 stays exact."""
import glob
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "profiles", "round[34]_parity_margins*.txt")))
# labels that keep their call-site cap: a comparison of two launch PLANS that happens to be bit-identical with the committed
# tuning table (the Bottleneck kernel and the fused pair round h the same way) but need not be with another choice of kernels
KEEP_CAP = {"yolo.engine_bneck_vs_default_plan"}


def up(v, digits=2):
    if v <= 0:
        return 0.0
    e = math.floor(math.log10(v)) - (digits - 1)
    return round(math.ceil(v / 10 ** e) * 10 ** e, 12)


runs = {}
for path in files:
    for ln in open(path):
        if ln.startswith("#") or "|" not in ln:
            continue
        f = [x.strip() for x in ln.split("|")]
        runs.setdefault(f[0], []).append(dict(max_abs=float(f[2]), max_rel=float(f[3]), r4=float(f[4]), need=float(f[5]),
                                              cap_r=float(f[9]), cap_a=float(f[10])))
def family(label):
    return label.split(":")[0].split("#")[0]


# one member of a test's labels varying marks the whole test (same library path)
loose = {family(k) for k, rs in runs.items() if len({(r["max_abs"], r["max_rel"]) for r in rs}) > 1}
table, nondet = {}, []
for label, rs in sorted(runs.items()):
    if label in KEEP_CAP:
        continue
    cap_r, cap_a = max(r["cap_r"] for r in rs), max(r["cap_a"] for r in rs)
    varies = family(label) in loose
    if varies:
        nondet.append(label)
        rtol = min(cap_r, up(max(16.0 * max(r["max_rel"] for r in rs), cap_r / 50.0, 2.4e-7))) if cap_r > 0 else 0.0
        atol = min(cap_a, up(max(16.0 * max(r["max_abs"] for r in rs), cap_a / 50.0, 1e-9))) if cap_a > 0 else 0.0
    else:
        rtol = min(cap_r, up(max(max(r["r4"] for r in rs), 2.4e-7))) if cap_r > 0 else 0.0
        atol = min(cap_a, up(max(4.0 * max(r["need"] for r in rs), 1e-9))) if cap_a > 0 else 0.0
    table[label] = {"rtol": rtol, "atol": atol}
json.dump(table, open(os.path.join(ROOT, "tests", "parity_tolerances.json"), "w"), indent=0, sort_keys=True)
print(f"{len(table)} labels from {len(files)} run(s) -> tests/parity_tolerances.json; {len(nondet)} vary from run to run: {nondet}")
