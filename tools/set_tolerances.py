#!/usr/bin/env python3
"""tests/parity_tolerances.json from the errors measured on the MI355X (profiles/round3_parity_margins.txt, written by a
`pytest -m gpu` session: tests/_margins.py). Per label: rtol = r4 (= min(cap, 4 x the measured relative error on elements
with |ref| >= 1e-3), atol = min(cap, 4 x need_atol) where need_atol is what the absolute term had to cover with that rtol;
both rounded UP to two significant digits; floors of 2.4e-7 (two fp32 ulp) on rtol and 1e-9 on atol so that an exact
match on one box does not make a one-ulp difference on the next a failure. A label whose cap is 0 stays exact."""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "round3_parity_margins.txt")


def up(v, digits=2):
    if v <= 0:
        return 0.0
    e = math.floor(math.log10(v)) - (digits - 1)
    return round(math.ceil(v / 10 ** e) * 10 ** e, 12)


table = {}
for ln in open(src):
    if ln.startswith("#") or "|" not in ln:
        continue
    f = [x.strip() for x in ln.split("|")]
    label, r4, need, cap_r, cap_a = f[0], float(f[4]), float(f[5]), float(f[9]), float(f[10])
    rtol = min(cap_r, up(max(r4, 2.4e-7))) if cap_r > 0 else 0.0
    atol = min(cap_a, up(max(4.0 * need, 1e-9))) if cap_a > 0 else 0.0
    table[label] = {"rtol": rtol, "atol": atol}
json.dump(table, open(os.path.join(ROOT, "tests", "parity_tolerances.json"), "w"), indent=0, sort_keys=True)
print(f"{len(table)} labels -> tests/parity_tolerances.json")
