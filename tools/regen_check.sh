#!/bin/bash
# Regenerates every golden fixture from /root/reference into a scratch directory and fails on ANY array difference against
# the committed files under tests/golden/ (build container only: the reference does not travel to the GPU box).
#   tools/regen_check.sh            # all generators
#   tools/regen_check.sh td,yolo    # some of them
set -e
cd "$(dirname "$0")/.."
OUT=$(mktemp -d /tmp/adaisp_regen.XXXXXX)
cp tests/golden/state_dict_keys.json "$OUT"/ 2>/dev/null || true
if [ -n "$1" ]; then python tests/golden/gen_golden.py --only "$1" --out "$OUT"; else python tests/golden/gen_golden.py --out "$OUT"; fi
python - "$OUT" <<'PY'
import json, os, sys
import numpy as np
out, here = sys.argv[1], "tests/golden"
bad = n = 0
for f in sorted(os.listdir(out)):
    if f.endswith(".npz"):
        a, b = np.load(os.path.join(out, f)), np.load(os.path.join(here, f))
        diff = [k for k in set(a.files) | set(b.files)
                if k not in a.files or k not in b.files or a[k].shape != b[k].shape or not np.array_equal(a[k], b[k], equal_nan=True)]
        n += 1
        print(f"{f}: {len(a.files)} arrays, {len(diff)} differing {diff[:5] if diff else ''}")
        bad += len(diff)
    elif f.endswith(".json"):
        same = json.load(open(os.path.join(out, f))) == json.load(open(os.path.join(here, f)))
        print(f"{f}: {'identical' if same else 'DIFFERENT'}")
        bad += 0 if same else 1
print(f"{n} fixture files regenerated, {bad} differences")
sys.exit(1 if bad else 0)
PY
rm -rf "$OUT"
