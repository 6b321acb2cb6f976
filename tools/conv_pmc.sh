#!/usr/bin/env bash
# Runs ON THE GPU BOX: SQ counters of the dominant conv kernel (variant 50) on the three big layer shapes, in separate
# --pmc passes (no trace domains besides kernel-trace). Output: gpurun_out/<tag>_conv_pmc.json
# usage: gpurun -- 'bash tools/conv_pmc.sh round1'
set -u
TAG=${1:-round1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for SHAPE in "92 160 128 256 3 1" "46 80 256 512 3 1" "23 40 512 1024 3 1"; do
  for GROUP in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
    i=$((i+1))
    rocprofv3 --pmc $GROUP --kernel-trace --output-format csv -d "$OUT/${TAG}_convpmc" -o "p$i" -- python3 "$R/tools/conv_one.py" $SHAPE 50 6 > "$OUT/${TAG}_convpmc_$i.log" 2>&1
  done
done
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, os, sys, collections
out, tag = sys.argv[1:3]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for p in sorted(glob.glob(os.path.join(out, f"{tag}_convpmc", "*_counter_collection.csv"))):
    for r in csv.DictReader(open(p)):
        if "k_conv_pp" in r["Kernel_Name"]:
            key = f'grid {r.get("Grid_Size", "?")}'
            res[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
summary = {}
for k, cs in res.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CU_CYCLES" in m and m["SQ_BUSY_CU_CYCLES"]:
        m["mfma_busy_over_cu_busy"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / m["SQ_BUSY_CU_CYCLES"]
    if "SQ_LDS_BANK_CONFLICT" in m and m.get("SQ_LDS_IDX_ACTIVE"):
        m["lds_conflict_frac"] = m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]
    if m.get("SQ_WAVE_CYCLES"):
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in m:
                m[c + "_frac_of_wave_cycles"] = m[c] / m["SQ_WAVE_CYCLES"]
    summary[k] = m
json.dump(summary, open(os.path.join(out, f"{tag}_conv_pmc.json"), "w"), indent=1)
print(json.dumps(summary, indent=1)[:3000])
PY
