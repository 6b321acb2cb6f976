#!/usr/bin/env bash
# Runs ON THE GPU BOX: SQ counters of the persistent chain kernel and of the launches it replaces (the fused-pair kernel), from
# the same process (tools/chain_ab.py runs both), in separate --pmc passes (no trace domains besides kernel-trace).
# Output: gpurun_out/<tag>_chain_pmc.json      usage: gpurun -- 'bash tools/chain_pmc.sh round5'
set -u
TAG=${1:-round5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for GROUP in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  i=$((i+1))
  rocprofv3 --pmc $GROUP --kernel-trace --output-format csv -d "$OUT/${TAG}_chainpmc" -o "p$i" -- python3 "$R/tools/chain_ab.py" --rounds 2 --reps 2 > "$OUT/${TAG}_chainpmc_$i.log" 2>&1
done
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, os, sys, collections
out, tag = sys.argv[1:3]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for p in sorted(glob.glob(os.path.join(out, f"{tag}_chainpmc", "*_counter_collection.csv"))):
    for r in csv.DictReader(open(p)):
        n = r["Kernel_Name"]
        key = "chain::k_conv_chain" if "k_conv_chain" in n else ("pp::k_conv_pp<0, true> grid " + r.get("Grid_Size", "?")) if "k_conv_pp<0, true>" in n else None
        if key:
            res[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
summary = {}
for k, cs in res.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    m["launches_seen"] = max(len(v) for v in cs.values())
    if m.get("SQ_BUSY_CU_CYCLES"):
        m["mfma_busy_over_cu_busy"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / m["SQ_BUSY_CU_CYCLES"]
    if m.get("SQ_LDS_IDX_ACTIVE"):
        m["lds_conflict_frac"] = m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"]
    if m.get("SQ_WAVE_CYCLES"):
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in m:
                m[c + "_frac_of_wave_cycles"] = m[c] / m["SQ_WAVE_CYCLES"]
    summary[k] = m
json.dump(summary, open(os.path.join(out, f"{tag}_chain_pmc.json"), "w"), indent=1)
print(json.dumps(summary, indent=1)[:4000])
PY
