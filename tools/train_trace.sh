#!/usr/bin/env bash
# Runs ON THE GPU BOX: kernel trace of the RL training iteration (tools/train_bench.py, HIP detector only) — GPU busy time per
# iteration and the kernels that make it up. Usage: gpurun -- 'bash tools/train_trace.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
TRAIN_BENCH_ONLY=hip rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/train_trace" -o tr -- python3 "$R/tools/train_bench.py" 10 > "$OUT/train_trace.log" 2>&1
tail -2 "$OUT/train_trace.log"
python3 - "$OUT/train_trace/tr_kernel_trace.csv" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steady state: the window between the TD-arithmetic launches (one per iteration, csrc/isp_rl_train.hip) of the 6th-from-last
# and the last iteration
starts = [i for i, r in enumerate(rows) if "k_td_fwd" in r["Kernel_Name"]]
NIT = 6
seg = rows[starts[-1 - NIT]:starts[-1]]
span = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
busy = 0; last_end = 0
agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - max(s, last_end) if e > last_end else 0
    last_end = max(last_end, e)
    k = r["Kernel_Name"].split("(")[0].replace("(anonymous namespace)::", "")[-60:]
    agg[k][0] += 1; agg[k][1] += e - s
print(f"{NIT} iterations: span {span/1e6/NIT:.2f} ms each, GPU busy {busy/1e6/NIT:.2f} ms ({100*busy/span:.0f} %), {len(seg)//NIT} kernels per iteration")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"{t/1e6/NIT:8.3f} ms/it {n/NIT:7.1f} x {t/n/1e3:8.1f} us  {k}")
PY
