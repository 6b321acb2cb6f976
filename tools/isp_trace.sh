#!/usr/bin/env bash
# Runs ON THE GPU BOX: kernel trace of the single-stream step (ISP episode, then the detector) — per-kernel durations
# and the gaps between consecutive kernels of the ISP episode. Usage: gpurun -- 'bash tools/isp_trace.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/isp_trace" -o st -- python3 "$R/bench.py" --no-pipeline --steps 10 --warmup 2 --no-cpu-baseline --no-detail > "$OUT/isp_trace.log" 2>&1
python3 - "$OUT/isp_trace/st_kernel_trace.csv" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last full step: find the last k_stem_down, walk back to the previous one
idx = [i for i, r in enumerate(rows) if "k_stem_down" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
seg = rows[a:b]
t0 = int(seg[0]["Start_Timestamp"])
prev_end = None
isp = [r for r in seg if "adaisp" in r["Kernel_Name"]]
print("kernels in the step:", len(seg), " ISP kernels:", len(isp))
tot_d = tot_g = 0
for r in isp:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) if prev_end is not None else 0
    name = r["Kernel_Name"].split("(")[0].split("::")[-1][:28]
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap / 1e3:6.1f}  {name}  grid {r.get('Grid_Size','?')}")
    tot_d += e - s
    if prev_end is not None: tot_g += gap
    prev_end = e
print(f"ISP episode: kernels {tot_d / 1e3:.1f} us, gaps {tot_g / 1e3:.1f} us, span {(prev_end - int(isp[0]['Start_Timestamp'])) / 1e3:.1f} us")
PY
