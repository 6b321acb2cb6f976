#!/usr/bin/env bash
# Runs ON THE GPU BOX (via gpurun): the official bench line, the rocprofv3 kernel-stats summary of the same
# command, and two separate PMC passes (FETCH_SIZE / WRITE_SIZE) for the HBM traffic of every kernel.
# Usage: gpurun -- 'bash tools/refresh_profiles.sh round1'   -> files land in gpurun_out/<tag>_*
set -u
TAG=${1:-round1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
# 1) tune once (writes the cache into this checkout and a copy into gpurun_out so it can be committed), 2) official line
# (RETUNE=1 re-measures the per-layer kernel table first; by default the committed table is used as it is)
if [ "${RETUNE:-0}" = "1" ]; then
  python3 "$R/bench.py" --retune --no-cpu-baseline --no-detail --no-extras > /dev/null 2>&1
  cp "$R/adaptiveisp_amd/yolo/tuning/mi355x.json" "$OUT/${TAG}_tuning_mi355x.json"
fi
python3 "$R/bench.py" > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_stats" -o bench -- python3 "$R/bench.py" --no-cpu-baseline --no-extras > "$OUT/${TAG}_stats.log" 2>&1
cp "$OUT/${TAG}_stats/bench_kernel_stats.csv" "$OUT/${TAG}_rocprofv3_kernel_stats.csv"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/${TAG}_pmc" -o $C -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-detail --no-graph > "$OUT/${TAG}_pmc_$C.log" 2>&1
done
python3 - "$OUT" "$TAG" <<'PY'
import collections, csv, json, os, sys
out, tag = sys.argv[1:3]
# one record per (kernel, launch geometry): the same kernel is launched at several sizes (a 64x64 policy tensor and a
# full frame are both k_pointwise) and an average over them says nothing about either
agg = collections.defaultdict(lambda: {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    p = os.path.join(out, f"{tag}_pmc", f"{c}_counter_collection.csv")
    for r in csv.DictReader(open(p)):
        if r["Counter_Name"] == c:
            key = (r["Kernel_Name"], r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?"))
            a = agg[key][c]
            a[0] += float(r["Counter_Value"]); a[1] += 1
res = []
for (name, grid, wg), v in agg.items():
    f, nf = v["FETCH_SIZE"]; w, nw = v["WRITE_SIZE"]
    if nf == 0 or nw == 0:
        continue
    # MI355X_MICROARCH.md: FETCH_SIZE counts 16-B/lane streaming reads at half their size on gfx950 -> x2; units KiB
    res.append({"kernel": name, "grid_size": grid, "workgroup_size": wg, "launches": nf, "fetch_kib_raw": f / nf,
                "write_kib": w / nw, "hbm_bytes_per_launch": (2.0 * f / nf + w / nw) * 1024.0})
res.sort(key=lambda r: -r["hbm_bytes_per_launch"] * r["launches"])
json.dump(res, open(os.path.join(out, f"{tag}_pmc_traffic.json"), "w"), indent=1)
print("kernel/geometry records with traffic:", len(res))
PY
head -c 600 "$OUT/${TAG}_bench.json"; echo
