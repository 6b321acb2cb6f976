#!/usr/bin/env bash
# Runs ON THE GPU BOX: where the HOST spends an RL training iteration (cProfile over tools/train_bench.py, HIP detector):
# cumulative time per function of the package + the most expensive torch calls. usage: gpurun -- 'bash tools/train_host_profile.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
TRAIN_BENCH_PROFILE=/tmp/train.prof TRAIN_BENCH_ONLY=hip python3 "$R/tools/train_bench.py" 40 > "$OUT/train_host_profile.log" 2>&1
tail -2 "$OUT/train_host_profile.log"
python3 - <<'PY' | tee "$OUT/train_host_profile.txt"
import pstats
p = pstats.Stats("/tmp/train.prof")
p.sort_stats("cumulative")
print("== cumulative, package functions (40 steady-state iterations, cProfile on: ~2x slower than unprofiled)")
p.print_stats("adaptiveisp_amd|train_bench", 45)
p.sort_stats("tottime")
print("== self time, everything")
p.print_stats(35)
PY
