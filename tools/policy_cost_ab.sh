#!/bin/bash
# What do the policy's 30 launches per episode cost the pipelined step? Interleaved fresh processes on one box:
#   A: the benchmark's step      B: the same with the policy's plans cached (BENCH_EXPERIMENT_NO_POLICY=1: the filters, pooling and
#   the detector run, the policy launches do not) — an experiment, not a benchmark line.   usage (GPU box): tools/policy_cost_ab.sh [rounds=4]
N="${1:-4}"
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
runA() { python bench.py --steps 40 --warmup 8 --no-detail --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])"; }
runB() { BENCH_EXPERIMENT_NO_POLICY=1 python bench.py --steps 40 --warmup 8 --no-detail --no-cpu-baseline --no-extras 2>/dev/null | grep EXPERIMENT | sed 's/.*: \([0-9.]*\) ms per step/\1/'; }
for i in $(seq 1 $N); do
  if [ $((i % 2)) -eq 1 ]; then a=$(runA); b=$(runB); else b=$(runB); a=$(runA); fi
  echo "round $i: with the policy $a ms   policy launches cached $b ms"
done
