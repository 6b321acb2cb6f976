#!/bin/bash
# Interleaved A/B of the HEADLINE (bench.py's pipelined step) between two environments on ONE box: N rounds of
#   A: env settings of $1 (e.g. "ADAYOLO_CHAIN=0")    B: env settings of $2 (e.g. "ADAYOLO_CHAIN=1")
# each a fresh `python bench.py --steps 40 --warmup 8 --no-detail --no-cpu-baseline --no-extras` (two processes per round, the
# order alternating), printing ms_per_step of every run and the medians.
# usage (GPU box): tools/headline_ab.sh "ADAYOLO_CHAIN=0" "ADAYOLO_CHAIN=1" [rounds]
A="$1"; B="$2"; N="${3:-4}"
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
run() { env $1 python bench.py --steps 40 --warmup 8 --no-detail --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])"; }
as=(); bs=()
for i in $(seq 1 $N); do
  if [ $((i % 2)) -eq 1 ]; then a=$(run "$A"); b=$(run "$B"); else b=$(run "$B"); a=$(run "$A"); fi
  echo "round $i: A[$A] $a ms   B[$B] $b ms"
  as+=($a); bs+=($b)
done
python - "${as[@]}" -- "${bs[@]}" <<'PY'
import statistics, sys
i = sys.argv.index("--")
a = [float(x) for x in sys.argv[1:i]]; b = [float(x) for x in sys.argv[i + 1:]]
print(f"A median {statistics.median(a):.3f} ms ({8e3 / statistics.median(a):.0f} images/s)   B median {statistics.median(b):.3f} ms "
      f"({8e3 / statistics.median(b):.0f} images/s)   B / A = {statistics.median(b) / statistics.median(a):.4f}")
PY
