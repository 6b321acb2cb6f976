#!/bin/bash
# rounds of fresh bench processes under several environments, interleaved
cd "${GRAFT_REPO_ROOT:-.}"
run() { env $1 python bench.py --steps 40 --warmup 8 --no-detail --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])"; }
for r in 1 2 3; do for e in "ADAYOLO_CHAIN=1" "ADAYOLO_CHAIN_GRID=248" "ADAYOLO_CHAIN_GRID=240" "ADAYOLO_CHAIN_GRID=224" "ADAYOLO_CHAIN_STAGGER=4000" "ADAYOLO_CHAIN=0"; do echo "round $r [$e] $(run "$e")"; done; done
