#!/usr/bin/env bash
# Runs ON THE GPU BOX: the RL iteration at 8 x 512 x 512 with the round-4 training kernels switched on / off, interleaved in one
# session (tools/train_bench.py; every switch is an environment variable of the package), then the ATen op census and the
# kernel trace of the default configuration. Usage: gpurun -- 'bash tools/train_kernels_ab.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
for r in 1 2 3; do
for m in "1 1 1 1 1 1 1" "1 1 1 1 1 1 0" "1 1 1 1 1 0 0" "1 1 1 1 0 0 0" "0 0 0 0 0 0 0"; do set -- $m
echo "== round $r: trunk kernels $1, critic pair $2, TD kernel $3, policy tail $4, detector pair $5, clip + Adam kernels $6, critic on a second stream $7"
ADAISP_TRUNK_KERNELS=$1 ADAISP_CRITIC_PAIR=$2 ADAISP_TD_KERNEL=$3 ADAISP_POLICY_TAIL_KERNEL=$4 ADAYOLO_TRAIN_PAIR=$5 ADAISP_ADAM_KERNEL=$6 ADAISP_CRITIC_STREAM=$7 TRAIN_BENCH_ONLY=hip \
  python tools/train_bench.py 40 2>&1 | grep -v amdgpu.ids | tail -2
done; done
echo "== ATen ops per iteration (default configuration)"
python tools/train_op_count.py 2>&1 | grep -v amdgpu.ids | tail -9
echo "== kernel trace (default configuration)"
bash tools/train_trace.sh | tail -34
