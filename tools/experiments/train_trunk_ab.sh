for r in 1 2; do
for m in "1 1 1 1" "0 0 0 0"; do set -- $m
echo "== TRUNK_KERNELS=$1 CRITIC_PAIR=$2 TD_KERNEL=$3 POLICY_TAIL=$4"
ADAISP_TRUNK_KERNELS=$1 ADAISP_CRITIC_PAIR=$2 ADAISP_TD_KERNEL=$3 ADAISP_POLICY_TAIL_KERNEL=$4 TRAIN_BENCH_ONLY=hip python tools/train_bench.py 40 2>&1 | grep -v amdgpu.ids | tail -2
done; done
python tools/train_op_count.py 2>&1 | grep -v amdgpu.ids | tail -12
