for r in 1 2; do
for m in "1 1" "1 0" "0 0"; do set -- $m
echo "== ADAYOLO_TRAIN_PAIR=$1 ADAISP_CRITIC_STREAM=$2"
ADAYOLO_TRAIN_PAIR=$1 ADAISP_CRITIC_STREAM=$2 TRAIN_BENCH_ONLY=hip python tools/train_bench.py 40 2>&1 | grep -v amdgpu.ids | tail -2 | head -1
done; done
