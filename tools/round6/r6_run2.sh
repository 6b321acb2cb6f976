set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6b
timeout 300 python tools/k1_bench.py --sets 2 > gpurun_out/r6b/k1_bench_sets2.txt 2>&1
timeout 300 python tools/k1_bench.py --sets 24 > gpurun_out/r6b/k1_bench_sets24.txt 2>&1
timeout 600 python tools/engine_env_ab.py "ADAYOLO_K1=0" "ADAYOLO_K1=1" --rounds 12 --per-layer > gpurun_out/r6b/k1_ab.txt 2>&1
bash tools/headline_ab.sh "ADAYOLO_K1=0" "ADAYOLO_K1=1" 4 > gpurun_out/r6b/headline_ab_k1.txt 2>&1
cat gpurun_out/r6b/k1_bench_sets2.txt; grep "detector forward" gpurun_out/r6b/k1_ab.txt; tail -2 gpurun_out/r6b/headline_ab_k1.txt
