set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6a
timeout 900 python -m pytest tests/test_gpu_yolo_k1.py tests/test_gpu_yolo_chain.py tests/test_cabi.py -x -q -m gpu > gpurun_out/r6a/chain_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r6a/chain_tests.log
timeout 300 python tools/k1_bench.py --sets 2 > gpurun_out/r6a/k1_bench_sets2.txt 2>&1
timeout 300 python tools/k1_bench.py --sets 24 > gpurun_out/r6a/k1_bench_sets24.txt 2>&1
for i in 1 2; do
  timeout 300 python tools/chain_ab.py --rounds 8 > gpurun_out/r6a/chain_ab_invwait1_$i.txt 2>&1
  ADAYOLO_LIB=build/variants/noinvwait/libadayolo.so timeout 300 python tools/chain_ab.py --rounds 8 > gpurun_out/r6a/chain_ab_invwait0_$i.txt 2>&1
done
timeout 600 python bench.py > gpurun_out/r6a/bench.json 2> gpurun_out/r6a/bench.err
tail -3 gpurun_out/r6a/chain_tests.log; cat gpurun_out/r6a/k1_bench_sets2.txt
