set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6l
mkdir -p $O
# soak of the round's new kernels: 1000 pipelined replays of the headline (bneck_ws + chain inside, ISP stream beside), then 20 rounds
# of the bneck_ws / chain / eval tests
timeout 1200 python tools/pipeline_stress.py 1000 > $O/pipeline_stress.txt 2>&1; echo "rc=$?" >> $O/pipeline_stress.txt
tail -3 $O/pipeline_stress.txt
for i in $(seq 1 20); do timeout 600 python -m pytest tests/test_gpu_yolo_bneck_ws.py tests/test_gpu_yolo_chain.py tests/test_gpu_eval.py tests/test_gpu_bench_pipeline.py -x -q -m gpu 2>&1 | tail -1; done > $O/soak_tests.txt 2>&1
cat $O/soak_tests.txt
