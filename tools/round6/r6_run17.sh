set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6q
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_yolo_variants.py -x -q -m gpu -k "stride2_weights" 2>&1 | tail -15
timeout 300 python tools/conv_bench.py 80,95,60,85 small > $O/dws_bench.txt 2>&1
timeout 300 python tools/conv_bench.py 95,80,60,85 small >> $O/dws_bench.txt 2>&1
grep -v amdgpu $O/dws_bench.txt
