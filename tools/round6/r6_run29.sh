cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6B
timeout 1500 python -m pytest tests/test_gpu_yolo_train.py tests/test_gpu_train.py tests/test_gpu_train_graph.py -q -m gpu -x > gpurun_out/r6B/tests.log 2>&1
echo "tests rc=$?"; tail -25 gpurun_out/r6B/tests.log | cut -c1-250
timeout 900 python tools/train_graph_ab.py --iters 40 --rounds 3 --early 0,3,2,4 > gpurun_out/r6B/early_ab.txt 2>&1
echo "ab rc=$?"; grep -v amdgpu gpurun_out/r6B/early_ab.txt | tail -20 | cut -c1-250
timeout 600 python tools/train_graph_ab.py --iters 40 --rounds 2 > gpurun_out/r6B/graph_ab.txt 2>&1
grep -v amdgpu gpurun_out/r6B/graph_ab.txt | tail -3 | cut -c1-250
