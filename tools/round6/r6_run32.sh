cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6D
timeout 1500 python -m pytest tests/test_gpu_train_graph.py tests/test_gpu_train.py tests/test_gpu_multirank_rehearsal.py -q -m gpu -x > gpurun_out/r6D/tests.log 2>&1
echo "tests rc=$?"; tail -30 gpurun_out/r6D/tests.log | cut -c1-250
timeout 600 python tools/train_graph_ab.py --iters 40 --rounds 3 > gpurun_out/r6D/graph_ab.txt 2>&1
grep -v amdgpu gpurun_out/r6D/graph_ab.txt | tail -8 | cut -c1-250
bash tools/train_timeline.sh 2>&1 | grep -v amdgpu | cut -c1-200 | head -60; rm -rf gpurun_out/train_timeline
