cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6K
timeout 1800 python -m pytest tests/test_gpu_train_graph.py tests/test_gpu_train.py tests/test_gpu_trunk_train.py tests/test_gpu_agent.py tests/test_gpu_multirank_rehearsal.py -q -m gpu -x > gpurun_out/r6K/tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r6K/tests.log | cut -c1-250
timeout 600 python tools/train_graph_ab.py --iters 40 --rounds 3 > gpurun_out/r6K/graph_ab.txt 2>&1
grep -v amdgpu gpurun_out/r6K/graph_ab.txt | tail -2 | cut -c1-250
