cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6M
timeout 1800 python -m pytest tests/test_gpu_heads_train.py tests/test_gpu_train_graph.py tests/test_gpu_train.py tests/test_gpu_trunk_train.py tests/test_gpu_agent.py -q -m gpu > gpurun_out/r6M/tests.log 2>&1
echo "tests rc=$?"; tail -25 gpurun_out/r6M/tests.log | cut -c1-250
grep "heads_train" gpurun_out/parity_margins.txt | cut -c1-160
