set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6u
timeout 900 python -m pytest tests/test_gpu_train_graph.py -x -q -m gpu > gpurun_out/r6u/tests.log 2>&1
echo "tests rc=$?"
tail -40 gpurun_out/r6u/tests.log
timeout 600 python tools/train_graph_ab.py --iters 40 --rounds 3 > gpurun_out/r6u/train_graph_ab.txt 2>&1
echo "ab rc=$?"
grep -v amdgpu gpurun_out/r6u/train_graph_ab.txt | tail -30
