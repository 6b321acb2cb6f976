set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6v
timeout 900 python -m pytest tests/test_gpu_train_graph.py -q -m gpu > gpurun_out/r6v/tests.log 2>&1
echo "tests rc=$?"
tail -60 gpurun_out/r6v/tests.log
TRAIN_GRAPH_AB_PROFILE=1 timeout 600 python tools/train_graph_ab.py --iters 40 --rounds 1 > gpurun_out/r6v/train_graph_prof.txt 2>&1
echo "ab rc=$?"
grep -v amdgpu gpurun_out/r6v/train_graph_prof.txt | cut -c1-200 | tail -75
