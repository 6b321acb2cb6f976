cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6E
timeout 900 python -m pytest tests/test_gpu_train_graph.py -q -m gpu -x > gpurun_out/r6E/tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r6E/tests.log | cut -c1-250
for m in direct staged direct staged; do
ADAISP_TRAIN_GRAPH_UPLOAD=$m timeout 300 python -m adaptiveisp_amd.train --iters 310 --warmup 10 2>&1 | grep -v amdgpu | tail -1 | cut -c1-130; done
bash tools/train_timeline.sh 2>&1 | grep -v amdgpu | cut -c1-200 | head -36; rm -rf gpurun_out/train_timeline
