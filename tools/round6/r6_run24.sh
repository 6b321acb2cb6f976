cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6x
timeout 900 python -m pytest tests/test_gpu_train_graph.py -q -m gpu > gpurun_out/r6x/tests.log 2>&1
echo "tests rc=$?"
tail -30 gpurun_out/r6x/tests.log | cut -c1-300
grep "train.graph" gpurun_out/parity_margins.txt | cut -c1-200
timeout 600 python tools/round6/dbg_graph_iter.py 2>&1 | grep -v amdgpu | grep "graph vs\|^graph\|ordinary value" | cut -c1-400
