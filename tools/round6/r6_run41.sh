cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6H
timeout 900 python tools/train_graph_ab.py --iters 40 --rounds 4 --streams 2,1 > gpurun_out/r6H/streams_ab.txt 2>&1
grep -v amdgpu gpurun_out/r6H/streams_ab.txt | tail -10 | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_train_graph.py -q -m gpu -x 2>&1 | tail -3
