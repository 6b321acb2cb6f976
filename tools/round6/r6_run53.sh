cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6N
timeout 900 python tools/train_graph_ab.py --iters 40 --rounds 4 --headsk 0,1 > gpurun_out/r6N/heads_ab.txt 2>&1
grep -v amdgpu gpurun_out/r6N/heads_ab.txt | grep "median\|!!" | cut -c1-220
ADAISP_TRAIN_GRAPH_STREAMS=1 timeout 900 python tools/train_graph_ab.py --iters 40 --rounds 3 --headsk 0,1 2>&1 | grep -v amdgpu | grep "median\|!!" | cut -c1-220
