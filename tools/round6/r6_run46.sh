cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6J
timeout 900 python tools/train_soak.py 20000 2>&1 | grep -v amdgpu > gpurun_out/r6J/soak_default.txt
tail -5 gpurun_out/r6J/soak_default.txt
ADAISP_TRAIN_GRAPH_STREAMS=1 timeout 900 python tools/train_soak.py 10000 2>&1 | grep -v amdgpu > gpurun_out/r6J/soak_one_stream.txt
tail -3 gpurun_out/r6J/soak_one_stream.txt
