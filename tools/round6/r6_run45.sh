cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6I
timeout 600 python tools/pipeline_graphs_ab.py 40 5 > gpurun_out/r6I/pipeline_graphs_ab.txt 2>&1
grep -v amdgpu gpurun_out/r6I/pipeline_graphs_ab.txt | tail -4 | cut -c1-250
