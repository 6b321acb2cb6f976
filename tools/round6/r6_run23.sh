cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6w
timeout 600 python tools/round6/dbg_graph_iter.py > gpurun_out/r6w/dbg.txt 2>&1
grep -v amdgpu gpurun_out/r6w/dbg.txt | cut -c1-400 | tail -30
