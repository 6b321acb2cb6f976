cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6C
timeout 900 python tools/train_graph_ab.py --iters 40 --rounds 4 --early 0,1,2,3 > gpurun_out/r6C/early_ab2.txt 2>&1
echo "ab rc=$?"; grep -v amdgpu gpurun_out/r6C/early_ab2.txt | tail -4 | cut -c1-250
timeout 900 python tools/train_graph_ab.py --iters 40 --rounds 3 --early 0,2 --ordinary > gpurun_out/r6C/early_ab_ordinary.txt 2>&1
echo "ab rc=$?"; grep -v amdgpu gpurun_out/r6C/early_ab_ordinary.txt | tail -2 | cut -c1-250
