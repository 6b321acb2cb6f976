cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6P
timeout 900 python tools/train_soak.py 8000 2>&1 | grep -v amdgpu > gpurun_out/r6P/soak_heads.txt
tail -3 gpurun_out/r6P/soak_heads.txt
bash tools/round6/r6_run48.sh
