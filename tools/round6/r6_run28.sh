cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6A
PLAN_ORDER=1 timeout 600 python tools/train_det_breakdown.py 16 512 > gpurun_out/r6A/bd16.txt 2>&1
PLAN_ORDER=1 timeout 600 python tools/train_det_breakdown.py 8 512 > gpurun_out/r6A/bd8.txt 2>&1
grep -v amdgpu gpurun_out/r6A/bd16.txt | sed -n '/training forward:/,/training backward:/p' | cut -c1-170 | head -90
echo ======== B=8
grep -v amdgpu gpurun_out/r6A/bd8.txt | sed -n '/training forward:/,/training backward:/p' | cut -c1-170 | head -90
