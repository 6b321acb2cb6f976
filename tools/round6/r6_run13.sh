set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6m
bash tools/policy_cost_ab.sh 4 > gpurun_out/r6m/policy_cost_ab.txt 2>&1
cat gpurun_out/r6m/policy_cost_ab.txt
