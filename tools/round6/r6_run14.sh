set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6n
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_agent.py -x -q -m gpu 2>&1 | tail -3
run() { env $1 python bench.py --steps 40 --warmup 8 --no-detail --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])"; }
for r in 1 2 3; do
  for w in 0 16 24 32 48 64 128; do echo "round $r ADAISP_POLICY_WGS=$w: $(run ADAISP_POLICY_WGS=$w) ms"; done
done > $O/policy_wgs_sweep.txt 2>&1
cat $O/policy_wgs_sweep.txt
