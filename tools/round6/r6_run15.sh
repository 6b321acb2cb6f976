set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6o
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_agent.py -x -q -m gpu -k narrow > $O/narrow_test.log 2>&1; grep -n "^E " $O/narrow_test.log | head -20; tail -3 $O/narrow_test.log
run() { env $1 python bench.py --steps 40 --warmup 8 --no-detail --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])"; }
for r in 1 2 3; do
  for w in 0 6 12 24; do echo "round $r ADAISP_EXPERIMENT_EXTRA_LAUNCHES=$w: $(run ADAISP_EXPERIMENT_EXTRA_LAUNCHES=$w) ms"; done
done > $O/extra_launches_sweep.txt 2>&1
cat $O/extra_launches_sweep.txt
