set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6r
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 1800 python -m pytest tests -q -m gpu > $O/all_gpu_tests.log 2>&1; echo "all tests rc=$?" >> $O/all_gpu_tests.log
tail -4 $O/all_gpu_tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench.err
head -c 300 $O/bench_driver_style.json
