set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6p
python bench.py > gpurun_out/r6p/bench.json 2> gpurun_out/r6p/bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6p/bench_driver_style.json 2> gpurun_out/r6p/bench_driver_style.err
head -c 400 gpurun_out/r6p/bench.json; echo; head -c 300 gpurun_out/r6p/bench_driver_style.json
