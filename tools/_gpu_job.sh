python -m pytest tests/test_checkpoint_import.py tests/test_gpu_agent.py tests/test_gpu_train.py -m gpu -q -x -s 2>&1 | tail -40 > gpurun_out/r2_t2.log
python bench.py --steps 20 --warmup 3 > gpurun_out/r2_bench1.json 2> gpurun_out/r2_bench1.err
tail -3 gpurun_out/r2_t2.log; head -c 3000 gpurun_out/r2_bench1.json
