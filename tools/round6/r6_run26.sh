cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6y
timeout 1500 python -m pytest tests/test_gpu_train_graph.py tests/test_gpu_train.py tests/test_gpu_multirank_rehearsal.py tests/test_gpu_yolo_train.py -q -m gpu > gpurun_out/r6y/tests.log 2>&1
echo "tests rc=$?"
tail -30 gpurun_out/r6y/tests.log | cut -c1-300
timeout 600 python -m adaptiveisp_amd.train --iters 3010 --warmup 10 > gpurun_out/r6y/soak_graph.txt 2>&1
echo "soak rc=$?"
grep -v amdgpu gpurun_out/r6y/soak_graph.txt | cut -c1-1500 | tail -3
ADAISP_TRAIN_GRAPH=0 timeout 600 python -m adaptiveisp_amd.train --iters 3010 --warmup 10 > gpurun_out/r6y/soak_ordinary.txt 2>&1
grep -v amdgpu gpurun_out/r6y/soak_ordinary.txt | cut -c1-700 | tail -2
