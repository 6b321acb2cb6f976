cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6G
timeout 1500 python -m pytest tests/test_gpu_train_graph.py tests/test_gpu_train.py tests/test_gpu_multirank_rehearsal.py -q -m gpu -x > gpurun_out/r6G/tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r6G/tests.log | cut -c1-250
for q in default 1 2 8 default; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  echo "GPU_MAX_HW_QUEUES=$q: $(timeout 300 python -m adaptiveisp_amd.train --iters 310 --warmup 10 2>&1 | grep -v amdgpu | tail -1 | cut -c1-120)"
done
unset GPU_MAX_HW_QUEUES
timeout 600 python tools/train_graph_ab.py --iters 40 --rounds 3 > gpurun_out/r6G/graph_ab.txt 2>&1
grep -v amdgpu gpurun_out/r6G/graph_ab.txt | tail -2 | cut -c1-250
