set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6i
mkdir -p $O
for i in 1 2 3 4 5; do timeout 600 python -m pytest tests/test_gpu_eval.py -x -q -m gpu 2>&1 | tail -2; done > $O/eval_x5.log 2>&1
cat $O/eval_x5.log
timeout 1800 python -m pytest tests -q -m gpu > $O/all_gpu_tests.log 2>&1; echo "all tests rc=$?" >> $O/all_gpu_tests.log
tail -8 $O/all_gpu_tests.log
cp gpurun_out/parity_margins.txt $O/parity_margins_run1.txt 2>/dev/null
