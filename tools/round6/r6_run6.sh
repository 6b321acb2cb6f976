set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6f
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_eval.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
timeout 600 python tools/eval_graph_prof.py 40 > $O/eval_graph_prof.txt 2>&1
timeout 300 python tools/eval_bench.py 40 > $O/eval_bench.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/eval_stats -o eval -- python3 $R/tools/eval_graph_prof.py 40 > $O/eval_rocprof.log 2>&1
cp $O/eval_stats/eval_kernel_stats.csv $O/eval_config3_kernel_stats.csv
cd $R
tail -4 $O/tests.log; cat $O/eval_graph_prof.txt; cat $O/eval_bench.txt; head -12 $O/eval_config3_kernel_stats.csv | cut -c1-150
