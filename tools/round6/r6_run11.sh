set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6k
mkdir -p $O
ADAYOLO_LIB=build/variants/measure/libadayolo.so timeout 300 python tools/stem_down_stamps.py > $O/stem_down_stamps.txt 2>&1
cat $O/stem_down_stamps.txt
bash tools/selfcheck.sh > $O/tools_selfcheck.txt 2>&1
cat $O/tools_selfcheck.txt
for i in 1 2 3; do
  rm -f gpurun_out/parity_margins.txt
  timeout 1800 python -m pytest tests -q -m gpu > $O/all_gpu_tests_$i.log 2>&1; echo "all tests rc=$?" >> $O/all_gpu_tests_$i.log
  tail -3 $O/all_gpu_tests_$i.log
  cp gpurun_out/parity_margins.txt $O/parity_margins_run$i.txt
done
