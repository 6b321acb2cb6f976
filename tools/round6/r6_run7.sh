set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6g
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_yolo_bneck_ws.py -x -q -m gpu > $O/bws_tests.log 2>&1; echo "tests rc=$?" >> $O/bws_tests.log
ADAYOLO_LIB=build/variants/measure/libadayolo.so timeout 300 python tools/bneck_ws_stamps.py > $O/bneck_ws_stamps.txt 2>&1
timeout 600 python tools/engine_env_ab.py "ADAYOLO_BNECK_WS=0" "ADAYOLO_BNECK_WS=1" --rounds 12 --per-layer > $O/bneck_ws_ab.txt 2>&1
bash tools/headline_ab.sh "ADAYOLO_BNECK_WS=0" "ADAYOLO_BNECK_WS=1" 4 > $O/headline_ab_bneck_ws.txt 2>&1
tail -3 $O/bws_tests.log; cat $O/bneck_ws_stamps.txt; grep "detector forward\|bneckws\|v90\|128->64" $O/bneck_ws_ab.txt; tail -2 $O/headline_ab_bneck_ws.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $O/all_gpu_tests.log 2>&1; echo "all tests rc=$?" >> $O/all_gpu_tests.log
tail -5 $O/all_gpu_tests.log
