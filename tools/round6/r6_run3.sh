set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6c
timeout 900 python -m pytest tests/test_gpu_yolo_bneck_ws.py -x -q -m gpu > gpurun_out/r6c/bneck_ws_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r6c/bneck_ws_tests.log
timeout 600 python tools/engine_env_ab.py "ADAYOLO_BNECK_WS=0" "ADAYOLO_BNECK_WS=1" --rounds 12 --per-layer > gpurun_out/r6c/bneck_ws_ab.txt 2>&1
bash tools/headline_ab.sh "ADAYOLO_BNECK_WS=0" "ADAYOLO_BNECK_WS=1" 4 > gpurun_out/r6c/headline_ab_bneck_ws.txt 2>&1
tail -5 gpurun_out/r6c/bneck_ws_tests.log; grep "detector forward\|bneckws\|ws\|128->64\|v90" gpurun_out/r6c/bneck_ws_ab.txt; tail -2 gpurun_out/r6c/headline_ab_bneck_ws.txt
