set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_yolo_chain.py -x -q -m gpu 2>&1 | tail -3
bash tools/refresh_profiles.sh round6
bash tools/chain_pmc.sh round6
bash tools/headline_ab.sh "ADAYOLO_BNECK_WS=0 ISP_DUMMY=0" "ADAYOLO_BNECK_WS=1" 4 > gpurun_out/round6_headline_ab_bneck_ws.txt 2>&1
tail -2 gpurun_out/round6_headline_ab_bneck_ws.txt
