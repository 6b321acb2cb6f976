set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6t
timeout 300 python tools/bneck_ws_lib_ab.py noprio=build/variants/bwsnoprio/libadayolo.so > gpurun_out/r6t/bws_prio_ab.txt 2>&1
grep -v amdgpu gpurun_out/r6t/bws_prio_ab.txt
