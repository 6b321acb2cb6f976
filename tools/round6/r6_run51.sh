cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6L
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6L/smoke.txt 2>&1; echo "smoke rc=$?"
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/r6L/all_gpu_tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r6L/all_gpu_tests.log
cp gpurun_out/parity_margins.txt gpurun_out/r6L/parity_margins_run6.txt
