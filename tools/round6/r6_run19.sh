set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6s
timeout 600 python tools/train_graph_probe.py > gpurun_out/r6s/train_graph_probe.txt 2>&1; echo "rc=$?" >> gpurun_out/r6s/train_graph_probe.txt
tail -30 gpurun_out/r6s/train_graph_probe.txt
