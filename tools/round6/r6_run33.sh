cd $GRAFT_REPO_ROOT
timeout 600 python tools/train_graph_hist.py 2>&1 | grep -v amdgpu | cut -c1-300
