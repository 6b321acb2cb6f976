cd $GRAFT_REPO_ROOT
timeout 900 python tools/train_graph_ab.py --iters 40 --rounds 4 --heads 1,0 2>&1 | grep -v amdgpu | grep "median\|!!" | cut -c1-200
