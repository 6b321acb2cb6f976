cd $GRAFT_REPO_ROOT
timeout 900 python tools/train_graph_ab.py --iters 40 --rounds 4 --streams 2,1 2>&1 | grep -v amdgpu | grep "!!\|median" | cut -c1-250
echo ----
timeout 900 python tools/train_graph_ab.py --iters 40 --rounds 4 --streams 1,2 2>&1 | grep -v amdgpu | grep "!!\|median" | cut -c1-250
echo ----
timeout 900 python tools/train_graph_ab.py --iters 40 --rounds 4 --streams 1,1 2>&1 | grep -v amdgpu | grep "!!\|median" | cut -c1-250
