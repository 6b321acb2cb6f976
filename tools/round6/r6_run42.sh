cd $GRAFT_REPO_ROOT
timeout 900 python tools/round6/dbg_nan.py 2>&1 | grep -v amdgpu | cut -c1-300
