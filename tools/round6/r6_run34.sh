cd $GRAFT_REPO_ROOT
timeout 600 python tools/round6/dbg_first_replay.py 2>&1 | grep -v amdgpu | cut -c1-400
