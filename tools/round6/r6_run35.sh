cd $GRAFT_REPO_ROOT
timeout 600 python tools/round6/dbg_race.py 2>&1 | grep -v amdgpu | cut -c1-300
