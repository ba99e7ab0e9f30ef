cd $GRAFT_REPO_ROOT
timeout 300 python tools/overlap_probe.py 24 2>&1 | grep -v amdgpu.ids | tail -5
timeout 300 python tools/overlap_probe.py 3 2>&1 | grep -v amdgpu.ids | tail -3
