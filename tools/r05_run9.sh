cd $GRAFT_REPO_ROOT; python tests/diag/bn3_bias_diag.py 2>&1 | grep -v amdgpu.ids
