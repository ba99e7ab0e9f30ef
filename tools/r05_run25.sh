cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_step_gpu.py -q -m gpu -k "fix_bn" 2>&1 | tail -15
