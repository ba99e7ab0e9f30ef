cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_conv1x1_gpu.py -q -m gpu -x 2>&1 | tail -8
