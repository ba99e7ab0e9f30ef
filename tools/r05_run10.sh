cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_conv1x1_fused_gpu.py -q -m gpu -k "atomic_links_survive" 2>&1 | grep -A12 "^>" | tail -40
