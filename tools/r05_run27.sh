cd $GRAFT_REPO_ROOT
timeout 900 python tests/diag/replay_noise_diag.py 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tail -14
timeout 600 python -m pytest tests/test_conv1x1_fused_gpu.py -q -m gpu -k "atomic_links" 2>&1 | grep -B30 "short test summary" | head -60
