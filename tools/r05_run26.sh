cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_conv1x1_gpu.py -q -m gpu 2>&1 | tail -3
timeout 300 python tools/wgrad_probe2.py 0 2>&1 | grep -v amdgpu.ids | grep "1x1\|per step"
UCD_WGRAD_REG=0 timeout 300 python tools/wgrad_probe2.py 0 2>&1 | grep -v amdgpu.ids | grep "1x1\|per step"
