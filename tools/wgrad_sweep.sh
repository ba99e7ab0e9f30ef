R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r04_wgrad; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_conv1x1_fused_gpu.py -x -q -k "wgrad" > $O/tests.txt 2>&1; tail -15 $O/tests.txt
UCD_WGRAD3=0 timeout 300 python tools/wgrad_probe2.py 2>&1 | grep -v amdgpu | grep "3x3\|per step" > $O/probe_9tap.txt
timeout 300 python tools/wgrad_probe2.py 2>&1 | grep -v amdgpu | grep "3x3\|per step" > $O/probe_3tap.txt
for t in 384 512; do UCD_WGRAD3_TARGET=$t timeout 300 python tools/wgrad_probe2.py 2>&1 | grep -v amdgpu | grep "3x3\|per step" > $O/probe_3tap_t$t.txt; done
tail -n 20 $O/probe_*.txt
