R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r04_check; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_abn_gpu.py tests/test_conv1x1_fused_gpu.py tests/test_conv1x1_gpu.py -x -q > $O/tests.txt 2>&1; tail -4 $O/tests.txt
SKIP_CONV_TESTS=1 GBS="24 3" bash tools/r04_quick.sh | tail -2
