R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r04_pipe; mkdir -p $O; cd $R
PIPES="2x64 lw256" bash tools/r04_pipe.sh | head -22
UCD_CONV_PIPE=lw256 timeout 900 python -m pytest tests/test_conv1x1_gpu.py tests/test_conv1x1_fused_gpu.py -x -q > $O/tests_lw256.txt 2>&1; tail -5 $O/tests_lw256.txt
