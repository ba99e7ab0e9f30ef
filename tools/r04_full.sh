# whole GPU suite + bench at 24 and 3 images -> gpurun_out/r04_full/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r04_full; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests_gpu.txt 2>&1; tail -8 $O/tests_gpu.txt
SKIP_CONV_TESTS=1 GBS="24 3" bash tools/r04_quick.sh | tail -2
