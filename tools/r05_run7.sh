cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05b; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x --deselect tests/test_step_gpu.py::test_twenty_step_trajectory_fp32_and_bf16_against_the_reference > $O/tests_gpu2.txt 2>&1; tail -30 $O/tests_gpu2.txt
