cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05c
timeout 2400 python -m pytest tests/test_ddp_gpu.py -q -m gpu -x 2>&1 | tail -40 > gpurun_out/r05c/ddp_tests2.txt
tail -25 gpurun_out/r05c/ddp_tests2.txt
