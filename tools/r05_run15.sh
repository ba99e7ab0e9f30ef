cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05c
timeout 2400 python -m pytest tests/test_ddp_gpu.py -q -m gpu -rs 2>&1 | tail -40 > gpurun_out/r05c/ddp_tests.txt
tail -15 gpurun_out/r05c/ddp_tests.txt
