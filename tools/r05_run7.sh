cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05b; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/tests_gpu3.txt 2>&1; grep -n "^E  \|^FAILED\|passed\|failed" $O/tests_gpu3.txt | head -60
