cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05b; mkdir -p $O
timeout 1200 python -m pytest tests/test_conv1x1_fused_gpu.py -q -m gpu -s -k "bench_shape_block_chain or bench_shape_aspp" 2>&1 | grep -v "^$" | grep "parameter gradients\|bench-shape\|passed\|failed\|Error" > $O/tests_chain.txt; cat $O/tests_chain.txt
