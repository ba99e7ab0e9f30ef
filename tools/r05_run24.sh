cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_conv1x1_fused_gpu.py -q -m gpu -k "flip" 2>&1 | tail -4
timeout 600 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_kernel_timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
