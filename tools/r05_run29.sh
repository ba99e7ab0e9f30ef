cd $GRAFT_REPO_ROOT
timeout 2000 python -m pytest tests/test_step_gpu.py tests/test_model_gpu.py -q -m gpu 2>&1 | tail -5
for v in 0 1 0 1; do UCD_VECTOR_CONV=$v timeout 300 python bench.py --steps 20 --warmup 6 --no_cpu_baseline --no_kernel_timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('UCD_VECTOR_CONV=$v', round(d['ms_per_step'],3))"; done
