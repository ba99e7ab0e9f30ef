cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_abn_gpu.py -q -m gpu -k "stem" 2>&1 | tail -12
for v in 0 1 0 1; do UCD_STEM_EVAL_FUSED=$v timeout 300 python bench.py --steps 20 --warmup 6 --no_cpu_baseline --no_kernel_timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('UCD_STEM_EVAL_FUSED=$v', round(d['ms_per_step'],3), d['losses']['loss'])"; done
