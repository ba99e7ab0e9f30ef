cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_conv1x1_fused_gpu.py -q -m gpu -k "aspp" 2>&1 | tail -12
for v in 0 1 0 1; do UCD_ASPP_FAN=$v timeout 300 python bench.py --steps 20 --warmup 6 --no_cpu_baseline --no_kernel_timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('UCD_ASPP_FAN=$v', round(d['ms_per_step'],3), d['losses']['loss'])"; done
