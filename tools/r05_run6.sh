cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05b; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/tests_gpu.txt 2>&1; tail -40 $O/tests_gpu.txt
for sa in 0 1 0 1; do UCD_STAT_ATOMIC=$sa timeout 600 python bench.py --steps 20 --warmup 6 --no_cpu_baseline --no_kernel_timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('UCD_STAT_ATOMIC=$sa', 'ms_per_step', round(d['ms_per_step'],3), 'img/s', round(d['value'],1), d['losses'])"; done > $O/bench_atomic_ab.txt 2>&1; cat $O/bench_atomic_ab.txt
