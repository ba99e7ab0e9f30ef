cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_pixcon_gpu.py -q -m gpu 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no_cpu_baseline --no_kernel_timing > /tmp/prof.log 2>&1
grep "prep_" /tmp/prof/t_kernel_stats.csv | cut -c1-160
