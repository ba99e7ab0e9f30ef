cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05a; mkdir -p $O
timeout 900 python -m pytest tests/test_conv1x1_fused_gpu.py -x -q -m gpu -k "atomic_statistics or resident_a" > $O/test_atomic.txt 2>&1; tail -25 $O/test_atomic.txt
timeout 900 python -m pytest tests/test_ddp_gpu.py -x -q -m gpu -k "aborted" > $O/test_aborted.txt 2>&1; tail -15 $O/test_aborted.txt
for sa in 0 1 0 1; do UCD_STAT_ATOMIC=$sa timeout 600 python bench.py --steps 20 --warmup 6 --no_cpu_baseline --no_kernel_timing 2>$O/bench_err_$sa.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('UCD_STAT_ATOMIC=$sa', 'ms_per_step', round(d['ms_per_step'],3), 'img/s', round(d['value'],1), d['losses'])"; done > $O/bench_atomic_ab.txt 2>&1; cat $O/bench_atomic_ab.txt; tail -5 $O/bench_err_1.txt
for sa in 0 1; do UCD_STAT_ATOMIC=$sa timeout 600 python bench.py --steps 20 --warmup 6 --global_batch 3 --no_cpu_baseline --no_kernel_timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b3 UCD_STAT_ATOMIC=$sa', 'ms_per_step', round(d['ms_per_step'],3), d['losses'])"; done >> $O/bench_atomic_ab.txt 2>&1; tail -2 $O/bench_atomic_ab.txt
cd /tmp && export TMPDIR=/tmp
UCD_STAT_ATOMIC=1 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof1 -o t --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --no_cpu_baseline --no_kernel_timing > /tmp/prof1.log 2>&1
python $R/tools/trace_summary.py /tmp/prof1/t_kernel_trace.csv $O/step_kernel_summary_atomic1_rep.txt "UCD_STAT_ATOMIC=1 (replicated accumulators) rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 2" > /dev/null
head -40 $O/step_kernel_summary_atomic1_rep.txt
