# The bench line, the kernel trace and the per-rank batches of the FINAL commit (the full collection: tools/measure_round.sh)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r05f; mkdir -p $O
cd $R
timeout 900 python bench.py --steps 20 --warmup 6 > $O/bench_final.json 2> $O/bench_final.err
brief() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms_per_step', round(d['ms_per_step'],3), 'img/s', round(d['value'],1), 'step_graph', d['execution']['step_graph'])"; }
(for gb in 12 6 3; do UCD_STEP_GRAPH=auto timeout 300 python bench.py --steps 16 --warmup 6 --global_batch $gb --no_cpu_baseline --no_kernel_timing 2>/dev/null | brief "global_batch $gb"; done) > $O/small_batch.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof -o t --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --no_cpu_baseline --no_kernel_timing > /tmp/prof.log 2>&1
UCD_TRACE_TOP=140 python $R/tools/trace_summary.py /tmp/prof/t_kernel_trace.csv $O/step_kernel_summary_final.txt "timeout 400 rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 2 (round 5, final commit; the timed steps are replays of the captured step graph)" > /dev/null
head -40 /tmp/prof/t_kernel_stats.csv > $O/kernel_stats_final.csv
cd $R
cut -c1-300 $O/bench_final.json; cat $O/small_batch.txt; head -5 $O/step_kernel_summary_final.txt; grep "stem_\|seg_losses\|flip_\|prep_" $O/step_kernel_summary_final.txt | head -12
