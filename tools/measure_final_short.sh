# the short end-of-round refresh: bench line, kernel-trace summary and the contrastive micro-bench (≈6 GPU-minutes);
# tools/measure_round.sh is the full collection (PMC passes, probes, B=3 trace)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02s; mkdir -p $O
cd $R
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_final.json 2> $O/bench_final.err
(for a in "" dom; do timeout 100 python tools/pixcon_bench.py f16 $a | tail -2; done) > $O/pixcon_bench_f16.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o t --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --no_cpu_baseline --no_kernel_timing > /tmp/prof.log 2>&1
python $R/tools/trace_summary.py /tmp/prof/t_kernel_trace.csv $O/step_kernel_summary_final.txt "timeout 300 rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 2 (round-2 final)" > /dev/null
head -40 /tmp/prof/t_kernel_stats.csv > $O/kernel_stats_final.csv
cd $R; cut -c1-250 $O/bench_final.json; cat $O/pixcon_bench_f16.txt; head -12 $O/step_kernel_summary_final.txt
