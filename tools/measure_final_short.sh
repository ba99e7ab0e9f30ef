R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03f; mkdir -p $O; cd $R
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_final.json 2> $O/bench_final.err
timeout 300 python tools/conv3x3_probe.py > $O/conv3x3_probe.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof -o t --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --no_cpu_baseline --no_kernel_timing > /tmp/prof.log 2>&1
python $R/tools/trace_summary.py /tmp/prof/t_kernel_trace.csv $O/step_kernel_summary_final.txt "timeout 400 rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 2 (round-3 final)" > /dev/null
head -40 /tmp/prof/t_kernel_stats.csv > $O/kernel_stats_final.csv
cd $R; cut -c1-250 $O/bench_final.json; head -5 $O/step_kernel_summary_final.txt
