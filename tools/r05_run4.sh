cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05a; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for sa in 0 1; do
  UCD_STAT_ATOMIC=$sa timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof$sa -o t --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --no_cpu_baseline --no_kernel_timing > /tmp/prof$sa.log 2>&1
  python $R/tools/trace_summary.py /tmp/prof$sa/t_kernel_trace.csv $O/step_kernel_summary_atomic$sa.txt "UCD_STAT_ATOMIC=$sa rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 2" > /dev/null
done
head -45 $O/step_kernel_summary_atomic0.txt; echo ======; head -45 $O/step_kernel_summary_atomic1.txt
