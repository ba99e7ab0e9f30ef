cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no_cpu_baseline --no_kernel_timing > /tmp/prof.log 2>&1
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r05e
UCD_TRACE_TOP=140 python $GRAFT_REPO_ROOT/tools/trace_summary.py /tmp/prof/t_kernel_trace.csv $GRAFT_REPO_ROOT/gpurun_out/r05e/step_summary140.txt "top 140" > /dev/null
