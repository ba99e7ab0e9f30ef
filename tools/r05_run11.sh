cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof1 -o t --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --no_cpu_baseline --no_kernel_timing > /tmp/prof1.log 2>&1
python $R/tools/trace_summary.py /tmp/prof1/t_kernel_trace.csv $O/step_kernel_summary.txt "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 2 (round 5, atomic statistics)" > /dev/null
sed -n '/GEMM kernels by grid/,$p' $O/step_kernel_summary.txt; sed -n '/ABN stream kernels/,/GEMM kernels/p' $O/step_kernel_summary.txt | head -30
