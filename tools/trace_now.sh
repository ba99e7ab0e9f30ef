# one rocprofv3 kernel trace of the benchmark step -> gpurun_out/$1/step_kernel_summary.txt  (usage: bash tools/trace_now.sh <tag> [bench flags])
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; TAG=${1:-trace}; [ $# -gt 0 ] && shift; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o t --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --no_cpu_baseline --no_kernel_timing "$@" > /tmp/prof_$TAG.log 2>&1
if [ ! -f /tmp/prof_$TAG/t_kernel_trace.csv ]; then echo "trace_now.sh: no kernel trace was written:" >&2; tail -20 /tmp/prof_$TAG.log >&2; exit 1; fi
python $R/tools/trace_summary.py /tmp/prof_$TAG/t_kernel_trace.csv $O/step_kernel_summary.txt "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 2 $*" > /dev/null
head -4 $O/step_kernel_summary.txt; tail -16 $O/step_kernel_summary.txt
