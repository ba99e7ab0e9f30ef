R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r6_cold; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cp
timeout 600 rocprofv3 --kernel-trace -d /tmp/cp -o t --output-format csv -- python3 $R/tools/cold_probe.py > $O/log.txt 2>&1
python3 $R/tools/cold_probe_parse.py /tmp/cp/t_kernel_trace.csv > $O/cold_probe.txt 2>&1
cat $O/cold_probe.txt
