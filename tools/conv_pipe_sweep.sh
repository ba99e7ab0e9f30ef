R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r04_pipe; mkdir -p $O; cd $R
for p in ${PIPES:-2x64 lw32 lw64}; do UCD_CONV_PIPE=$p timeout 300 python tools/conv_pipe_probe.py > $O/probe_$p.txt 2>&1; done
paste -d'\n' $(for p in ${PIPES:-2x64 lw32 lw64}; do echo $O/probe_$p.txt; done) | grep -v amdgpu.ids > $O/probe_all.txt
cat $O/probe_all.txt
