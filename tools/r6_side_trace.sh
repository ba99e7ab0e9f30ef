# kernel trace of the 3-image step with the weight gradients on the compute stream / on the side stream: do they overlap?
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r6_side; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  export UCD_WGRAD_STREAM=$v
  timeout 400 rocprofv3 --kernel-trace -d /tmp/ps$v -o t --output-format csv -- python3 $R/bench.py --global_batch ${GB:-3} --steps 3 --warmup 2 --no_cpu_baseline --no_kernel_timing > /tmp/ps$v.log 2>&1
  UCD_TRACE_TOP=12 python $R/tools/trace_summary.py /tmp/ps$v/t_kernel_trace.csv $O/side$v.txt "UCD_WGRAD_STREAM=$v batch ${GB:-3}" > /dev/null
  python - <<PY
import csv
rows=list(csv.DictReader(open('/tmp/ps$v/t_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'pixcon_reduce_kernel' in r['Kernel_Name']]
win=rows[idx[-2]:idx[-1]]
qs={}
for r in win: qs.setdefault(r.get('Queue_Id','?'),[]).append(r)
print('STREAM=$v queues:', {k:len(v) for k,v in qs.items()})
wg=[r for r in win if 'wgrad' in r['Kernel_Name']]
print(' wgrad launches', len(wg), 'queues', sorted(set(r.get('Queue_Id','?') for r in wg)))
# overlap of wgrad kernels with non-wgrad kernels
oth=[(int(r['Start_Timestamp']),int(r['End_Timestamp'])) for r in win if 'wgrad' not in r['Kernel_Name']]
ov=0
for r in wg:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    for a,b in oth:
        if b<=s: continue
        if a>=e: break
        ov+=min(e,b)-max(s,a)
print(' wgrad time %.3f ms, of it overlapped with other kernels %.3f ms' % (sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in wg)/1e6, ov/1e6))
PY
  head -4 $O/side$v.txt
done
