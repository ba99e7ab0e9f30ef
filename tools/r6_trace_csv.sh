# raw kernel trace (last replay only) of a bench configuration -> gpurun_out/r6_trace/<name>.csv   usage: r6_trace_csv.sh name [bench args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r6_trace; mkdir -p $O; N=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d /tmp/pt_$N -o t --output-format csv -- python3 $R/bench.py "$@" --steps 3 --warmup 2 --no_cpu_baseline --no_kernel_timing > /tmp/pt_$N.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open('/tmp/pt_$N/t_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'pixcon_reduce_kernel' in r['Kernel_Name']]
win=rows[idx[-2]:idx[-1]]
t0=int(win[0]['Start_Timestamp'])
with open('$O/$N.csv','w') as f:
    for r in win:
        f.write('%d,%d,%s,%s,%s\n' % (int(r['Start_Timestamp'])-t0, int(r['End_Timestamp'])-t0, r.get('Queue_Id','?'), r.get('Grid_Size_X','?'), r['Kernel_Name'][:110].replace(',',';')))
print('$N', len(win), 'kernels')
PY
