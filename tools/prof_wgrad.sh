# kernel trace + SQ counters of the weight-gradient kernels on the probe's layer shapes -> gpurun_out/wgrad_prof/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wgrad_prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/wgp /tmp/wgc
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/wgp -o t --output-format csv -- python3 $R/tools/wgrad_probe2.py ${1:-512} > /tmp/wgp.log 2>&1
python3 - <<'PY' > $O/kernels.txt
import csv, collections
rows = list(csv.DictReader(open('/tmp/wgp/t_kernel_trace.csv')))
agg = collections.OrderedDict()
for r in rows:
    n = r['Kernel_Name']
    if 'wgrad' not in n: continue
    key = (n[:90], r.get('Grid_Size', r.get('Grid_Size_X')), r.get('LDS_Block_Size'))
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    agg.setdefault(key, []).append(d)
for k, v in agg.items():
    v = sorted(v)
    print("%8.1f us median  n=%3d  grid %8s lds %6s  %s" % (v[len(v)//2], len(v), k[1], k[2], k[0]))
PY
cat $O/kernels.txt
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d /tmp/wgc -o s --output-format csv -- python3 $R/tools/wgrad_probe2.py ${1:-512} > /tmp/wgc.log 2>&1
python3 - <<'PY' > $O/counters.txt
import csv, collections, glob
f = glob.glob('/tmp/wgc/*counter_collection.csv')
agg = collections.OrderedDict()
for r in csv.DictReader(open(f[0])):
    n = r['Kernel_Name']
    if 'wgrad_kernel' not in n: continue
    key = (n[:80], r['Grid_Size'])
    agg.setdefault(key, collections.defaultdict(list))[r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in agg.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    w = m.get('SQ_WAVE_CYCLES', 1)
    print(k[0][-40:], 'grid', k[1], ' mfma busy/(4*wave quad-cycles) %.3f  lds conflict/active %.3f  wait_any %.2f  wait_inst %.2f  active %.2f' % (
        m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (4 * w), m.get('SQ_LDS_BANK_CONFLICT', 0) / max(1, m.get('SQ_LDS_IDX_ACTIVE', 1)),
        m.get('SQ_WAIT_ANY', 0) / w, m.get('SQ_WAIT_INST_ANY', 0) / w, m.get('SQ_ACTIVE_INST_ANY', 0) / w))
PY
cat $O/counters.txt
