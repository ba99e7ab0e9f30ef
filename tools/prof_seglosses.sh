cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/pk2 -o k --output-format csv -- python3 $R/tools/seglosses_bench.py > /tmp/pk2.log 2>&1
python3 - <<'PY'
import csv, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open('/tmp/pk2/k_kernel_trace.csv')):
    agg[r["Kernel_Name"][:80]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:6]:
    print("%-82s n=%3d avg %.1f us  min %.1f" % (k, len(v), sum(v) / len(v), min(v)))
PY
timeout 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace -d /tmp/psq2 -o s --output-format csv -- python3 $R/tools/seglosses_bench.py > /tmp/psq2.log 2>&1
python3 - <<'PY'
import csv, collections, glob
f = glob.glob('/tmp/psq2/*counter_collection.csv')
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "seg_losses_pk_kernel" not in k and "seg_losses_kernel" not in k: continue
    agg[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    meta = (r.get("VGPR_Count"), r.get("LDS_Block_Size"), r.get("Grid_Size"), r.get("Workgroup_Size"))
for name, cs in agg.items():
    print(name, meta)
    for c, v in sorted(cs.items()): print("   %-28s %.4e" % (c, sum(v) / len(v)))
PY
tail -3 /tmp/psq2.log
