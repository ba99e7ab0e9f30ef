cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/pk -o k --output-format csv -- python3 $R/tools/pixcon_bench.py f16 dom > /tmp/pk.log 2>&1
python3 - <<'PY'
import csv, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open('/tmp/pk/k_kernel_trace.csv')):
    agg[r["Kernel_Name"][:90]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if 'pixcon' in k: print("%-92s n=%3d avg %.1f us  min %.1f" % (k, len(v), sum(v) / len(v), min(v)))
PY
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d /tmp/psq -o s --output-format csv -- python3 $R/tools/pixcon_bench.py f16 dom > /tmp/psq.log 2>&1
python3 - <<'PY'
import csv, collections, glob
f = glob.glob('/tmp/psq/*counter_collection.csv')
agg = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "pixcon16" not in k: continue
    name = ("sweep1" if "ILi0E" in k else "sweep2_prob" if "ILi1ELb1" in k else "sweep2") if "sweep" in k else k.split("(")[0][-40:]
    agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    meta[name] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Grid_Size"), r.get("Workgroup_Size"))
for name, cs in agg.items():
    print(name, "VGPR/AGPR/SGPR/LDS/grid/wg =", meta[name], "dispatches", len(next(iter(cs.values()))))
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    for c, v in sorted(m.items()): print("   %-28s %.4e" % (c, v))
    if "SQ_WAVE_CYCLES" in m and m["SQ_WAVE_CYCLES"]:
        print("   mfma busy / (4 * wave quad-cycles) = %.3f   lds conflict / lds active = %.3f" % (m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * m["SQ_WAVE_CYCLES"]), m.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, m.get("SQ_LDS_IDX_ACTIVE", 1))))
PY
