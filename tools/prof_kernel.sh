# SQ counters (one pass, 8 slots) + durations of the kernels whose name matches $1, for the command after `--`:
#   bash tools/prof_kernel.sh conv1x1_kernel -- python3 tools/conv3x3_probe.py      (a relative program path is resolved against the repo root)
# -> gpurun_out/prof_kernel/<pattern>.txt   (put the program itself after --: no env / bash -c hops under rocprofv3)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; PAT=$1
if [ -z "$PAT" ] || [ "$2" != "--" ] || [ $# -lt 3 ]; then echo "usage: prof_kernel.sh <kernel-name-pattern> -- <program> [args]" >&2; exit 2; fi
shift; shift; O=$R/gpurun_out/prof_kernel; mkdir -p $O
# the profiler runs from /tmp: make every relative path argument that names a file of the repo absolute first
ARGS=(); for a in "$@"; do if [ "${a#/}" = "$a" ] && [ -e "$R/$a" ]; then a="$R/$a"; fi; ARGS+=("$a"); done; set -- "${ARGS[@]}"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk_t /tmp/pk_c
timeout 400 rocprofv3 --kernel-trace -d /tmp/pk_t -o t --output-format csv -- "$@" > /tmp/pk_t.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d /tmp/pk_c -o s --output-format csv -- "$@" > /tmp/pk_c.log 2>&1
if ! ls /tmp/pk_t/*kernel_trace.csv > /dev/null 2>&1; then echo "prof_kernel.sh: no kernel trace was written:" >&2; tail -20 /tmp/pk_t.log >&2; exit 1; fi
if ! ls /tmp/pk_c/*counter_collection.csv > /dev/null 2>&1; then echo "prof_kernel.sh: no counter file was written:" >&2; tail -20 /tmp/pk_c.log >&2; exit 1; fi
PK_PAT="$PAT" python3 - <<'PY' > $O/$PAT.txt
import csv, collections, glob, os
pat = os.environ["PK_PAT"]
dur = collections.OrderedDict()
for r in csv.DictReader(open(glob.glob('/tmp/pk_t/*kernel_trace.csv')[0])):
    n = r['Kernel_Name']
    if pat not in n: continue
    key = (n[:110], r.get('Grid_Size', r.get('Grid_Size_X')), r.get('Workgroup_Size', r.get('Workgroup_Size_X')))
    dur.setdefault(key, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
cnt = collections.OrderedDict()
for r in csv.DictReader(open(glob.glob('/tmp/pk_c/*counter_collection.csv')[0])):
    n = r['Kernel_Name']
    if pat not in n: continue
    key = (n[:110], r.get('Grid_Size', r.get('Grid_Size_X')), r.get('Workgroup_Size', r.get('Workgroup_Size_X')))
    cnt.setdefault(key, collections.defaultdict(list))[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in dur.items():
    v = sorted(v)
    m = {c: sum(x) / len(x) for c, x in cnt.get(k, {}).items()}
    w = m.get('SQ_WAVE_CYCLES', 0) or 1
    print("%8.1f us (n=%3d) grid %8s | mfma/wave %.3f lds-conflict %.3f wait_any %.2f wait_inst %.2f active %.2f | %s" % (
        v[len(v) // 2], len(v), k[1], m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (4 * w),
        m.get('SQ_LDS_BANK_CONFLICT', 0) / max(1, m.get('SQ_LDS_IDX_ACTIVE', 1)), m.get('SQ_WAIT_ANY', 0) / w,
        m.get('SQ_WAIT_INST_ANY', 0) / w, m.get('SQ_ACTIVE_INST_ANY', 0) / w, k[0][30:]))
PY
cat $O/$PAT.txt
