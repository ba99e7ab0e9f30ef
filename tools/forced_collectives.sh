# the N > 1 step on ONE GPU: a one-rank RCCL group with every SyncBN / gradient collective issued (bench.py --force_dist), at the per-rank
# batches of the 8- / 4- / 2-GPU split, with the whole-step graph and without, beside the plain single-process step
# usage: bash tools/forced_collectives.sh [out file]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; OUT=${1:-$R/gpurun_out/forced_collectives.txt}; mkdir -p "$(dirname "$OUT")"; cd $R
: > $OUT
for b in 3 6 12; do
  for mode in plain forced_graph forced_eager; do
    case $mode in
      plain) FL=""; SG=auto;;
      forced_graph) FL="--force_dist"; SG=1;;
      forced_eager) FL="--force_dist"; SG=0;;
    esac
    UCD_STEP_GRAPH=$SG timeout 600 python bench.py --global_batch $b --steps 20 --warmup 6 --no_cpu_baseline --no_kernel_timing $FL 2>/tmp/fc_err.txt | grep '^{' | tail -1 > /tmp/fc.json
    python - $b $mode /tmp/fc.json >> $OUT <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[3]))
    e = d["execution"]
    print("global_batch %2s %-13s: %6.2f ms/step  step_graph=%s forced_collectives=%s loss=%.5f%s" % (
        sys.argv[1], sys.argv[2], d["ms_per_step"], e["step_graph"], e["forced_collectives"], d["losses"]["loss"],
        ("  step_graph_error=" + str(e["step_graph_error"])[:160]) if e["step_graph_error"] else ""))
except Exception as ex:
    print("global_batch %2s %-13s: FAILED (%r) %s" % (sys.argv[1], sys.argv[2], ex, open("/tmp/fc_err.txt").read()[-600:]))
PY
  done
done
cat $OUT
