# step time at the per-rank batches of a 2 / 4 / 8-GPU split of the global batch 24, on one GPU  (usage: bash tools/small_batch.sh [out file])
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; OUT=${1:-$R/gpurun_out/small_batch.txt}; mkdir -p "$(dirname "$OUT")"; cd $R
: > $OUT
for b in 24 12 6 3; do
  python bench.py --global_batch $b --steps 20 --warmup 5 --no_cpu_baseline --no_kernel_timing 2>/dev/null | tail -1 > /tmp/sb_$b.json
  python - $b /tmp/sb_$b.json >> $OUT <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print("global_batch %2s: %7.1f img/s  %6.2f ms/step  step_graph=%s" % (sys.argv[1], d["value"], d["ms_per_step"], d["execution"]["step_graph"]))
PY
done
cat $OUT
