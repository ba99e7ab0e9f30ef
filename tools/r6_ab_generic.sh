# alternating same-box A/B of one environment switch: bash tools/r6_ab_generic.sh VAR A B [batch] [reps]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; V=$1; A=$2; B=$3; BATCH=${4:-24}; N=${5:-3}
for i in $(seq 1 $N); do for v in $A $B; do
  env $V=$v python bench.py --global_batch $BATCH --steps 20 --warmup 5 --no_cpu_baseline --no_kernel_timing 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$V=$v batch $BATCH: %.2f ms/step %.1f img/s peak %s GB' % (d['ms_per_step'], d['value'], d.get('memory', {}).get('peak_allocated_gb')))"
done; done
