# side stream of the weight gradients: off / groups of G calls per fork, at the per-rank batches
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
run() { env "$@" python bench.py --global_batch $GB --steps 20 --warmup 5 --no_cpu_baseline --no_kernel_timing 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('batch $GB $*: %.2f ms/step' % d['ms_per_step'])"; }
for GB in ${BATCHES:-3 6 24}; do for rep in 1 2; do
  run UCD_WGRAD_STREAM=0
  for g in ${GROUPS_:-1 4 8 16 32}; do run UCD_WGRAD_STREAM=1 UCD_WGRAD_STREAM_GROUP=$g; done
done; done
