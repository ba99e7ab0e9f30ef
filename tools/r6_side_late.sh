# side stream: launch a group at the call that fills it (LATE=0) or one call later, behind a fork point recorded earlier (LATE=1)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
run() { env "$@" python bench.py --global_batch $GB --steps 20 --warmup 5 --no_cpu_baseline --no_kernel_timing 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('batch $GB $*: %.2f ms/step' % d['ms_per_step'])"; }
for GB in ${BATCHES:-3 24}; do for rep in 1 2; do
  for g in ${GROUPS_:-1 4 32}; do for l in ${LATES:-0 1}; do run UCD_WGRAD_STREAM_GROUP=$g UCD_WGRAD_STREAM_LATE=$l; done; done
done; done
