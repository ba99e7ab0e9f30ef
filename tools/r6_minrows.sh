R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
for i in 1 2; do for b in 3 6; do for v in 8192 0; do
  UCD_CONV3_MIN_ROWS=$v python bench.py --global_batch $b --steps 20 --warmup 5 --no_cpu_baseline --no_kernel_timing 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('conv3_min_rows=$v batch $b: %.2f ms/step' % d['ms_per_step'])"
done; done; done
