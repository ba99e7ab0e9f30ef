R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; O=gpurun_out/r6_defer.txt; : > $O
python -m pytest tests/test_conv1x1_fused_gpu.py tests/test_step_gpu.py -x -q -m gpu -k "deferred" 2>&1 | tail -5
for i in 1 2 3; do for b in 24 3; do for v in 0 1; do
  UCD_WGRAD_DEFER=$v python bench.py --global_batch $b --steps 20 --warmup 5 --no_cpu_baseline --no_kernel_timing 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('defer=$v batch $b: %.2f ms/step %.1f img/s' % (d['ms_per_step'], d['value']))" | tee -a $O
done; done; done
