R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
for i in 1 2; do for sg in auto 0; do for ov in 1 0; do
  UCD_STEP_GRAPH=$sg UCD_TEACHER_OVERLAP=$ov python bench.py --steps 20 --warmup 6 --no_cpu_baseline --no_kernel_timing 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('step_graph=$sg teacher_overlap=$ov: %.2f ms/step %.1f img/s graph=%s' % (d['ms_per_step'], d['value'], d['execution']['step_graph']))"
done; done; done
