# A/B of one environment switch on ONE box: bench at 24 images, alternating settings.  usage: VAR=UCD_CONV_PIPE A=2x64 B=auto bash tools/switch_ab.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/switch_ab; mkdir -p $O; cd $R
for rep in 1 2; do for v in "$A" "$B"; do
  if [ "$v" = "auto" ]; then unset $VAR; else export $VAR=$v; fi
  timeout 400 python bench.py --steps 30 --warmup 6 --global_batch ${GB:-24} --no_cpu_baseline --no_kernel_timing 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v rep $rep ms_per_step %.3f img/s %.1f' % (d['ms_per_step'], d['value']))"
done; done 2>&1 | tee $O/${VAR}_${GB:-24}.txt
