# round 6: cross-step fragment prefetch of the loader-wave GEMM (UCD_CONV_LW_PF = 0 | 1 | 2): tests, probes, step time
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; O=gpurun_out/r6_pf; mkdir -p $O
python -m pytest tests/test_conv1x1_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
for pf in 0 1 2; do
  UCD_CONV_LW_PF=$pf python tools/conv3x3_probe.py > $O/conv3x3_pf$pf.txt 2>&1
  UCD_CONV_LW_PF=$pf python tools/conv_pipe_probe.py > $O/pipe_pf$pf.txt 2>&1
done
for pf in 0 1 2 0 1 2; do
  for b in 24 3; do
    UCD_CONV_LW_PF=$pf python bench.py --global_batch $b --steps 20 --warmup 5 --no_cpu_baseline --no_kernel_timing 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('pf $pf batch $b: %.2f ms/step %.1f img/s' % (d['ms_per_step'], d['value']))" >> $O/step.txt
  done
done
cat $O/step.txt
