R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; O=gpurun_out/r6_lw64; mkdir -p $O
python -m pytest tests/test_conv1x1_fused_gpu.py tests/test_conv1x1_gpu.py -x -q -m gpu -k "atomic or dilated or exact or small or three_images or bn64" 2>&1 | tail -5
python tools/lw_probe.py 3 > $O/lw64_on.txt 2>&1
UCD_CONV_LW64_TILES=0 python tools/lw_probe.py 3 > $O/lw64_off.txt 2>&1
paste -d'\n' $O/lw64_off.txt $O/lw64_on.txt | grep -v amdgpu
for i in 1 2; do for v in 0 128; do
  UCD_CONV_LW64_TILES=$v python bench.py --global_batch 3 --steps 20 --warmup 5 --no_cpu_baseline --no_kernel_timing 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('lw64 tiles<=$v batch 3: %.2f ms/step' % d['ms_per_step'])" | tee -a $O/step.txt
done; done
