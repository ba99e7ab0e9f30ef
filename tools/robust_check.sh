cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_robust; mkdir -p $O
for p in lw64 lw64x2 lw256; do UCD_CONV_PIPE=$p timeout 1500 python -m pytest tests/test_conv1x1_gpu.py tests/test_conv1x1_fused_gpu.py tests/test_step_gpu.py -x -q > $O/tests_$p.txt 2>&1; echo "UCD_CONV_PIPE=$p: $(tail -1 $O/tests_$p.txt)"; done
for cfg in "--dataset ade --task 100-50 --global_batch 3 --crop 512" "--dataset city --task 13-6 --global_batch 2 --crop 768" "--task 15-5s --step 2 --global_batch 3"; do timeout 400 python bench.py --steps 10 --warmup 6 --no_cpu_baseline --no_kernel_timing $cfg 2>$O/err.txt | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', 'ms_per_step %.2f' % d['ms_per_step'], d['execution']['step_graph'], d['losses'])" || tail -5 $O/err.txt; done
