# conv tests + bench at 3 and 24 images (no kernel timing / cpu baseline) -> gpurun_out/r04_quick/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r04_quick; mkdir -p $O; cd $R
[ -n "$SKIP_CONV_TESTS" ] || timeout 900 python -m pytest tests/test_conv1x1_gpu.py tests/test_conv1x1_fused_gpu.py -x -q > $O/tests_conv.txt 2>&1; tail -3 $O/tests_conv.txt
for gb in ${GBS:-3 24}; do
  timeout 400 python bench.py --steps 20 --warmup 6 --global_batch $gb --no_cpu_baseline --no_kernel_timing > $O/bench_b${gb}.json 2> $O/bench_b${gb}.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_b${gb}.json").read().strip().splitlines()[-1])
    print("global_batch $gb ms_per_step %.3f img/s %.1f" % (d["ms_per_step"], d["value"]), d["execution"], d["losses"])
except Exception as e:
    print("global_batch $gb FAILED", e); print(open("$O/bench_b${gb}.err").read()[-1500:])
PY
done 2>&1 | tee $O/summary.txt
