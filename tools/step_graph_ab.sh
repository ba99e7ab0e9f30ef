# whole-step graph A/B: the two new step tests, then bench at 24 and 3 images with and without the graph -> gpurun_out/r04a/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r04a; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_step_gpu.py -x -q -s -k "whole_step_graph or twenty_step" > $O/tests.txt 2>&1; tail -30 $O/tests.txt
for gb in 24 3; do for sg in auto 0; do
  UCD_STEP_GRAPH=$sg timeout 400 python bench.py --steps 20 --warmup 6 --global_batch $gb --no_cpu_baseline --no_kernel_timing > $O/bench_b${gb}_sg$sg.json 2> $O/bench_b${gb}_sg$sg.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_b${gb}_sg$sg.json").read().strip().splitlines()[-1])
    print("global_batch $gb UCD_STEP_GRAPH=$sg ms_per_step %.3f img/s %.1f" % (d["ms_per_step"], d["value"]), d["execution"], d["losses"])
except Exception as e:
    print("global_batch $gb UCD_STEP_GRAPH=$sg FAILED", e); print(open("$O/bench_b${gb}_sg$sg.err").read()[-1500:])
PY
done; done 2>&1 | tee $O/summary.txt
