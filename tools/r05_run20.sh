cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05d
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05d/bench.json 2> gpurun_out/r05d/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05d/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], d["execution"]["step_graph"], d["losses"])
k=d["kernels"]["ucd_seg_losses"]; print("seg", k)
PY
timeout 1500 python -m pytest tests/test_step_gpu.py tests/test_seglosses_gpu.py -q -m gpu -x 2>&1 | tail -5
