# round 6: the fp32 mode (--opt_level O0, the arithmetic that holds north_star's 1e-3) beside the benchmarked bf16 one
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; O=gpurun_out/r6_o0; mkdir -p $O
python bench.py --opt_level O0 --first_step_losses --steps 10 --warmup 3 --no_cpu_baseline --no_kernel_timing 2>$O/o0.err | tail -1 > $O/o0.json
python bench.py --opt_level O1 --first_step_losses --steps 20 --warmup 5 --no_cpu_baseline --no_kernel_timing 2>$O/o1.err | tail -1 > $O/o1.json
python - <<'PY'
import json
o0, o1 = json.load(open("gpurun_out/r6_o0/o0.json")), json.load(open("gpurun_out/r6_o0/o1.json"))
out = {"note": "same box, same synthetic batch and checkpoint: --opt_level O0 (fp32 activations and weights: the mode the reference goldens are held to 1e-3 in) against the benchmarked O1 (bf16)",
       "O0": {k: o0[k] for k in ("value", "ms_per_step", "dtype", "first_step_losses", "execution")},
       "O1": {k: o1[k] for k in ("value", "ms_per_step", "dtype", "first_step_losses", "execution")}}
f0, f1 = o0["first_step_losses"], o1["first_step_losses"]
out["first_step_losses_rel_O1_vs_O0"] = {k: (f1[k] - f0[k]) / abs(f0[k]) if f0.get(k) else None for k in f0}
json.dump(out, open("gpurun_out/r6_o0/bench_o0.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
PY
