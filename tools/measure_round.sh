R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
# every profiler / probe call under its own timeout: a hung collection must not eat the box's time limit
cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -5 > $O/gputests.log
python bench.py --steps 20 --warmup 5 > $O/bench_final.json 2> $O/bench_final.err
for gb in 12 6 3; do python bench.py --steps 10 --warmup 3 --global_batch $gb --no_cpu_baseline --no_kernel_timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('global_batch', $gb, 'ms_per_step', round(d['ms_per_step'],2), 'img/s', round(d['value'],1))"; done > $O/small_batch.txt 2>&1
# optimiser step A/B on this box: torch's fused step vs the one-launch step (csrc/sgd.hip), benchmark batch and the 8-GPU per-rank batch
(for mode in torch hip; do for gb in 24 3; do UCD_SGD=$mode timeout 120 python bench.py --steps 12 --warmup 4 --global_batch $gb --no_cpu_baseline --no_kernel_timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('UCD_SGD=$mode', 'global_batch', $gb, 'ms_per_step', round(d['ms_per_step'],3), 'img/s', round(d['value'],1))"; done; done) > $O/sgd_ab.txt 2>&1
timeout 300 python tools/conv1x1_probe.py > $O/conv1x1_probe.txt 2>&1
timeout 300 python tools/conv3x3_probe.py > $O/conv3x3_probe.txt 2>&1
timeout 200 python tools/conv_strided_probe.py > $O/conv_strided_probe.txt 2>&1
(timeout 100 python tools/stem_conv_probe.py | tail -1; timeout 100 python tools/stem_conv_probe.py 3 | tail -1) > $O/stem_conv_probe.txt 2>&1
timeout 100 python tools/blocklink_probe.py > $O/blocklink_probe.txt 2>&1
timeout 300 python tools/wgrad_probe2.py > $O/wgrad_probe.txt 2>&1
timeout 120 python tools/pixcon_pairs.py > $O/pixcon_pairs.txt 2>&1
(for m in f16 f16_split; do timeout 100 python tools/pixcon_bench.py $m | tail -2; timeout 100 python tools/pixcon_bench.py $m dom | tail -2; done) > $O/pixcon_bench.txt 2>&1
timeout 100 python tools/seglosses_bench.py > $O/seglosses_bench.txt 2>&1
timeout 100 python tools/seglosses_bench.py ade >> $O/seglosses_bench.txt 2>&1
(for sw in UCD_STEM_FOLD UCD_BLOCK_LINK UCD_OWN_WGRAD UCD_OWN_STRIDED UCD_PROJ_ALIAS UCD_OWN_STEM; do env $sw=0 timeout 120 python bench.py --steps 12 --warmup 4 --no_cpu_baseline --no_kernel_timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$sw=0', 'ms_per_step', round(d['ms_per_step'],3), 'img/s', round(d['value'],1))"; done; timeout 120 python bench.py --steps 12 --warmup 4 --no_cpu_baseline --no_kernel_timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', 'ms_per_step', round(d['ms_per_step'],3), 'img/s', round(d['value'],1))") > $O/switch_ab.txt 2>&1
timeout 200 python tools/abn_bench.py > $O/abn_bench.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof -o t --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --no_cpu_baseline --no_kernel_timing > /tmp/prof.log 2>&1
python $R/tools/trace_summary.py /tmp/prof/t_kernel_trace.csv $O/step_kernel_summary_final.txt "timeout 400 rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 2 (round-3 final)" > /dev/null
head -40 /tmp/prof/t_kernel_stats.csv > $O/kernel_stats_final.csv
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof3 -o t --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --global_batch 3 --no_cpu_baseline --no_kernel_timing > /tmp/prof3.log 2>&1
python $R/tools/trace_summary.py /tmp/prof3/t_kernel_trace.csv $O/step_kernel_summary_b3.txt "same, --global_batch 3 (per-rank batch of the 8-GPU run)" > /dev/null
timeout 400 rocprofv3 --pmc FETCH_SIZE -d /tmp/pf -o f --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_kernel_timing > /tmp/pf.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d /tmp/pw -o w --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_kernel_timing > /tmp/pw.log 2>&1
python $R/tools/pmc_to_json.py /tmp/pf/f_counter_collection.csv /tmp/pw/w_counter_collection.csv $O/pmc_bench.json 24 > $O/pmc_bench.txt 2>&1
echo "# rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -- python3 tools/pixcon_bench.py <mode> dom" > $O/pixcon_sq.txt
echo "# (B=24, 513^2 shapes, one teacher class dominating like the benchmark step); mean per dispatch; SQ_WAVE_CYCLES etc. count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles" >> $O/pixcon_sq.txt
for mode in f16 f16_split; do
rm -rf /tmp/psq
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d /tmp/psq -o s --output-format csv -- python3 $R/tools/pixcon_bench.py $mode dom > /tmp/psq.log 2>&1
echo "## precision $mode ($(grep 'median' /tmp/psq.log | tail -1))" >> $O/pixcon_sq.txt
python - <<'PY' >> $O/pixcon_sq.txt 2>&1
import csv, collections, glob
f = glob.glob('/tmp/psq/*counter_collection.csv')
agg = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
def label(k):
    if "pixcon16p_sweep" in k: return "planned sweep 1 (pixcon16p_sweep_kernel<0>)" if "ILi0E" in k else ("planned sweep 2 with probabilities (<1, true>)" if "ILi1ELb1" in k else "planned sweep 2 (<1, false>)")
    if "pixcon16p_plan" in k: return "pixcon16p_plan_kernel"
    if "pixcon16p_finalize" in k: return "pixcon16p_finalize_kernel"
    if "pixcon16_neg" in k: return "fixed-split sweep 1 (pixcon16_neg_kernel)"
    if "pixcon16_pos" in k: return "fixed-split sweep 2 (pixcon16_pos_kernel)"
    if "pixcon16_finalize" in k: return "pixcon16_finalize_kernel"
    return None
for r in csv.DictReader(open(f[0])):
    name = label(r["Kernel_Name"])
    if name is None: continue
    agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    meta[name] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Grid_Size"), r.get("Workgroup_Size"))
for name, cs in agg.items():
    print(name, "VGPR/AGPR/SGPR/LDS/grid/wg =", meta[name], "dispatches", len(next(iter(cs.values()))))
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    for c, v in sorted(m.items()): print("   %-28s %.4e" % (c, v))
    if m.get("SQ_WAVE_CYCLES"):
        print("   mfma busy / (4 * wave quad-cycles) = %.3f   lds conflict / lds active = %.3f" % (m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * m["SQ_WAVE_CYCLES"]), m.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, m.get("SQ_LDS_IDX_ACTIVE", 1))))
PY
done
python tools/kernel_resources.py > $O/kernel_resources.txt 2>&1
cd $R; cat $O/gputests.log; cat $O/sgd_ab.txt; cut -c1-250 $O/bench_final.json; cat $O/small_batch.txt; head -8 $O/step_kernel_summary_final.txt; head -5 $O/step_kernel_summary_b3.txt; cat $O/pmc_bench.txt | head -30; cat $O/pixcon_sq.txt | head -40
