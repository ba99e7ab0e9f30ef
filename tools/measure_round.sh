R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -5 > $O/gputests.log
python bench.py --steps 20 --warmup 5 > $O/bench_final.json 2> $O/bench_final.err
for gb in 12 6 3; do python bench.py --steps 10 --warmup 3 --global_batch $gb --no_cpu_baseline --no_kernel_timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('global_batch', $gb, 'ms_per_step', round(d['ms_per_step'],2), 'img/s', round(d['value'],1))"; done > $O/small_batch.txt 2>&1
python tools/conv1x1_probe.py > $O/conv1x1_probe.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof -o t --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --no_cpu_baseline --no_kernel_timing > /tmp/prof.log 2>&1
python $R/tools/trace_summary.py /tmp/prof/t_kernel_trace.csv $O/step_kernel_summary_final.txt "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 2 (round-2 final)" > /dev/null
head -40 /tmp/prof/t_kernel_stats.csv > $O/kernel_stats_final.csv
rocprofv3 --kernel-trace --stats -d /tmp/prof3 -o t --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --global_batch 3 --no_cpu_baseline --no_kernel_timing > /tmp/prof3.log 2>&1
python $R/tools/trace_summary.py /tmp/prof3/t_kernel_trace.csv $O/step_kernel_summary_b3.txt "same, --global_batch 3 (per-rank batch of the 8-GPU run)" > /dev/null
rocprofv3 --pmc FETCH_SIZE -d /tmp/pf -o f --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_kernel_timing > /tmp/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/pw -o w --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_kernel_timing > /tmp/pw.log 2>&1
python $R/tools/pmc_to_json.py /tmp/pf/f_counter_collection.csv /tmp/pw/w_counter_collection.csv $O/pmc_bench.json 24 > $O/pmc_bench.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d /tmp/psq -o s --output-format csv -- python3 $R/tools/pixcon_bench.py f16 > /tmp/psq.log 2>&1
python - <<'PY' > $O/pixcon_sq.txt 2>&1
import csv, collections, glob
f = glob.glob('/tmp/psq/*counter_collection.csv')
agg = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "pixcon16" not in k: continue
    name = "pixcon16_neg_kernel" if "neg" in k else ("pixcon16_pos_kernel" if "pos" in k else "pixcon16_finalize_kernel")
    agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    meta[name] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Grid_Size"), r.get("Workgroup_Size"))
print("# rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -- python3 tools/pixcon_bench.py f16")
print("# (B=24, 513^2 shapes: A ~ 19.6k anchors, C ~ 35k contrast rows); mean per dispatch; SQ_WAVE_CYCLES etc. count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles")
for name, cs in agg.items():
    print(name, "VGPR/AGPR/SGPR/LDS/grid/wg =", meta[name], "dispatches", len(next(iter(cs.values()))))
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    for c, v in sorted(m.items()): print("   %-28s %.4e" % (c, v))
    if "SQ_WAVE_CYCLES" in m and m["SQ_WAVE_CYCLES"]:
        print("   mfma busy / (4 * wave quad-cycles) = %.3f   lds conflict / lds active = %.3f" % (m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * m["SQ_WAVE_CYCLES"]), m.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, m.get("SQ_LDS_IDX_ACTIVE", 1))))
PY
tail -3 /tmp/psq.log >> $O/pixcon_sq.txt
cd $R; cat $O/gputests.log; cut -c1-250 $O/bench_final.json; cat $O/small_batch.txt; head -8 $O/step_kernel_summary_final.txt; head -5 $O/step_kernel_summary_b3.txt; cat $O/pmc_bench.txt | head -30; cat $O/pixcon_sq.txt | head -40
