# Round measurement set -> gpurun_out/r06/ (copied to profiles/r06_* by hand).  Every profiler / probe call under its own timeout.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r06; mkdir -p $O
cd $R
if [ -z "$SKIP_TESTS" ]; then timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $O/gputests.log; fi
timeout 900 python bench.py --steps 20 --warmup 6 > $O/bench_final.json 2> $O/bench_final.err
brief() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms_per_step', round(d['ms_per_step'],3), 'img/s', round(d['value'],1), 'step_graph', d['execution']['step_graph'])"; }
# the per-rank batches of the multi-GPU split on one GPU, with the captured step and without
(for gb in 24 12 6 3; do for sg in auto 0; do UCD_STEP_GRAPH=$sg timeout 300 python bench.py --steps 16 --warmup 6 --global_batch $gb --no_cpu_baseline --no_kernel_timing 2>/dev/null | brief "global_batch $gb UCD_STEP_GRAPH=$sg"; done; done) > $O/small_batch.txt 2>&1
# A/B of this round's switches on this box (alternating, two repetitions)
(for rep in 1 2; do
  for gb in 24 6 3; do for v in 0 1; do UCD_WGRAD_STREAM=$v timeout 300 python bench.py --steps 20 --warmup 6 --global_batch $gb --no_cpu_baseline --no_kernel_timing 2>/dev/null | brief "rep $rep $gb images UCD_WGRAD_STREAM=$v"; done; done
  for v in 0 1; do UCD_WGRAD_DEFER=$v timeout 300 python bench.py --steps 20 --warmup 6 --no_cpu_baseline --no_kernel_timing 2>/dev/null | brief "rep $rep 24 images UCD_WGRAD_DEFER=$v"; done
  for v in 0 1; do UCD_WGRAD_DEFER=$v timeout 300 python bench.py --steps 20 --warmup 6 --global_batch 3 --no_cpu_baseline --no_kernel_timing 2>/dev/null | brief "rep $rep 3 images UCD_WGRAD_DEFER=$v"; done
  for v in 0 128; do UCD_CONV_LW64_TILES=$v timeout 300 python bench.py --steps 20 --warmup 6 --global_batch 3 --no_cpu_baseline --no_kernel_timing 2>/dev/null | brief "rep $rep 3 images UCD_CONV_LW64_TILES=$v"; done
  for v in 8192 0; do UCD_CONV3_MIN_ROWS=$v timeout 300 python bench.py --steps 20 --warmup 6 --global_batch 3 --no_cpu_baseline --no_kernel_timing 2>/dev/null | brief "rep $rep 3 images UCD_CONV3_MIN_ROWS=$v"; done
  for v in 8192 0; do UCD_CONV3_MIN_ROWS=$v timeout 300 python bench.py --steps 20 --warmup 6 --global_batch 6 --no_cpu_baseline --no_kernel_timing 2>/dev/null | brief "rep $rep 6 images UCD_CONV3_MIN_ROWS=$v"; done
done) > $O/kernel_ab.txt 2>&1
# the multi-rank step with its collectives, as far as one GPU can run it: eager first, then captured (bench.py --force_dist)
(for gb in 3 6 12; do timeout 400 python bench.py --force_dist --steps 16 --warmup 6 --global_batch $gb --no_cpu_baseline --no_kernel_timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('--force_dist global_batch $gb', 'ms_per_step', round(d['ms_per_step'],3), 'eager_ms', d['execution']['eager_ms'], 'graph_ms', d['execution']['graph_ms'], d['execution']['step_graph_error'])"; done) > $O/forced_collectives.txt 2>&1
timeout 300 python tools/conv1x1_probe.py > $O/conv1x1_probe.txt 2>&1
timeout 300 python tools/conv3x3_probe.py > $O/conv3x3_probe.txt 2>&1
timeout 300 python tools/lw_probe.py 3,6,24 > $O/lw_probe.txt 2>&1
(echo "## UCD_WGRAD3=0 (9-tap form)"; UCD_WGRAD3=0 timeout 300 python tools/wgrad_probe2.py 2>&1 | grep -v amdgpu.ids; echo "## default (three-tap form for the 3x3 layers)"; timeout 300 python tools/wgrad_probe2.py 2>&1 | grep -v amdgpu.ids) > $O/wgrad_probe.txt 2>&1
timeout 100 python tools/blocklink_probe.py > $O/blocklink_probe.txt 2>&1
timeout 200 python tools/abn_bench.py > $O/abn_bench.txt 2>&1
(for m in f16 f16_split; do timeout 100 python tools/pixcon_bench.py $m | tail -2; timeout 100 python tools/pixcon_bench.py $m dom | tail -2; done) > $O/pixcon_bench.txt 2>&1
python tools/kernel_resources.py > $O/kernel_resources.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof -o t --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --no_cpu_baseline --no_kernel_timing > /tmp/prof.log 2>&1
python $R/tools/trace_summary.py /tmp/prof/t_kernel_trace.csv $O/step_kernel_summary_final.txt "timeout 400 rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 2 (round 6; the timed steps are replays of the captured step graph)" > /dev/null
head -40 /tmp/prof/t_kernel_stats.csv > $O/kernel_stats_final.csv
# the same with nothing overlapping (no side stream for the weight gradients, teacher in front of the student): the durations of the
# kernels running ALONE - what bench.py's per-call table and the roofline are measured on
export UCD_WGRAD_STREAM=0 UCD_TEACHER_OVERLAP=0
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/profs -o t --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --no_cpu_baseline --no_kernel_timing > /tmp/profs.log 2>&1
unset UCD_WGRAD_STREAM UCD_TEACHER_OVERLAP
python $R/tools/trace_summary.py /tmp/profs/t_kernel_trace.csv $O/step_kernel_summary_serial.txt "UCD_WGRAD_STREAM=0 UCD_TEACHER_OVERLAP=0 rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 2 (nothing overlaps: kernel durations alone)" > /dev/null
head -40 /tmp/profs/t_kernel_stats.csv > $O/kernel_stats_serial.csv
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof3 -o t --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --global_batch 3 --no_cpu_baseline --no_kernel_timing > /tmp/prof3.log 2>&1
python $R/tools/trace_summary.py /tmp/prof3/t_kernel_trace.csv $O/step_kernel_summary_b3.txt "same, --global_batch 3 (per-rank batch of the 8-GPU run)" > /dev/null
timeout 400 rocprofv3 --pmc FETCH_SIZE -d /tmp/pf -o f --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_kernel_timing > /tmp/pf.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d /tmp/pw -o w --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_kernel_timing > /tmp/pw.log 2>&1
python $R/tools/pmc_to_json.py /tmp/pf/f_counter_collection.csv /tmp/pw/w_counter_collection.csv $O/pmc_bench.json 24 > $O/pmc_bench.txt 2>&1
cd $R
# SQ counters of the round's new kernels (one pass each)
bash tools/prof_kernel.sh conv_lw_kernel -- python3 tools/conv3x3_probe.py > /dev/null 2>&1; cp gpurun_out/prof_kernel/conv_lw_kernel.txt $O/conv3x3_sq.txt 2>/dev/null
bash tools/prof_kernel.sh wgrad3_kernel -- python3 tools/wgrad_probe2.py > /dev/null 2>&1; cp gpurun_out/prof_kernel/wgrad3_kernel.txt $O/wgrad3_sq.txt 2>/dev/null
cat $O/gputests.log; cut -c1-400 $O/bench_final.json; cat $O/small_batch.txt $O/kernel_ab.txt $O/forced_collectives.txt; head -6 $O/step_kernel_summary_final.txt; head -4 $O/step_kernel_summary_b3.txt; head -12 $O/pmc_bench.txt
