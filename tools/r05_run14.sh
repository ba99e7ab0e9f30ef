cd $GRAFT_REPO_ROOT; O=gpurun_out/r05c; mkdir -p $O
timeout 1500 python -m pytest tests/test_ddp_gpu.py -q -m gpu 2>&1 | tail -8
timeout 600 python bench.py --steps 10 --warmup 6 > $O/bench_full.json 2> $O/bench_full.err; python -c "
import json; d=json.loads(open('$O/bench_full.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline'], {k:(round(v['frac'],3), round(v['ms_total'],2)) for k,v in d['kernels'].items()})"
(for gb in 3 6 12; do timeout 400 python bench.py --force_dist --steps 16 --warmup 6 --global_batch $gb --no_cpu_baseline --no_kernel_timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('--force_dist global_batch $gb', 'ms_per_step', round(d['ms_per_step'],3), 'eager_ms', round(d['execution']['eager_ms'],3), 'graph_ms', d['execution']['graph_ms'], d['execution']['step_graph_error'])"; done) > $O/forced_collectives.txt 2>&1; cat $O/forced_collectives.txt
