# Final check of a round on one box: the GPU suite, smoke(), the default bench line.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/final; mkdir -p $O
cd $R
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -30 > $O/tests_gpu.txt
tail -4 $O/tests_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -2
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json | head -c 400; echo
python -c "import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('bench', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms', d['execution'])"
