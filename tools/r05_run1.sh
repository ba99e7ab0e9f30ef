cd $GRAFT_REPO_ROOT; O=gpurun_out/r05a; mkdir -p $O
timeout 900 python -m pytest tests/test_conv1x1_fused_gpu.py -x -q -m gpu -k "resident_a" > $O/test_ra.txt 2>&1; tail -15 $O/test_ra.txt
for ra in 1 0; do UCD_CONV_RA=$ra timeout 300 python tools/conv_ra_probe.py 2>&1 | grep -v amdgpu.ids; done > $O/conv_ra_probe.txt 2>&1; cat $O/conv_ra_probe.txt
for ra in 0 1 0 1; do UCD_CONV_RA=$ra timeout 600 python bench.py --steps 20 --warmup 6 --no_cpu_baseline --no_kernel_timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('UCD_CONV_RA=$ra', 'ms_per_step', round(d['ms_per_step'],3), 'img/s', round(d['value'],1), d['losses'])"; done > $O/bench_ab.txt 2>&1; cat $O/bench_ab.txt
