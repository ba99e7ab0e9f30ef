cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05e
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -30 > gpurun_out/r05e/tests_gpu.txt
tail -6 gpurun_out/r05e/tests_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3
