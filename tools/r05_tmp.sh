cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_seglosses_gpu.py -q -m gpu 2>&1 | tail -4
timeout 200 python tools/seglosses_bench.py 2>&1 | grep "B=24"
bash tools/nd_matrix.sh 2>&1 | tail -2
