cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_seglosses_gpu.py -q -m gpu -x 2>&1 | tail -8
timeout 200 python tools/seglosses_bench.py 2>&1 | grep "ucd_seg_losses"
UCD_SEG_PK=0 timeout 200 python tools/seglosses_bench.py 2>&1 | grep "ucd_seg_losses"
