cd $GRAFT_REPO_ROOT; SKIP_TESTS=1 bash tools/measure_round.sh 2>&1 | tail -60
