# How often does a second / third run of three iterations (tests/diag/poison_step_diag.py, no poisoning) part from the first of its process?
# usage: bash tools/nd_matrix.sh [ENV=VALUE ...]     (16 runs per line; before the fixed-point logit gradient of round 5: 2 of 16)
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
run() { local hits=0; for rep in 1 2 3 4 5 6 7 8; do r=$(env DIAG_POISON=none "$@" timeout 300 python tests/diag/poison_step_diag.py 2>&1 | grep 'poisoned\|huge  ' | grep -c "identical: False"); hits=$((hits + r)); done; echo "$* -> $hits divergent of 16 runs"; }
run UCD_X=1
run UCD_X=2
run UCD_SEG_PK=0
