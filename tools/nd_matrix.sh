cd $GRAFT_REPO_ROOT
run() { local hits=0; for rep in 1 2 3 4 5 6 7 8; do r=$(env "$@" DIAG_POISON=cpp DIAG_POISON_BYTE=0 timeout 300 python tests/diag/poison_step_diag.py 2>&1 | grep 'poisoned\|huge  ' | grep -c "identical: False"); hits=$((hits + r)); done; echo "$* -> $hits divergent of 16 runs"; }
run UCD_X=0
run UCD_BWD_LINK=0 UCD_BLOCK_LINK=0
run UCD_DDP_LATE_COPY=0
run UCD_SGD=torch
run UCD_ABN_NODE=0
