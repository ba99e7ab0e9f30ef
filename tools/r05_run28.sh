cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for b in 0 127 63; do
echo "== rep $rep byte $b: $(DIAG_POISON=cpp DIAG_POISON_BYTE=$b timeout 600 python tests/diag/poison_step_diag.py 2>&1 | grep 'poisoned\|huge  ' | sed 's/.*\] identical: \([A-Za-z]*\) *\(nan: [A-Za-z]*\)*.*/\1 \2/' | tr '\n' ' ')"
done; done
