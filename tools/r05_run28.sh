cd $GRAFT_REPO_ROOT
for ov in 0 1; do for rep in 1 2 3 4 5; do
echo "== overlap $ov rep $rep: $(UCD_TEACHER_OVERLAP=$ov DIAG_POISON=cpp DIAG_POISON_BYTE=0 timeout 600 python tests/diag/poison_step_diag.py 2>&1 | grep 'poisoned\|huge  ' | sed 's/.*\] identical: \([A-Za-z]*\).*/\1/' | tr '\n' ' ')"
done; done
