"""Where do a fresh run and a perturbed run of the first iteration part (tests/diag/poison_params_diag.py: 323 of 543 tensors differ after
one update, intermittently)?  The inputs and the input gradients of the two loss calls are recorded in both runs and compared.
usage: python tests/diag/poison_where_diag.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
from ucd_amd import abn, switches
import ucd_amd.train as TR
import test_step_gpu as T

switches.set("UCD_STAT_ATOMIC", "0")
rec = {}
orig_seg, orig_con = TR.fused_seg_losses, TR.ucd_contrastive_loss
def seg(sem, sem_old, labels, *a, **k):
    if "seg_sem" not in rec:
        rec["seg_sem"] = sem.detach().float().cpu().clone()
        rec["seg_sem_old"] = None if sem_old is None else sem_old.detach().float().cpu().clone()
        sem.register_hook(lambda g: rec.__setitem__("seg_dsem", g.detach().float().cpu().clone()))
    return orig_seg(sem, sem_old, labels, *a, **k)
def con(f_n, labels, l_po, f_o, *a, **k):
    if "con_fn" not in rec:
        rec["con_fn"] = f_n.detach().float().cpu().clone()
        rec["con_lpo"] = l_po.detach().float().cpu().clone()
        rec["con_fo"] = f_o.detach().float().cpu().clone()
        f_n.register_hook(lambda g: rec.__setitem__("con_dfn", g.detach().float().cpu().clone()))
    return orig_con(f_n, labels, l_po, f_o, *a, **k)
TR.fused_seg_losses, TR.ucd_contrastive_loss = seg, con
def run():
    rec.clear()
    got = {}
    def probe(net, after):
        for n, p in net.named_parameters():
            got[n] = p.detach().float().cpu().clone()
    T._scheduled_steps("0", steps=1, probe=probe)
    return dict(rec), got
r0, p0 = run()
node = abn._abn_node()
for attempt in range(int(os.environ.get("DIAG_ATTEMPTS", "40"))):
    node.poison_workspaces(0, -1)
    r1, p1 = run()
    nbad = sum(int(not torch.equal(p0[n], p1[n])) for n in p0)
    print(f"attempt {attempt}: {nbad} parameter tensors differ after the update", flush=True)
    if nbad:
        for k in ("seg_sem", "seg_sem_old", "con_fn", "con_lpo", "con_fo", "seg_dsem", "con_dfn"):
            a, b = r0.get(k), r1.get(k)
            if a is None or b is None:
                print("   %-12s missing" % k); continue
            print("   %-12s %s  rel %.3e  elements differing %d of %d" % (k, "same" if torch.equal(a, b) else "DIFFERENT",
                  float((a - b).norm() / (a.norm() + 1e-30)), int((a != b).sum()), a.numel()))
        break
