"""Where do a fresh run and a perturbed run of the first iteration part (tests/diag/poison_params_diag.py: most tensors differ after one
update, intermittently)?  Device-side checksums (no host synchronisation: a version of this script that copied tensors to the host
never saw the divergence) of the loss inputs and of the gradients arriving at the logits, the head output and the body output.
usage: python tests/diag/poison_where_diag.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from ucd_amd import abn, switches
import ucd_amd.train as TR
import test_step_gpu as T

switches.set("UCD_STAT_ATOMIC", "0")
rec = {}
def chk(t):
    t = t.detach().double()
    return torch.stack([t.sum(), t.abs().sum(), (t * t).sum()])
orig_seg, orig_con = TR.fused_seg_losses, TR.ucd_contrastive_loss
def seg(sem, sem_old, labels, *a, **k):
    if "in_sem" not in rec:
        rec["in_sem"] = chk(sem)
        if sem_old is not None:
            rec["in_sem_old"] = chk(sem_old)
        sem.register_hook(lambda g: rec.__setitem__("grad_sem", chk(g)))
    return orig_seg(sem, sem_old, labels, *a, **k)
def con(f_n, labels, l_po, f_o, *a, **k):
    if "in_x_pl" not in rec:
        rec["in_x_pl"], rec["in_l_po"], rec["in_f_o"] = chk(f_n), chk(l_po), chk(f_o)
        f_n.register_hook(lambda g: rec.__setitem__("grad_x_pl_total", chk(g)))
    return orig_con(f_n, labels, l_po, f_o, *a, **k)
TR.fused_seg_losses, TR.ucd_contrastive_loss = seg, con
def run():
    rec.clear()
    got = {}
    def probe(net, after):
        for n, p in net.named_parameters():
            got[n] = p.detach().float().cpu().clone()
    T._scheduled_steps("0", steps=1, probe=probe)
    return {k: v.cpu() for k, v in rec.items()}, got
r0, p0 = run()
node = abn._abn_node()
hits = 0
for attempt in range(int(os.environ.get("DIAG_ATTEMPTS", "40"))):
    node.poison_workspaces(0, -1)
    r1, p1 = run()
    bad = [n for n in p0 if not torch.equal(p0[n], p1[n])]
    if bad:
        hits += 1
        print(f"attempt {attempt}: {len(bad)} parameter tensors differ; cls differ: {[n for n in bad if n.startswith('cls')][:4]}; head differ: {len([n for n in bad if n.startswith('head')])}", flush=True)
        for k in sorted(r0):
            print("   %-18s %s" % (k, "same" if torch.equal(r0[k], r1[k]) else "DIFFERENT  %s vs %s" % (r0[k].tolist(), r1[k].tolist())))
        if hits >= 2:
            break
print("divergent runs:", hits)
