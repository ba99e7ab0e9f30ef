"""Which parameters differ after ONE iteration between a fresh run and a run behind scratch-buffer fills (tests/diag/poison_step_diag.py
found intermittent differences from iteration 2 on)?  usage: python tests/diag/poison_params_diag.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
from ucd_amd import abn, hip, switches
import test_step_gpu as T

switches.set("UCD_STAT_ATOMIC", "0")
def run():
    got = {}
    def probe(net, after):
        for n, p in net.named_parameters():
            got[n] = p.detach().float().cpu().clone()
        for n, b in net.named_buffers():
            if b.is_floating_point():
                got["buf:" + n] = b.detach().float().cpu().clone()
    l, _, _, _, _ = T._scheduled_steps("0", steps=int(os.environ.get("DIAG_STEPS", "1")), probe=probe)
    return l, got
l0, p0 = run()
node = abn._abn_node()
for attempt in range(8):
    node.poison_workspaces(0, -1)
    l1, p1 = run()
    bad = [(n, float((p1[n] - p0[n]).norm() / (p0[n].norm() + 1e-30))) for n in p0 if not torch.equal(p0[n], p1[n])]
    print(f"attempt {attempt}: losses equal {np.array_equal(l0, l1)}; {len(bad)} of {len(p0)} tensors differ", flush=True)
    if bad:
        bad.sort(key=lambda t: -t[1])
        for n, d in bad[:25]:
            print("   %-70s %.3e" % (n, d))
        names = dict(bad)
        print("   cls / head tensors:", [(n, "%.2e" % names[n] if n in names else "same") for n in p0 if n.startswith(("cls", "head"))])
        print("   buffers that differ:", [n for n in names if n.startswith("buf:")][:8])
        order = [n for n in p0]
        first = min(order.index(n) for n, _ in bad)
        print("first differing tensor in registration order:", order[first], " last:", order[max(order.index(n) for n, _ in bad)])
        break
