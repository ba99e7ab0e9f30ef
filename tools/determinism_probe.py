"""Which module's forward output is not bit-reproducible run to run?  (python tools/determinism_probe.py on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from ucd_amd import synth
import test_step_gpu as T

dev = torch.device("cuda:0")
opts = T._opts()
model, model_old, classes = T._build(opts, dev)
size = int(sys.argv[1]) if len(sys.argv) > 1 else 65
if "cl" in sys.argv:
    model = model.to(memory_format=torch.channels_last)
img = synth.images(500, 2, size).to(dev)
model.train()
rec = []
def hook(name):
    def f(mod, inp, out):
        if torch.is_tensor(out):
            x = inp[0] if inp and torch.is_tensor(inp[0]) else None
            rec.append((name, type(mod).__name__, out.detach().float().cpu().clone(),
                        None if x is None else x.detach().float().cpu().clone()))
    return f
for n, m in model.named_modules():
    if not list(m.children()):
        m.register_forward_hook(hook(n))
runs = []
for r in range(3):
    rec.clear()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled="bf16" in sys.argv):
        model(img.clone())
    runs.append(list(rec))
for r in (1, 2):
    bad = 0
    for (n, t, o0, i0), (_, _, o1, i1) in zip(runs[0], runs[r]):
        same_out = torch.equal(o0, o1)
        same_in = i0 is None or torch.equal(i0, i1)
        if not same_out and same_in:
            print(f"run{r}: {n} ({t}) same input, different output; max abs diff {(o0 - o1).abs().max().item():.3e} shape {tuple(o0.shape)}")
            bad += 1
            if bad > 6:
                break
    print(f"run{r}: {bad} modules with non-reproducible output on identical input")
