import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from ucd_amd import synth
from ucd_amd.ddp import DistributedDataParallel
from ucd_amd.run import make_optimizer
from ucd_amd.train import Trainer
import test_step_gpu as T
dev = torch.device("cuda:0")
img = synth.images(501, 2, 129); labels = synth.seg_labels(501, 2, 129, 129, range(16, 21))
out = []
modes = [bool(int(c)) for c in (sys.argv[1] if len(sys.argv) > 1 else '01')]
torch.backends.cudnn.deterministic = len(sys.argv) > 2
for shadows in modes:
    opts = T._opts(["--opt_level", "O1"]); opts.bf16_weights = shadows; opts.graph_teacher = False
    model, model_old, classes = T._build(opts, dev)
    optim = make_optimizer(opts, model)
    ddp = DistributedDataParallel(model, bf16_weights=shadows)
    tr = Trainer(ddp, model_old, device=dev, opts=opts, classes=classes)
    ddp.train()
    before = {k: v.detach().clone() for k, v in ddp.named_parameters()}
    r = tr.train_step(img, labels, optim, None)
    grads = {k: v.grad.detach().clone() for k, v in ddp.named_parameters() if v.grad is not None}
    print('no grad:', [k for k, v in ddp.named_parameters() if v.grad is None][:5])
    after = {k: v.detach().clone() for k, v in ddp.named_parameters()}
    out.append((before, grads, after, {k: v.item() for k, v in r.items()}))
(b0, g0, a0, r0), (b1, g1, a1, r1) = out
print(r0); print(r1)
names = [k for k in g0 if k.endswith("weight") and g0[k].dim() == 4][:3] + [k for k in g0 if "mod4.block3.convs.conv" in k] + [k for k in g0 if "head" in k][:6]
for k in names:
    rel = lambda x, y: ((x.float() - y.float()).norm() / (x.float().norm() + 1e-30)).item()
    print(f"{k:50s} before {rel(b0[k], b1[k]):.2e} grad {rel(g0[k], g1[k]):.2e} |g| {g0[k].norm().item():.3e} {g1[k].norm().item():.3e} after {rel(a0[k], a1[k]):.2e} |w| {b0[k].norm().item():.3e}")
