"""GPU diagnostic (not part of the product): per-parameter gradient comparison of the product's fp32 UCD
step against the CPU oracle on identical inputs, in backward order, to localise a divergence."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ucd_amd import argparser, synth
from ucd_amd.run import build_models, load_step_checkpoint
from ucd_amd.train import Trainer
from oracle import step as OS, model as OM
from oracle.params import template_state

S = int(sys.argv[1]) if len(sys.argv) > 1 else 129
if len(sys.argv) > 2 and sys.argv[2] == "nomiopen":
    torch.backends.cudnn.enabled = False      # native (exact fp32) convolution kernels instead of MIOpen
dev = torch.device("cuda:0")
opts = argparser.modify_command_options(argparser.get_argparser().parse_args(
    ["--method", "UCD", "--task", "15-5", "--step", "1", "--lr", "0.001", "--no_pretrained"]))
classes = [16, 5]
model, model_old = build_models(opts, dev, classes)
state = synth.fill_state_dict({k: v.cpu() for k, v in model_old.state_dict().items()}, 42)
load_step_checkpoint(opts, model, model_old, state, dev)
trainer = Trainer(model, model_old, device=dev, opts=opts, classes=classes)
optim = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=0.0)
img = synth.images(501, 2, S); labels = synth.seg_labels(501, 2, S, S, range(16, 21))
model.train()
r = trainer.train_step(img, labels, optim, None)
torch.cuda.synchronize()
print({k: float(v) for k, v in r.items()})
# oracle
Pt = OS.make_params(state, requires_grad=False)
st = template_state(classes); st.update({k: v.clone() for k, v in state.items()})
Ps = OS.make_params(st)
OM.init_new_classifier(Ps, 2, 5)
torch.set_num_threads(32)
t0 = time.time()
o = OS.ucd_losses(Ps, Pt, img, labels, classes)
(o["loss"] + o["lkd"]).backward()
print("oracle", {k: float(o[k]) for k in ("loss", "lkd", "ce", "con")}, "t=%.1fs" % (time.time() - t0))
params = dict(model.named_parameters())
rows = []
for n, p in params.items():
    if p.grad is None or Ps[n].grad is None:
        continue
    g, gr = p.grad.cpu().double(), Ps[n].grad.double()
    rel = (g - gr).norm().item() / max(gr.norm().item(), 1e-30)
    rows.append((n, rel, gr.norm().item()))
for n, rel, nrm in reversed(rows):
    flag = " <<<" if rel > 1e-2 else ""
    print(f"{rel:10.3e}  {nrm:10.3e}  {n}{flag}")
