"""GPU probe: spread of the bf16 (--opt_level O1) step's losses around the fp32 (O0) step of the product, over input seeds and
with / without the fused conv+ABN nodes (UCD_FUSED_CONV1X1=0: module path).  2 x 513^2, deterministic solvers.
usage: python tools/bf16_bias_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from ucd_amd import switches, synth
from ucd_amd.run import make_optimizer
from ucd_amd.train import Trainer
import test_step_gpu as T

dev = torch.device("cuda:0")
torch.backends.cudnn.deterministic = True
for seed in (502, 503, 504):
    img = synth.images(seed, 2, 513)
    labels = synth.seg_labels(seed, 2, 513, 513, range(16, 21))
    res = {}
    for tag, lvl, env in (("O0", "O0", None), ("O1 fused", "O1", None), ("O1 module", "O1", "0")):
        if env is not None:
            switches.set("UCD_FUSED_CONV1X1", env)
        try:
            opts = T._opts(["--opt_level", lvl])
            model, model_old, classes = T._build(opts, dev)
            trainer = Trainer(model, model_old, device=dev, opts=opts, classes=classes)
            optim = make_optimizer(opts, model)
            model.train()
            res[tag] = {k: v.item() for k, v in trainer.train_step(img, labels, optim, None).items()}
        finally:
            switches.unset("UCD_FUSED_CONV1X1")
    for tag in ("O1 fused", "O1 module"):
        print("seed", seed, tag, {k: "%+.2f%%" % (100 * (res[tag][k] - res["O0"][k]) / abs(res["O0"][k])) for k in ("ce", "con", "lkd", "loss")})
