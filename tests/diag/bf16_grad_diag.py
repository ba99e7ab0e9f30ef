"""Which switch moves the bf16 step's gradient abs-sums away from the fp32 golden?  (2 x 513^2, calibrated checkpoint)
usage: python tests/diag/bf16_grad_diag.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden  # noqa: E402
from ucd_amd import argparser, synth, tasks  # noqa: E402


def run(env):
    for k in ("UCD_BLOCK_LINK", "UCD_OWN_WGRAD", "UCD_BWD_LINK", "UCD_FUSED_CONV1X1"):
        os.environ.pop(k, None)
    os.environ.update(env)
    from ucd_amd.ddp import DistributedDataParallel
    from ucd_amd.run import build_models, load_step_checkpoint, make_optimizer
    from ucd_amd.train import Trainer
    g = load_golden("ucd_step_513_cal.npz")
    dev = torch.device("cuda:0")
    opts = argparser.modify_command_options(argparser.get_argparser().parse_args(
        ["--method", "UCD", "--task", "15-5", "--step", "1", "--lr", "0.001", "--no_pretrained", "--norm_act", "iabn_sync", "--opt_level", "O1"]))
    classes = tasks.get_per_task_classes("voc", "15-5", 1)
    model, model_old = build_models(opts, dev, classes)
    state = synth.fill_state_dict({k: v.cpu() for k, v in model_old.state_dict().items()}, 42, calibrated=True)
    optim = make_optimizer(opts, model)
    net = model
    model = DistributedDataParallel(model, delay_allreduce=True, bf16_weights=True)
    load_step_checkpoint(opts, model, model_old, state, dev)
    trainer = Trainer(model, model_old, device=dev, opts=opts, classes=classes)
    model.train()
    trainer.train_step(synth.images(502, 2, 513), synth.seg_labels(502, 2, 513, 513, range(16, 21)), optim, None)
    torch.cuda.synchronize()
    params = dict(net.named_parameters())
    out = []
    for k in g:
        if k.startswith("grad_abs::"):
            n = k.split("::")[1]
            out.append((n, params[n].grad.double().abs().sum().item() / float(g[k])))
    return out


for env in ({}, {"UCD_BLOCK_LINK": "0"}, {"UCD_OWN_WGRAD": "0"}, {"UCD_BWD_LINK": "0"}, {"UCD_FUSED_CONV1X1": "0"}):
    r = run(env)
    print(env or "default", " ".join("%s=%.3f" % (n.replace("body.", "").replace(".convs", "").replace(".weight", ".w"), v) for n, v in r))
