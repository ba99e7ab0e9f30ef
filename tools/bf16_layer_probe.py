"""Where does the bf16 (--opt_level O1) forward of the student leave the fp32 forward of the product?  Relative L2 of every
block output / the head / the logits, train mode, 2 x 513^2, calibrated synthetic checkpoint; fused conv+ABN nodes against the
module-by-module bf16 path (UCD_FUSED_CONV1X1=0) - is the error bf16 storage, or a kernel's?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ucd_amd import argparser, switches, synth, tasks  # noqa: E402
from ucd_amd.ddp import DistributedDataParallel  # noqa: E402
from ucd_amd.run import build_models, load_step_checkpoint  # noqa: E402

dev = torch.device("cuda:0")
B, S = int(os.environ.get("PROBE_B", 2)), int(os.environ.get("PROBE_S", 513))
img = synth.images(502, B, S).to(dev).contiguous(memory_format=torch.channels_last)


def run(level, fused):
    switches.set("UCD_FUSED_CONV1X1", "1" if fused else "0")
    opts = argparser.modify_command_options(argparser.get_argparser().parse_args(
        ["--method", "UCD", "--task", "15-5", "--step", "1", "--no_pretrained", "--norm_act", "iabn_sync", "--opt_level", level]))
    classes = tasks.get_per_task_classes("voc", "15-5", 1)
    model, model_old = build_models(opts, dev, classes)
    state = synth.fill_state_dict({k: v.cpu() for k, v in model_old.state_dict().items()}, 42, calibrated=True)
    net = model
    if level != "O0":
        model = DistributedDataParallel(model, delay_allreduce=True, bf16_weights=True)
    load_step_checkpoint(opts, model, model_old, state, dev)
    model.train()
    outs = {}
    hooks = []
    for name, m in net.named_modules():
        if name.count(".") == 2 and name.startswith("body.mod") and "block" in name or name in ("body.mod1", "head", "body"):
            hooks.append(m.register_forward_hook(lambda mod, a, o, name=name: outs.__setitem__(name, o.detach().float().clone())))
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=level != "O0"):
        _, f = model(img, upsample=False)
    outs["sem"] = f["sem"].detach().float()
    for h in hooks:
        h.remove()
    return outs


ref = run("O0", True)
for fused in (True, False):
    got = run("O1", fused)
    print("---- bf16", "fused nodes" if fused else "module path", "vs fp32 product; rel L2 per stage")
    for k in ref:
        if k in got:
            e = ((got[k] - ref[k]).norm() / ref[k].norm()).item()
            m = ref[k].mean(dim=(0, 2, 3)); s = ref[k].std(dim=(0, 2, 3))
            print(f"{k:24s} {e:.4e}   |mean|/std of the fp32 map {(m.abs().mean() / s.mean()).item():.2f}")
