"""Is the contrastive loss (forward + backward, fp16 performance mode) bit-reproducible when other work perturbs the chip between two
calls on identical inputs?  usage: python tests/diag/pixcon_repro_diag.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from ucd_amd import synth
from ucd_amd.contrastive import ucd_contrastive_loss
dev = torch.device("cuda:0")
B, N, h, K, H = int(os.environ.get("DIAG_B", "3")), 256, 33, 16, 257
g = torch.Generator(dev).manual_seed(3)
f_n = torch.randn(B, N, h, h, device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
f_o = torch.randn(B, N, h, h, device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
l_po = (torch.randn(B, K, h, h, device=dev, generator=g) * 4).bfloat16().contiguous(memory_format=torch.channels_last)
labels = synth.seg_labels(7, B, H, H, range(K, 21)).to(dev)
def once(prec):
    x = f_n.clone().requires_grad_(True)
    loss = ucd_contrastive_loss(x, labels, l_po, f_o, 0.07, 20, prec)
    loss.backward()
    torch.cuda.synchronize()
    return loss.item(), x.grad.float().clone()
junk = torch.empty(1 << 26, device=dev)
for prec in ("f16", "f32"):
    l0, g0 = once(prec)
    same = 0
    for i in range(int(os.environ.get("DIAG_REPS", "300"))):
        if i % 2:
            junk.normal_(); (junk[: 1 << (14 + i)] * 2).sum()        # other kernels of varying length in front
            t = torch.empty((1 << 20) + 7 * i, device=dev)           # and a shifted allocator state
        l1, g1 = once(prec)
        same += int(l1 == l0 and torch.equal(g0, g1))
        if not torch.equal(g0, g1):
            print(prec, "call", i, "differs: loss", l0, l1, "grad rel", ((g1 - g0).norm() / g0.norm()).item(), flush=True)
    print(prec, f"{same} of the repeated calls bit-identical to the first", flush=True)
