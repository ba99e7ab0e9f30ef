import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ucd_amd import synth
from ucd_amd.contrastive import pixcon_loss_raw, pixcon_prepare
dev = torch.device("cuda:0")
for (B, N, h, K, H, new_ids, max_label) in [(2, 64, 12, 101, 192, list(range(101, 151)), 150), (3, 256, 33, 16, 513, list(range(16, 21)), 20), (2, 64, 12, 40, 192, list(range(40, 60)), 150)]:
    f_n, f_o, l_po, labels = synth.contrastive_case(2000 + N + h, B, N, h, h, K, H, H, new_ids)
    fn_d, fo_d, lpo_d, lab_d = [t.to(dev) for t in (f_n, f_o, l_po, labels)]
    fn_d = fn_d.contiguous(memory_format=torch.channels_last)
    pb = pixcon_prepare(fn_d, lab_d, lpo_d, fo_d, max_label=max_label, sort_by_label=False, fp16=True)
    loss_out, grad_a, stats = pixcon_loss_raw(pb, 0.07, True, True, need_grad=True, row_stats=True, precision="f16")
    m = pb.meta_host()
    print("K", K, "A", m.A, "loss", loss_out[0].item(), "nan grad rows", int(torch.isnan(grad_a[:m.A]).any(dim=1).sum()))
