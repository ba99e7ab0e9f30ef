"""GPU diagnostic: label distribution of the contrast batch of the benchmark step and the (anchor block x contrast tile)
pairs by kind - pure negative (sweep 1 only), pure positive (sweep 2 only), mixed (both).
usage: python tools/pixcon_pairs.py [global_batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from ucd_amd import contrastive

sys.argv = [sys.argv[0]] + (["--global_batch", sys.argv[1]] if len(sys.argv) > 1 else [])
args = bench.parse()
dev = torch.device("cuda:0")
trainer, optimizer, scheduler, images, labels, classes = bench.build(args, dev, args.global_batch, 0)
seen = []
orig = contrastive.pixcon_prepare
def spy(*a, **k):
    pb = orig(*a, **k)
    seen.append(pb)
    return pb
contrastive.pixcon_prepare = spy
for _ in range(2):
    trainer.train_step(images, labels, optimizer, scheduler)
torch.cuda.synchronize()
pb = seen[-1]
m = pb.meta_host()
print("A", m.A, "Co", m.Co, "Apad", m.Apad, "Cpad", m.Cpad, "n_valid", m.n_valid, "min_new", m.min_new)
ca, cc = np.array(m.label_count_a), np.array(m.label_count_c)
for L in np.nonzero(cc)[0]:
    print(f"  label {L:3d}: anchors {ca[L]:6d}  contrast rows {cc[L]:6d}")
lab = pb.row_label.cpu().numpy()[: m.Cpad]
nb = (m.A + 127) // 128
nt = m.Cpad // 32
kinds = np.zeros(3, dtype=np.int64)
for b in range(nb):
    la = lab[b * 128: min((b + 1) * 128, m.A)]
    for t in range(nt):
        lc = lab[t * 32:(t + 1) * 32]
        v = lc[lc != 255]
        if v.size == 0:
            continue
        eq = la[:, None] == v[None, :]
        kinds[0 if not eq.any() else (1 if eq.all() and v.size == 32 else 2)] += 1
tot = kinds.sum()
print("block x tile pairs: pure-negative %d (%.1f%%)  pure-positive %d (%.1f%%)  mixed %d (%.1f%%)" % (
    kinds[0], 100 * kinds[0] / tot, kinds[1], 100 * kinds[1] / tot, kinds[2], 100 * kinds[2] / tot))
