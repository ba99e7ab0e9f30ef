"""Does any kernel read memory it did not write?  Run the student's train-mode forward on a fresh allocator, then
fill every workspace and a few GiB of freed allocator blocks with NaN bit patterns and run it again: the two results
must be bit-identical (run on the GPU box: python tools/poison_probe.py)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from ucd_amd import hip, synth
import test_step_gpu as T

g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "model_full.npz"))
dev = torch.device("cuda:0")
opts = T._opts()
model, model_old, classes = T._build(opts, dev)
img = synth.images(500, 2, 65).to(dev)
model.train()
ref = g["student_train_sem"]

def run(tag):
    ls, fs = model(img.clone())
    got = fs["sem"].detach().cpu().numpy()
    print(tag, "rel L2 vs golden", float(np.linalg.norm(got - ref) / np.linalg.norm(ref)), flush=True)
    return got

a = run("fresh   ")
b = run("again   ")
def poison():
    for k, buf in hip._workspaces.items():
        buf.view(torch.int32)[: buf.numel() // 4].fill_(0x7FC00000 if False else 0x7F800001)
    blocks = [torch.full((64 << 20,), float("nan"), device=dev) for _ in range(8)]
    small = [torch.full((n,), float("nan"), device=dev) for n in (64, 256, 512, 2048, 8192, 65536, 1 << 20) for _ in range(64)]
    del blocks, small
poison()
c = run("poisoned")
print("fresh==again", np.array_equal(a, b), " fresh==poisoned", np.array_equal(a, c), " nan", np.isnan(c).any())
big = [torch.full((n,), 1e30, device=dev) for n in (64, 256, 512, 2048, 8192, 65536, 1 << 20) for _ in range(64)]
del big
for k, buf in hip._workspaces.items():
    buf.view(torch.float32)[: buf.numel() // 4].fill_(1e30)
d = run("huge    ")
print("fresh==huge", np.array_equal(a, d))
