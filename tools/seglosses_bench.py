"""GPU micro-benchmark of the fused logit losses (ucd_seg_losses) at the benchmark shape (B = 24, 513^2, 21 student / 16 teacher
classes) or, with `ade`, at the per-rank shape of configs[3] (B = 3, 512^2, 151 / 101 classes).
usage: python tools/seglosses_bench.py [ade]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ucd_amd import synth
from ucd_amd.loss import fused_seg_losses
dev = torch.device("cuda:0")
B, H, h, Ctot, K = (3, 512, 32, 151, 101) if "ade" in sys.argv[1:] else (24, 513, 33, 21, 16)
torch.manual_seed(0)
sem = torch.randn(B, Ctot, h, h, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
sem_old = torch.randn(B, K, h, h, device=dev).contiguous(memory_format=torch.channels_last)
labels = synth.seg_labels(7, B, H, H, range(K, Ctot)).to(dev)
def run():
    return fused_seg_losses(sem, sem_old, labels, K, 1.0, 10.0)
for _ in range(3): out = run()
torch.cuda.synchronize()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
for s, e in evs:
    s.record(); out = run(); e.record()
torch.cuda.synchronize()
ts = sorted(s.elapsed_time(e) for s, e in evs)
print(f"B={B} {H}x{H} classes {Ctot}/{K}: ucd_seg_losses (+ wrapper): median {ts[10] * 1e3:.1f} us  min {ts[0] * 1e3:.1f} us; losses {[float(o) for o in out[:2]]}")
