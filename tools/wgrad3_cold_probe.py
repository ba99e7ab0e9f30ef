"""Is the three-tap weight gradient slower in the step (73 us) than in its probe (53 us) because its operands are cold?  One call
with (a) everything warm (back-to-back calls), (b) after a 1 GB stream through the memory-side cache, (c) only X cold (dZ touched
again after the flush - the step's case: dZ has just been written by the ABN backward).  usage: python tools/wgrad3_cold_probe.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ucd_amd import hip
dev = torch.device("cuda:0")
B, H, W, C = 24, 33, 33, 256
M = B * H * W
x = torch.randn(M, C, device=dev).bfloat16()
dz = torch.randn(M, C, device=dev).bfloat16()
dw = torch.empty(C, 9 * C, device=dev, dtype=torch.bfloat16)
big = torch.empty(1 << 28, device=dev, dtype=torch.float32)          # 1 GiB
def run(): hip.conv_wgrad(dz, x, dw, conv3=(H, W, 1))
def timed(prep):
    ts = []
    for _ in range(12):
        prep()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]
for _ in range(3): run()
def warm(): run()
def cold(): big.add_(1.0)
def x_cold():
    big.add_(1.0); dz.mul_(1.0)
print(f"wgrad3 256->256 33x33 B=24 (two launches: product + slab sum): warm {timed(warm):.1f} us, all operands cold {timed(cold):.1f} us, "
      f"X cold / dZ warm {timed(x_cold):.1f} us", flush=True)
# back to back (the step's regime: the chip never idles, clocks settle at their sustained level)
torch.cuda.synchronize()
for n in (20, 200, 1000):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): run()
    b.record(); torch.cuda.synchronize()
    print(f"wgrad3 back to back x{n}: {a.elapsed_time(b) * 1e3 / n:.1f} us per call (product + sum)", flush=True)
# behind a heavy MFMA kernel stream (3x3 forward products), interleaved 1:1
from ucd_amd import hip as _h
w3 = (torch.randn(C, 9 * C, device=dev) * 0.02).bfloat16()
y = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
def fwd(): _h.conv1x1(x, w3, y, conv3=(H, W, 1))
for _ in range(3): fwd()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(300): fwd()
b.record(); torch.cuda.synchronize()
t_f = a.elapsed_time(b) * 1e3 / 300
a.record()
for _ in range(300): fwd(); run()
b.record(); torch.cuda.synchronize()
print(f"3x3 forward alone {t_f:.1f} us; forward + wgrad3 interleaved {a.elapsed_time(b) * 1e3 / 300:.1f} us per pair", flush=True)
