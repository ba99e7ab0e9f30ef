"""The stem's convolution kernels alone at the benchmark shape (24 x 3 x 513 x 513 fp32 image): ucd_stem_conv7x7 and the one-kernel
frozen-statistics stem, repeated for the profiler.  usage: python tools/stem_probe.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ucd_amd import hip
dev = torch.device("cuda:0")
x = torch.randn(24, 3, 513, 513, device=dev).contiguous(memory_format=torch.channels_last)
w = (torch.randn(64, 3, 7, 7, device=dev) * 0.1).bfloat16().contiguous(memory_format=torch.channels_last)
mean, scale, beta = torch.zeros(64, device=dev), torch.ones(64, device=dev), torch.zeros(64, device=dev)
for name, f in (("conv7x7", lambda: hip.stem_conv7x7(x, w)), ("conv+norm+pool", lambda: hip.stem_conv_pool(x, w, mean, scale, beta, 1, 0.01))):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): f()
    b.record(); torch.cuda.synchronize()
    print(f"{name:16s} {a.elapsed_time(b) / 20 * 1e3:7.1f} us", flush=True)
