"""GPU probe: the stem's 7x7 / 2 convolution (csrc/stem.hip::stem_conv7x7_kernel, fp32 image in) against what the step used
before it - layout copy + bf16 cast + MIOpen's convolution.  usage: python tools/stem_conv_probe.py [B]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ucd_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
B = int(sys.argv[1]) if len(sys.argv) > 1 else 24


def bench(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


cl = torch.channels_last
x = torch.randn(B, 3, 513, 513, device=dev)
w = (torch.randn(64, 3, 7, 7, device=dev) * 0.1).bfloat16().contiguous(memory_format=cl)
xc = x.contiguous(memory_format=cl)
xb = xc.bfloat16()
t_lib = bench(lambda: F.conv2d(xb, w, None, 2, 3))
t_full = bench(lambda: F.conv2d(x.contiguous(memory_format=cl).bfloat16(), w, None, 2, 3))
t_own = bench(lambda: hip.stem_conv7x7(xc, w))
t_own_nchw = bench(lambda: hip.stem_conv7x7(x, w))
z = hip.stem_conv7x7(xc, w)
ref = F.conv2d(xb.float(), w.float(), None, 2, 3)
err = ((z.float() - ref).norm() / ref.norm()).item()
out_mb = z.numel() * 2 / 1e6
print(f"B = {B}: MIOpen conv alone {t_lib:.1f} us, with layout copy + cast {t_full:.1f} us | own (channels-last fp32 in) {t_own:.1f} us "
      f"= {(xc.numel() * 4 + z.numel() * 2) / t_own / 1e6:.2f} TB/s, own (NCHW in) {t_own_nchw:.1f} us | err {err:.1e}, output {out_mb:.0f} MB")
