"""GPU probe: the block-link input-gradient product (ucd_conv1x1 out_mode 4: conv1's dgrad + the shortcut's gradient + the
previous block's activation derivative and bn3 sums) against its plain accumulate form (out_mode 0) at the mod4 shape.
usage: python tools/blocklink_probe.py   (bash tools/prof_kernel.sh conv1x1_kernel -- python3 tools/blocklink_probe.py for counters)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ucd_amd import hip  # noqa: E402

dev = torch.device("cuda:0")


def bench(fn, iters=30, warm=5):
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


for M, K, N in ((26136, 256, 1024), (26136, 512, 2048), (101400, 128, 512), (399384, 64, 256)):
    dz = torch.randn(M, K, device=dev).bfloat16()
    wt = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
    dskip = torch.randn(M, N, device=dev).bfloat16()
    out = torch.randn(M, N, device=dev).bfloat16()          # the previous block's output (sign of the activation)
    z3 = torch.randn(M, N, device=dev).bfloat16()           # its conv3 output
    mean, invstd = torch.randn(N, device=dev) * 0.1, torch.rand(N, device=dev) + 0.5
    partial = torch.empty(hip.conv1x1_row_tiles(M), 2, N, device=dev)
    y0 = dskip.clone()
    t_acc = bench(lambda: hip.conv1x1(dz, wt, y0, accumulate=True))
    y4 = dskip.clone()
    t_link = bench(lambda: hip.conv1x1(dz, wt, y4, out_mode=4, out_norm=(mean, None, None, invstd, 1, 0.01), residual=out, side2=z3,
                                       partial=partial, accumulate=True))
    by0 = 2 * (M * K + 2 * M * N)
    by4 = 2 * (M * K + 4 * M * N)
    print(f"M={M} K={K} N={N}: accumulate {t_acc:6.1f} us ({by0 / t_acc / 1e6:5.2f} TB/s)   block link {t_link:6.1f} us ({by4 / t_link / 1e6:5.2f} TB/s)")
