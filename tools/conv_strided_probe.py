"""GPU probe: the strided layers (first block of a stage: conv2 3x3 stride 2, proj_conv 1x1 stride 2) on the own kernels
(csrc/conv1x1.hip ``stride``, csrc/wgrad.hip STR) against MIOpen (solver search on) - forward, forward + statistics, weight
gradient, and the library's input gradient on its own.  usage: python tools/conv_strided_probe.py [B]"""
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


def rows(t):
    b, c, h, w = t.shape
    return t.permute(0, 2, 3, 1).reshape(b * h * w, c)


print(f"B = {B}   layer                       lib fwd   own fwd  own+stats   lib wgrad  own wgrad   lib dgrad   (us)")
for K, N, H, k in ((256, 512, 129, 1), (512, 1024, 65, 1), (128, 128, 129, 3), (256, 256, 65, 3)):
    cl = torch.channels_last
    pad = 1 if k == 3 else 0
    x = torch.randn(B, K, H, H, device=dev).bfloat16().contiguous(memory_format=cl)
    w = (torch.randn(N, K, k, k, device=dev) * (2.0 / (k * k * K)) ** 0.5).bfloat16().contiguous(memory_format=cl)
    OH = (H - 1) // 2 + 1
    y = torch.empty(B, N, OH, OH, device=dev, dtype=torch.bfloat16).contiguous(memory_format=cl)
    wr = w.permute(0, 2, 3, 1).reshape(N, k * k * K)
    kw = dict(conv3=(H, H, 1, 2)) if k == 3 else dict(strided=(H, H, 2))
    part = hip.conv1x1_stats_partial(B * OH * OH, N, dev)
    hip.conv1x1(rows(x), wr, rows(y), **kw)
    ref = F.conv2d(x, w, None, 2, pad)
    err = ((y.float() - ref.float()).norm() / ref.float().norm()).item()
    t_lib = bench(lambda: F.conv2d(x, w, None, 2, pad))
    t_own = bench(lambda: hip.conv1x1(rows(x), wr, rows(y), **kw))
    t_st = bench(lambda: hip.conv1x1(rows(x), wr, rows(y), out_mode=2, partial=part, **kw))
    dz = torch.randn_like(y)
    dw = torch.empty(N, k * k * K, device=dev, dtype=torch.bfloat16)
    cb = torch.ops.aten.convolution_backward
    t_lw = bench(lambda: cb(dz, x, w, None, [2, 2], [pad, pad], [1, 1], False, [0, 0], 1, [False, True, False]))
    t_ow = bench(lambda: hip.conv_wgrad(rows(dz), rows(x), dw, **kw))
    t_ld = bench(lambda: cb(dz, x, w, None, [2, 2], [pad, pad], [1, 1], False, [0, 0], 1, [True, False, False]))
    print(f"  {k}x{k}/2 {K:4d} -> {N:4d} at {H:3d}^2 (err {err:.1e})  {t_lib:8.1f} {t_own:8.1f} {t_st:9.1f}  {t_lw:10.1f} {t_ow:9.1f}  {t_ld:10.1f}")
