"""GPU probe: the pipeline forms of the double-buffered GEMM / implicit-GEMM kernel (csrc/conv1x1.hip: UCD_CONV_PIPE = 2x64 | 4x32 |
4x64, read once per process - run this script once per setting) on the layer shapes of the step at 24 and at 3 images: result
against an fp32 product and time per call.
usage: UCD_CONV_PIPE=4x32 python tools/conv_pipe_probe.py"""
import os
import sys

import torch
import torch.nn.functional as F

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


def rows(t):
    b, c, h, w = t.shape
    return t.permute(0, 2, 3, 1).reshape(b * h * w, c)


def run3(B, K, N, H, d):
    cl = torch.channels_last
    x = torch.randn(B, K, H, H, device=dev).bfloat16().contiguous(memory_format=cl)
    w = (torch.randn(N, K, 3, 3, device=dev) * (2.0 / (9 * K)) ** 0.5).bfloat16().contiguous(memory_format=cl)
    y = torch.empty(B, N, H, H, device=dev, dtype=torch.bfloat16).contiguous(memory_format=cl)
    w2 = w.permute(0, 2, 3, 1).reshape(N, 9 * K)
    part = hip.conv1x1_stats_partial(B * H * H, N, dev)
    call = lambda **kw: hip.conv1x1(rows(x), w2, rows(y), conv3=(H, H, d), **kw)
    call()
    ref = F.conv2d(x.float(), w.float(), None, 1, d, d)
    err = ((y.float() - ref).norm() / ref.norm()).item()
    t0, t2 = bench(call), bench(lambda: call(out_mode=2, partial=part))
    gf = 2 * B * H * H * K * N * 9 / 1e9
    print(f"3x3 B={B:2d} {K:4d}->{N:4d} {H}^2 d={d:2d} err {err:.1e} | plain {t0:7.1f} us ({gf / t0 * 1e3:5.0f} TF/s)  +stats {t2:7.1f}", flush=True)


def run1(M, K, N):
    a = (torch.randn(M, K, device=dev) * 1.3 + 0.2).bfloat16()
    w = (torch.randn(N, K, device=dev) * (2.0 / K) ** 0.5).bfloat16()
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    res = torch.randn(M, N, device=dev).bfloat16()
    part = hip.conv1x1_stats_partial(M, N, dev)
    v = torch.rand(N, device=dev) + 0.5
    hip.conv1x1(a, w, y)
    ref = a.float() @ w.float().t()
    err = ((y.float() - ref).norm() / ref.norm()).item()
    t0 = bench(lambda: hip.conv1x1(a, w, y))
    t2 = bench(lambda: hip.conv1x1(a, w, y, out_mode=2, partial=part))
    t1 = bench(lambda: hip.conv1x1(a, w, y, out_mode=1, out_norm=(v, v, v, None, 1, 0.01), residual=res))
    gb = 2.0 * (M * K + M * N) / 1e9
    print(f"1x1 M={M:6d} {K:4d}->{N:4d} err {err:.1e} | plain {t0:7.1f} us ({gb / t0 * 1e3:5.2f} TB/s)  +stats {t2:7.1f}  affine+res {t1:7.1f}", flush=True)


if __name__ == "__main__":
    print("UCD_CONV_PIPE =", os.environ.get("UCD_CONV_PIPE", "(auto)"))
    for B in (24, 3):
        for cfg in [(256, 256, 33, 1), (512, 512, 33, 2), (2048, 256, 33, 12), (128, 128, 65, 1), (64, 64, 129, 1)]:
            if B == 24 and cfg[2] > 33:
                continue                      # full grids: the single-stage form, not a DB launch
            run3(B, *cfg)
        M = B * 33 * 33
        for K, N in [(1024, 256), (256, 1024), (2048, 512), (512, 2048), (1024, 2048), (2048, 256), (1024, 1024)]:
            run1(M, K, N)
        if B == 3:
            for (hw, K, N) in [(65, 512, 128), (65, 128, 512), (129, 256, 64), (129, 64, 256)]:
                run1(B * hw * hw, K, N)
