"""GPU probe: the implicit-GEMM 3x3 mode of csrc/conv1x1.hip against F.conv2d (MIOpen, solver search on) - result and time,
forward and the input gradient (same kernel on the flipped / transposed weight), on the 3x3 layer shapes of the workload.
usage: python tools/conv3x3_probe.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ucd_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True


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


def own(x, w, y, d, **kw):
    B, K, H, W = x.shape
    N = w.shape[0]
    return hip.conv1x1(rows(x), w.permute(0, 2, 3, 1).reshape(N, 9 * K), rows(y), conv3=(H, W, d), **kw)


def run(B, K, N, H, W, d):
    cl = torch.channels_last
    x = torch.randn(B, K, H, W, device=dev).bfloat16().contiguous(memory_format=cl)
    w = (torch.randn(N, K, 3, 3, device=dev) * (2.0 / (9 * K)) ** 0.5).bfloat16().contiguous(memory_format=cl)
    y = torch.empty(B, N, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=cl)
    own(x, w, y, d)
    ref = F.conv2d(x.float(), w.float(), None, 1, d, d)
    err = ((y.float() - ref).norm() / ref.norm()).item()
    part = hip.conv1x1_stats_partial(B * H * W, N, dev)
    t_lib = bench(lambda: F.conv2d(x, w, None, 1, d, d))
    t_own = bench(lambda: own(x, w, y, d))
    t_stats = bench(lambda: own(x, w, y, d, out_mode=2, partial=part))
    v = torch.rand(N, device=dev) + 0.5
    t_aff = bench(lambda: own(x, w, y, d, out_mode=1, out_norm=(v, v, v, None, 1, 0.01)))
    # input gradient: conv of dy [B, N, H, W] with w.flip(2, 3).transpose(0, 1) [K, N, 3, 3]
    dy = torch.randn(B, N, H, W, device=dev).bfloat16().contiguous(memory_format=cl)
    wt = w.flip(2, 3).transpose(0, 1).contiguous(memory_format=cl)
    dx = torch.empty_like(x)
    own(dy, wt, dx, d)
    refdx = torch.nn.grad.conv2d_input(x.shape, w.float(), dy.float(), 1, d, d)
    errdx = ((dx.float() - refdx).norm() / refdx.norm()).item()
    t_dlib = bench(lambda: F.conv2d(dy, wt, None, 1, d, d))
    t_down = bench(lambda: own(dy, wt, dx, d))
    gf = 2 * B * H * W * K * N * 9 / 1e9
    print(f"B={B} {K:4d}->{N:4d} {H}x{W} d={d:2d}  err {err:.1e} dx {errdx:.1e} | fwd MIOpen {t_lib:7.1f} us own {t_own:7.1f} ({gf / t_own * 1e3:6.0f} TF/s) "
          f"+stats {t_stats:7.1f} affine {t_aff:7.1f} | dgrad MIOpen(fwd solver) {t_dlib:7.1f} own {t_down:7.1f}", flush=True)


if __name__ == "__main__":
    for cfg in [(24, 256, 256, 33, 33, 1), (24, 512, 512, 33, 33, 2), (24, 128, 128, 65, 65, 1), (24, 64, 64, 129, 129, 1),
                (24, 2048, 256, 33, 33, 6), (24, 2048, 256, 33, 33, 18), (3, 256, 256, 33, 33, 1)]:
        run(*cfg)
