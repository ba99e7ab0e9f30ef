"""GPU probe: the short-K wide-N 1x1 products (conv3 of the mod4 blocks, 256 -> 1024) in every output mode the step uses, timed inside
a replayed hipGraph (no host gaps).  Run once per kernel form: UCD_CONV_RA=1 (resident-A form, default) and UCD_CONV_RA=0 (tiled forms).
usage: UCD_CONV_RA=0|1 python tools/conv_ra_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ucd_amd import hip  # noqa: E402

dev = torch.device("cuda:0")


def bench(fn, iters=40):
    """median-free: `iters` back-to-back launches inside one graph replay, three replays, best"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters):
                fn()
    best = 1e9
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / iters * 1e3)
    return best


print("UCD_CONV_RA =", os.environ.get("UCD_CONV_RA", "(default 1)"))
for B in (24, 12, 6, 3):
    M, K, N = B * 33 * 33, 256, 1024
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    res = torch.randn(M, N, device=dev).bfloat16()
    z3 = torch.randn(M, N, device=dev).bfloat16()
    v = torch.rand(N, device=dev) + 0.5
    part = hip.conv1x1_stats_partial(M, N, dev)
    part2 = torch.empty(hip.conv1x1_row_tiles(M), 2, N, device=dev)
    t_plain = bench(lambda: hip.conv1x1(a, w, y))
    t_stats = bench(lambda: hip.conv1x1(a, w, y, out_mode=2, partial=part))
    t_aff = bench(lambda: hip.conv1x1(a, w, y, out_mode=1, out_norm=(v, v, v, None, 1, 0.01), residual=res))
    t_acc = bench(lambda: hip.conv1x1(a, w, y, accumulate=True))
    t_link = bench(lambda: hip.conv1x1(a, w, y, out_mode=4, out_norm=(v, None, None, v, 1, 0.01), residual=res, side2=z3, partial=part2,
                                       accumulate=True))
    by = lambda extra: 2 * (M * K + (1 + extra) * M * N) / 1e6   # MB
    print(f"images {B:2d} M={M:6d} {K}->{N}: plain {t_plain:6.1f} us ({by(0) / t_plain:5.2f} TB/s)  +stats {t_stats:6.1f}  "
          f"affine+res {t_aff:6.1f} ({by(1) / t_aff:5.2f})  accumulate {t_acc:6.1f} ({by(1) / t_acc:5.2f})  "
          f"block link {t_link:6.1f} ({by(3) / t_link:5.2f} TB/s)", flush=True)
