"""GPU probe (round 6): graph-replayed time per launch of the GEMM / implicit-GEMM kernel at the per-rank shapes of the multi-GPU
split (3 and 6 images) and at 24 images - the kernel alone, no launch gaps (96 launches in one replay).  Environment switches of
csrc/conv1x1.hip (UCD_CONV_LW_PF, UCD_CONV_LW_NL, UCD_CONV_PIPE, ...) are read once per process: run once per setting.
usage: [UCD_...=..] python tools/lw_probe.py [batches, default 3,24]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ucd_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
cl = torch.channels_last


def rows(t):
    b, c, h, w = t.shape
    return t.permute(0, 2, 3, 1).reshape(b * h * w, c)


def replay_us(fn, n=96, reps=5):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (reps * n) * 1e3


def run3(B, K, N, H, d):
    x = torch.randn(B, K, H, H, device=dev).bfloat16().contiguous(memory_format=cl)
    w = (torch.randn(N, K, 3, 3, device=dev) * (2.0 / (9 * K)) ** 0.5).bfloat16().contiguous(memory_format=cl)
    y = torch.empty(B, N, H, H, device=dev, dtype=torch.bfloat16).contiguous(memory_format=cl)
    w2 = w.permute(0, 2, 3, 1).reshape(N, 9 * K)
    part = hip.conv1x1_stats_partial(B * H * H, N, dev)
    t0 = replay_us(lambda: hip.conv1x1(rows(x), w2, rows(y), conv3=(H, H, d)))
    t2 = replay_us(lambda: hip.conv1x1(rows(x), w2, rows(y), conv3=(H, H, d), out_mode=2, partial=part))
    gf = 2 * B * H * H * K * N * 9 / 1e9
    print(f"3x3 B={B:2d} {K:4d}->{N:4d} {H}^2 d={d:2d} | plain {t0:7.2f} us ({gf / t0 * 1e3:5.0f} TF/s)  +stats {t2:7.2f}", flush=True)


def run1(M, K, N):
    a = (torch.randn(M, K, device=dev) * 1.3 + 0.2).bfloat16()
    w = (torch.randn(N, K, device=dev) * (2.0 / K) ** 0.5).bfloat16()
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    res = torch.randn(M, N, device=dev).bfloat16()
    part = hip.conv1x1_stats_partial(M, N, dev)
    v = torch.rand(N, device=dev) + 0.5
    t0 = replay_us(lambda: hip.conv1x1(a, w, y))
    t2 = replay_us(lambda: hip.conv1x1(a, w, y, out_mode=2, partial=part))
    t1 = replay_us(lambda: hip.conv1x1(a, w, y, out_mode=1, out_norm=(v, v, v, None, 1, 0.01), residual=res))
    gb = 2.0 * (M * K + M * N) / 1e9
    print(f"1x1 M={M:6d} {K:4d}->{N:4d} | plain {t0:7.2f} us ({gb / t0 * 1e3:5.2f} TB/s)  +stats {t2:7.2f}  affine+res {t1:7.2f}", flush=True)


if __name__ == "__main__":
    batches = [int(b) for b in (sys.argv[1] if len(sys.argv) > 1 else "3,24").split(",")]
    print("env:", {k: v for k, v in os.environ.items() if k.startswith("UCD_")})
    for B in batches:
        for cfg in [(256, 256, 33, 1), (512, 512, 33, 2), (2048, 256, 33, 12), (128, 128, 65, 1), (64, 64, 129, 1)]:
            run3(B, *cfg)
        M = B * 33 * 33
        for K, N in [(1024, 256), (256, 1024), (2048, 512), (512, 2048), (1024, 2048), (2048, 256)]:
            run1(M, K, N)
