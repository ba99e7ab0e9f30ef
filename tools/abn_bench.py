"""GPU micro-benchmark of the ABN kernels over the layer shapes of ResNet-101/DeepLab-V3 at B=24, 513^2 (bf16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ucd_amd import hip
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
shapes = [(64, 257), (64, 129), (256, 129), (128, 65), (512, 65), (256, 33), (1024, 33), (2048, 33), (512, 33)]
def timeit(f, n=20):
    for _ in range(3): f()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in evs:
        s.record(); f(); e.record()
    torch.cuda.synchronize()
    t = sorted(s.elapsed_time(e) for s, e in evs)
    return t[n // 2] * 1e3   # us
if not os.environ.get("ABN_ONLY"): print("%-14s %8s | %18s | %18s | %18s | %18s" % ("C x HW", "MB", "stats us (GB/s)", "apply us (GB/s)", "bwd_reduce", "bwd_apply"))
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for C, hw in shapes:
    x = torch.randn(B, C, hw, hw, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn_like(x); y = torch.empty_like(x); dx = torch.empty_like(x)
    M, HW = B * hw * hw, hw * hw
    buf = torch.zeros(6 * C, device=dev); w = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    sums, ks, mean, invstd, scale = buf[:2*C], buf[2*C:3*C], buf[3*C:4*C], buf[4*C:5*C], buf[5*C:]
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    mb = x.numel() * 2 / 1e6
    def f_stats(): hip.abn_stats_finalize(x, C, M, C, None, HW, sums, ks, w, rm, rv, 0.1, 1e-5, mean, invstd, scale)
    def f_apply(): hip.abn_apply(x, C, y, C, None, 0, M, C, None, HW, mean, scale, b, 1, 0.01)
    def f_red(): hip.abn_bwd_reduce(x, C, dy, C, None, 0, M, C, None, HW, mean, invstd, scale, b, 1, 0.01, sums)
    def f_bapp(): hip.abn_bwd_apply(x, C, dy, C, None, 0, dx, C, None, 0, M, C, None, HW, mean, invstd, scale, b, w, sums, M, 0, 1, 0.01)
    f_stats()
    res = []
    only = os.environ.get("ABN_ONLY")
    if only:
        f = {"STATS": f_stats, "APPLY": f_apply, "BRED": f_red, "BAPP": f_bapp}[only]
        print("%s=%.1f" % (f"{C}x{hw}", timeit(f)), end=" ", flush=True)
        continue
    for f, nb in ((f_stats, 1), (f_apply, 2), (f_red, 2), (f_bapp, 3)):
        def g():
            f()
        us = timeit(g)
        res.append("%8.1f (%6.0f)" % (us, nb * mb / us * 1e3))
    print("%-14s %8.1f | %18s | %18s | %18s | %18s" % (f"{C}x{hw}^2", mb, *res))

print()
