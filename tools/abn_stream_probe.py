"""GPU probe: the element-wise ABN passes against torch's plain element-wise kernels moving the same bytes (copy_ = one read + one
write like abn_apply; add(out=) = two reads + one write like abn_bwd_apply), every call timed inside a replayed hipGraph of 20 launches
(no host in the loop).  UCD_ABN_GENERIC=1 selects the per-element kernels (the A/B of the packed-math forms of round 4).
usage: python tools/abn_stream_probe.py [images]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ucd_amd import hip
dev = torch.device("cuda:0")
def bench(fn, iters=10, warm=3, reps=20):
    fn(); torch.cuda.synchronize()
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        fn()
        with torch.cuda.graph(g, stream=st):
            for _ in range(reps): fn()
    torch.cuda.synchronize()
    for _ in range(warm): g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters / reps * 1e3
B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
out = []
for C, hw in ((256, 33), (1024, 33), (512, 33), (2048, 33), (128, 65), (512, 65), (64, 129), (256, 129)):
    x = torch.randn(B, C, hw, hw, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn_like(x); y = torch.empty_like(x); dx = torch.empty_like(x); r = torch.randn_like(x)
    M, HW = B * hw * hw, hw * hw
    buf = torch.zeros(6 * C, device=dev); w = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    sums, ks, mean, invstd, scale = buf[:2*C], buf[2*C:3*C], buf[3*C:4*C], buf[4*C:5*C], buf[5*C:]
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    hip.abn_stats_finalize(x, C, M, C, None, HW, sums, ks, w, rm, rv, 0.1, 1e-5, mean, invstd, scale)
    t_copy = bench(lambda: y.copy_(x))
    t_add = bench(lambda: torch.add(x, dy, out=dx))
    t_app = bench(lambda: hip.abn_apply(x, C, y, C, None, 0, M, C, None, HW, mean, scale, b, 1, 0.01))
    xi = x.clone()
    t_inpl = bench(lambda: hip.abn_apply(xi, C, xi, C, None, 0, M, C, None, HW, mean, scale, b, 1, 0.01))
    big = [torch.empty_like(x) for _ in range(max(2, int(600e6 / (x.numel() * 2))))]      # > 256 MB of distinct outputs: no cache reuse
    it = [0]
    def cold():
        it[0] = (it[0] + 1) % len(big)
        hip.abn_apply(x, C, big[it[0]], C, None, 0, M, C, None, HW, mean, scale, b, 1, 0.01)
    t_cold = bench(cold)
    it[0] = 0
    def cold_copy():
        it[0] = (it[0] + 1) % len(big)
        big[it[0]].copy_(x)
    t_ccopy = bench(cold_copy)
    del big
    t_appr = bench(lambda: hip.abn_apply(x, C, y, C, r, C, M, C, None, HW, mean, scale, b, 1, 0.01))
    t_bapp = bench(lambda: hip.abn_bwd_apply(x, C, dy, C, None, 0, dx, C, None, 0, M, C, None, HW, mean, invstd, scale, b, w, sums, M, 0, 1, 0.01))
    t_red = bench(lambda: hip.abn_bwd_reduce(x, C, dy, C, None, 0, M, C, None, HW, mean, invstd, scale, b, 1, 0.01, sums))
    print(f"{C:5d}x{hw:3d}^2 {x.numel()*2/1e6:6.1f} MB | copy {t_copy:6.1f} add3 {t_add:6.1f} | apply {t_app:6.1f} in-place {t_inpl:6.1f} rotating-out {t_cold:6.1f} (copy {t_ccopy:6.1f}) apply+res {t_appr:6.1f} bwd_apply {t_bapp:6.1f} bwd_reduce {t_red:6.1f}", flush=True)
