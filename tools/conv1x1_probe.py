"""GPU probe: the fused 1x1-convolution GEMM (csrc/conv1x1.hip) - every mode against fp32 torch on the device, and its time
against the tuned hipBLASLt entry point it replaces, on the layer shapes of the benchmark workload (B = 24, 513^2).
usage: python tools/conv1x1_probe.py [--quick]"""
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
    return s.elapsed_time(e) / iters * 1e3   # us


def lrelu(x, s=0.01):
    return F.leaky_relu(x, s)


def check(M, K, N, g):
    a = (torch.randn(M, K, device=dev, generator=g) * 1.3 + 0.2).bfloat16()
    w = (torch.randn(N, K, device=dev, generator=g) * (2.0 / K) ** 0.5).bfloat16()
    res = torch.randn(M, N, device=dev, generator=g).bfloat16()
    im, isc, ish = torch.randn(K, device=dev, generator=g) * 0.3, torch.rand(K, device=dev, generator=g) + 0.5, torch.randn(K, device=dev, generator=g) * 0.2
    om, osc, osh = torch.randn(N, device=dev, generator=g) * 0.3, torch.rand(N, device=dev, generator=g) + 0.5, torch.randn(N, device=dev, generator=g) * 0.2
    oinv = torch.rand(N, device=dev, generator=g) + 0.5
    af, wf = a.float(), w.float()
    errs = {}
    # mode 0
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    hip.conv1x1(a, w, y)
    ref = af @ wf.t()
    errs["plain"] = ((y.float() - ref).norm() / ref.norm()).item()
    # mode 0 + accumulate
    y2 = res.clone()
    hip.conv1x1(a, w, y2, accumulate=True)
    ref2 = ref + res.float()
    errs["acc"] = ((y2.float() - ref2).norm() / ref2.norm()).item()
    # prologue + mode 1 (+ residual)
    ap = lrelu((af - im) * isc + ish).bfloat16().float()
    refp = ap @ wf.t()
    hip.conv1x1(a, w, y, in_norm=(im, isc, ish, hip.ACT_LEAKY_RELU, 0.01), out_mode=1,
                out_norm=(om, osc, osh, None, hip.ACT_LEAKY_RELU, 0.01), residual=res)
    ref3 = lrelu((refp - om) * osc + osh + res.float())
    errs["pro+affine+res"] = ((y.float() - ref3).norm() / ref3.norm()).item()
    # mode 2: statistics
    tiles = hip.load().ucd_conv1x1_row_tiles(M)
    part = hip.conv1x1_stats_partial(M, N, dev)
    hip.conv1x1(a, w, y, out_mode=2, partial=part)
    errs["stats_y"] = ((y.float() - ref).norm() / ref.norm()).item()
    buf = torch.zeros(6 * N, device=dev)
    rm, rv = torch.zeros(N, device=dev), torch.ones(N, device=dev)
    gamma = torch.rand(N, device=dev, generator=g) - 0.3
    hip._check(hip.load().ucd_conv1x1_stats_finalize(hip.ptr(part), M, N, hip.ptr(gamma), hip.ptr(rm), hip.ptr(rv), 0.1, 1e-5,
                                                     hip.ptr(buf), None, hip.NORM_ABS_GAMMA, hip.stream()), "fin")
    yf = y.float()
    mean, var = yf.mean(0), yf.var(0, unbiased=False)
    errs["mean"] = ((buf[3 * N:4 * N] - mean).abs().max() / mean.abs().max()).item()
    errs["invstd"] = ((buf[4 * N:5 * N] * torch.sqrt(var + 1e-5) - 1).abs().max()).item()
    errs["scale"] = ((buf[5 * N:] - (gamma.abs() + 1e-5) / torch.sqrt(var + 1e-5)).abs().max()).item()
    errs["rvar"] = ((rv - (0.9 + 0.1 * var * M / (M - 1))).abs().max()).item()
    # mode 3: activation backward + sums (x = res as the fused layer's pre-norm input)
    part2 = torch.zeros(tiles, 2, N, device=dev)
    hip.conv1x1(a, w, y, out_mode=3, out_norm=(om, osc, osh, oinv, hip.ACT_LEAKY_RELU, 0.01), residual=res, partial=part2)
    z = (res.float() - om) * osc + osh
    dz = ref * torch.where(z > 0, 1.0, 0.01)
    errs["dz"] = ((y.float() - dz).norm() / dz.norm()).item()
    sums = torch.zeros(2 * N, device=dev)
    hip._check(hip.load().ucd_abn_reduce_partials(hip.ptr(part2), tiles, N, hip.ptr(sums), None, None, 0, hip.stream()), "red")
    dzr = y.float()
    s1, s2 = dzr.sum(0), (dzr * (res.float() - om) * oinv).sum(0)
    errs["sum_dz"] = ((sums[:N] - s1).abs().max() / s1.abs().max()).item()
    errs["sum_dzx"] = ((sums[N:] - s2).abs().max() / s2.abs().max()).item()
    # weight gradient
    if N % 128 == 0 and K % 128 == 0:
        dy = res
        dw = torch.empty(N, K, device=dev, dtype=torch.bfloat16)
        hip.conv1x1_wgrad(dy, a, dw)
        refw = dy.float().t() @ af
        errs["wgrad"] = ((dw.float() - refw).norm() / refw.norm()).item()
        hip.conv1x1_wgrad(dy, a, dw, in_norm=(im, isc, ish, hip.ACT_LEAKY_RELU, 0.01))
        refw = dy.float().t() @ ap
        errs["wgrad_pro"] = ((dw.float() - refw).norm() / refw.norm()).item()
    wt = torch.empty(K, N, device=dev, dtype=torch.bfloat16)
    hip.transpose_bf16(w, wt)
    errs["transpose"] = float(not torch.equal(wt, w.t().contiguous()))
    bad = {k: v for k, v in errs.items() if not (v < (5e-3 if k not in ("transpose",) else 0.5))}
    print(f"check M={M} K={K} N={N}: " + " ".join(f"{k}={v:.1e}" for k, v in errs.items()) + ("  BAD " + str(bad) if bad else "  ok"))
    return not bad


def timing(M, K, N):
    a = torch.randn(M, K, device=dev).bfloat16()
    w = torch.randn(N, K, device=dev).bfloat16()
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    res = torch.randn(M, N, device=dev).bfloat16()
    v = torch.rand(max(K, N), device=dev) + 0.5
    tiles = hip.load().ucd_conv1x1_row_tiles(M)
    part = hip.conv1x1_stats_partial(M, N, dev)
    t_lib = bench(lambda: hip.gemm_bf16(0, a, w, y)) if hip.gemm_available() else float("nan")
    t_plain = bench(lambda: hip.conv1x1(a, w, y))
    t_stats = bench(lambda: hip.conv1x1(a, w, y, out_mode=2, partial=part))
    sbuf = torch.zeros(6 * N, device=dev)
    t_fin = bench(lambda: hip.conv1x1_stats_finalize(part, M, N, None, None, None, 0.1, 1e-5, sbuf))
    t_pro = bench(lambda: hip.conv1x1(a, w, y, in_norm=(v, v, v, 1, 0.01)))
    t_aff = bench(lambda: hip.conv1x1(a, w, y, out_mode=1, out_norm=(v, v, v, None, 1, 0.01)))
    t_affr = bench(lambda: hip.conv1x1(a, w, y, out_mode=1, out_norm=(v, v, v, None, 1, 0.01), residual=res))
    part2 = torch.zeros(tiles, 2, N, device=dev)
    t_bwd = bench(lambda: hip.conv1x1(a, w, y, out_mode=3, out_norm=(v, v, v, v, 1, 0.01), residual=res, partial=part2))
    t_full = bench(lambda: hip.conv1x1(a, w, y, in_norm=(v, v, v, 1, 0.01), out_mode=1, out_norm=(v, v, v, None, 1, 0.01), residual=res))
    byt = 2 * (M * K + M * N + N * K)
    line = (f"M={M:6d} K={K:4d} N={N:4d}  hipBLASLt {t_lib:7.1f} us | own plain {t_plain:7.1f} us ({byt / t_plain / 1e6:5.2f} TB/s, "
            f"{2 * M * K * N / t_plain / 1e6:6.1f} TF/s)  +stats {t_stats:6.1f} (finalize {t_fin:4.1f}) pro {t_pro:6.1f} aff {t_aff:6.1f} aff+res {t_affr:6.1f} bwdact {t_bwd:6.1f} pro+aff+res {t_full:6.1f}")
    if N % 128 == 0 and K % 128 == 0:
        dw = torch.empty(N, K, device=dev, dtype=torch.bfloat16)
        t_w = bench(lambda: hip.conv1x1_wgrad(res, a, dw))
        S = 8 if M % 8 == 0 else 1
        t_wl = bench(lambda: torch.bmm(res.view(S, M // S, N).transpose(1, 2), a.view(S, M // S, K)).sum(0))
        line += f" | wgrad own {t_w:7.1f} us  bmm+sum {t_wl:7.1f} us"
    print(line, flush=True)


if __name__ == "__main__":
    g = torch.Generator(dev).manual_seed(3)
    ok = True
    for M, K, N in [(300, 64, 64), (1000, 128, 256), (2178, 256, 128), (4356, 512, 1024), (777, 64, 256)]:
        ok &= check(M, K, N, g)
    print("CONV1X1_CHECK", "OK" if ok else "FAILED")
    if "--quick" not in sys.argv:
        M33, M65, M129 = 24 * 33 * 33, 24 * 65 * 65, 24 * 129 * 129
        for M, K, N in [(M33, 1024, 256), (M33, 256, 1024), (M33, 2048, 512), (M33, 512, 2048), (M33, 1024, 2048),
                        (M33, 2048, 256), (M33, 1024, 512), (M65, 512, 128), (M65, 128, 512), (M65, 512, 256),
                        (M129, 256, 64), (M129, 64, 256), (M129, 64, 64), (M129, 256, 128)]:
            timing(M, K, N)
