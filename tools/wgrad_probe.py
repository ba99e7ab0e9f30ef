"""Weight gradient of the wide 1x1 convolutions, dW[Co,Ci] = dY^T[Co,M] . X[M,Ci] with M = B*H*W = 26136: the GEMM
hipBLASLt picks for the plain mm has a 64x256 macro tile and no split-K -> 16 workgroups on 256 CUs (~150 us).
Compare with a per-image batched GEMM + sum (python tools/wgrad_probe.py on the GPU box)."""
import torch
dev = torch.device("cuda:0")
import sys
B, HW = (int(sys.argv[1]) if len(sys.argv) > 1 else 24), 33 * 33
M = B * HW
def timeit(f, n=20):
    for _ in range(3): f()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in evs:
        s.record(); f(); e.record()
    torch.cuda.synchronize()
    t = sorted(s.elapsed_time(e) for s, e in evs)
    return t[n // 2] * 1e3
for Co, Ci in ((1024, 256), (256, 1024), (2048, 512), (512, 2048), (2048, 1024), (256, 2048)):
    x = torch.randn(M, Ci, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(M, Co, device=dev, dtype=torch.bfloat16)
    w = torch.randn(Co, Ci, device=dev, dtype=torch.bfloat16)
    ref = (dy.float().t() @ x.float())
    res = {}
    def plain(): return dy.t() @ x
    def bmm_img(): return torch.bmm(dy.view(B, HW, Co).transpose(1, 2), x.view(B, HW, Ci)).sum(0, dtype=torch.float32)
    def bmm_img_f32():
        return torch.bmm(dy.view(B, HW, Co).transpose(1, 2), x.view(B, HW, Ci), out_dtype=torch.float32).sum(0)
    def bmm_s(S):
        def f(): return torch.bmm(dy.view(S, M // S, Co).transpose(1, 2), x.view(S, M // S, Ci)).sum(0, dtype=torch.float32)
        return f
    cands = {"plain": plain, "bmmB": bmm_img}
    for S in (2, 3, 4, 6, 8, 9, 11, 12, 16):
        if M % S == 0 and S != B:
            cands[f"bmm{S}"] = bmm_s(S)
    try:
        bmm_img_f32()
    except Exception as e:
        print("out_dtype unsupported:", repr(e)[:80])
    line = f"Co={Co:5d} Ci={Ci:5d} "
    for name, f in cands.items():
        us = timeit(f)
        err = ((f().float() - ref).norm() / ref.norm()).item()
        line += f"| {name} {us:6.1f} "
    # the other two GEMMs of the layer for scale
    us_f = timeit(lambda: x @ w.t()); us_d = timeit(lambda: dy @ w)
    print(line + f"| fwd {us_f:6.1f} dgrad {us_d:6.1f}", flush=True)
