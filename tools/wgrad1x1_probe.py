"""Narrow 1x1 convolutions: MIOpen's weight-gradient solver vs the split-K batched GEMM on the row matrices."""
import torch
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
def timeit(f, n=20):
    for _ in range(5): f()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in evs:
        s.record(); f(); e.record()
    torch.cuda.synchronize()
    return sorted(s.elapsed_time(e) for s, e in evs)[n // 2] * 1e3
B = 24
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
for (ci, co, hw) in [(64, 256, 129), (256, 64, 129), (128, 512, 65), (512, 128, 65), (256, 128, 129), (512, 256, 65), (64, 64, 129)]:
    x = cl(torch.randn(B, ci, hw, hw, device=dev, dtype=torch.bfloat16))
    w = cl(torch.randn(co, ci, 1, 1, device=dev, dtype=torch.bfloat16) * 0.05)
    dy = cl(torch.randn(B, co, hw, hw, device=dev, dtype=torch.bfloat16))
    wrw = lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    M = B * hw * hw
    xr = x.permute(0, 2, 3, 1).reshape(M, ci); dyr = dy.permute(0, 2, 3, 1).reshape(M, co)
    line = f"{ci:4d}->{co:4d} {hw}^2: MIOpen wrw {timeit(wrw):6.1f} |"
    ref = wrw().float().view(co, ci)
    for S in (8, 24, 48, 72, 96):
        if M % S: continue
        f = lambda: torch.bmm(dyr.view(S, M // S, co).transpose(1, 2), xr.view(S, M // S, ci)).sum(0)
        err = ((f().float() - ref).norm() / ref.norm()).item()
        line += f" bmm{S} {timeit(f):6.1f}"
    print(line + f" (err {err:.0e})", flush=True)
