"""Narrow 1x1 convolutions (MIOpen path): backward-data solver vs forward solver on the transposed weight."""
import torch, torch.nn.functional as F
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
    bwd = lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])[0]
    wrw = lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    wt = cl(w.transpose(0, 1).contiguous())
    via = lambda: F.conv2d(dy, wt)
    fwd = lambda: F.conv2d(x, w)
    print(f"{ci:4d}->{co:4d} {hw}^2: fwd {timeit(fwd):6.1f} | dgrad MIOpen {timeit(bwd):6.1f} | dgrad via fwd solver {timeit(via):6.1f} | wgrad {timeit(wrw):6.1f}", flush=True)
