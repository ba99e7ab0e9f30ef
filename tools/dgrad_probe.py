"""Input gradient of the stride-1 3x3 convolutions: MIOpen's backward-data solver vs the FORWARD solver applied to the
flipped / transposed weight (dx = conv2d(dy, w.flip(2,3).transpose(0,1), padding=d, dilation=d)); bf16, channels-last."""
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
for (ci, co, hw, d) in [(256, 256, 33, 1), (512, 512, 33, 2), (128, 128, 65, 1), (64, 64, 129, 1), (2048, 256, 33, 6), (2048, 256, 33, 12)]:
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    x = cl(torch.randn(B, ci, hw, hw, device=dev, dtype=torch.bfloat16))
    w = cl(torch.randn(co, ci, 3, 3, device=dev, dtype=torch.bfloat16) * 0.05)
    dy = cl(torch.randn(B, co, hw, hw, device=dev, dtype=torch.bfloat16))
    bwd = lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [d, d], [d, d], False, [0, 0], 1, [True, False, False])[0]
    wrw = lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [d, d], [d, d], False, [0, 0], 1, [False, True, False])[1]
    wt = cl(w.flip(2, 3).transpose(0, 1))
    fwd_as_bwd = lambda: F.conv2d(dy, wt, padding=d, dilation=d)
    mk_wt = lambda: cl(w.flip(2, 3).transpose(0, 1))
    fwd = lambda: F.conv2d(x, w, padding=d, dilation=d)
    ref = bwd().float(); got = fwd_as_bwd().float()
    err = ((ref - got).norm() / ref.norm()).item()
    print(f"{ci:4d}->{co:4d} {hw}^2 d={d}: fwd {timeit(fwd):6.1f} us | dgrad MIOpen {timeit(bwd):6.1f} us | dgrad via fwd kernel {timeit(fwd_as_bwd):6.1f} us (+ weight transform {timeit(mk_wt):5.1f}) | wgrad {timeit(wrw):6.1f} | rel diff {err:.1e}", flush=True)
