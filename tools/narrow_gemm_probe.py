"""Narrow 1x1 convolutions (the ones still routed to MIOpen): forward+backward through MIOpen vs through the tuned GEMM path."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ucd_amd.blocks import Conv1x1
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
def timeit(f, n=15):
    for _ in range(4): f()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in evs:
        s.record(); f(); e.record()
    torch.cuda.synchronize()
    return sorted(s.elapsed_time(e) for s, e in evs)[n // 2] * 1e3
B = 24
shapes = ((64, 256, 129), (256, 64, 129), (128, 512, 65), (512, 128, 65), (256, 128, 129), (512, 256, 65), (1024, 512, 33), (64, 64, 129))
if len(sys.argv) > 1 and sys.argv[1] == 'wide':
    shapes = ((1024, 256, 33), (256, 1024, 33), (2048, 512, 33), (512, 2048, 33), (1024, 2048, 33), (2048, 256, 33), (1024, 512, 33))
for ci, co, hw in shapes:
    res = []
    for as_gemm in (False, True):
        conv = Conv1x1(ci, co).to(dev).to(memory_format=torch.channels_last)
        conv.as_gemm = as_gemm
        x = torch.randn(B, ci, hw, hw, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        dy = torch.randn(B, co, hw, hw, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        def step():
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = conv(x)
            y.backward(dy)
            x.grad = None; conv.weight.grad = None
        res.append(timeit(step))
    print(f"{ci:4d}->{co:4d} at {hw}^2: MIOpen {res[0]:7.1f} us   GEMM path {res[1]:7.1f} us", flush=True)
