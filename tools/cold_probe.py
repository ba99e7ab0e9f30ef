"""GPU probe (round 6): what a kernel of the 3-image step costs BETWEEN other kernels and on fresh buffers, against the same kernel
launched back to back on one buffer set.  Run under `rocprofv3 --kernel-trace`; tools/cold_probe_parse.py reads the trace.
Phases (separated by 50 ms of idle): A same kernel, same buffers; B same kernel, rotating activation buffers; C rotating
activations and weights; D same buffers, nine other kernels of the library between two launches; E = D + rotating buffers.
usage: rocprofv3 --kernel-trace -d /tmp/cp -o t --output-format csv -- python3 tools/cold_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ucd_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
B, K, N, H, d = 3, 256, 256, 33, 1
M = B * H * H
NB = 48
cl = torch.channels_last


def rows(t):
    b, c, h, w = t.shape
    return t.permute(0, 2, 3, 1).reshape(b * h * w, c)


xs = [torch.randn(B, K, H, H, device=dev).bfloat16().contiguous(memory_format=cl) for _ in range(NB)]
ys = [torch.empty(B, N, H, H, device=dev, dtype=torch.bfloat16).contiguous(memory_format=cl) for _ in range(NB)]
ws = [(torch.randn(N, K, 3, 3, device=dev) * 0.02).bfloat16().contiguous(memory_format=cl).permute(0, 2, 3, 1).reshape(N, 9 * K) for _ in range(NB)]
pad = [torch.empty(64 << 20, device=dev, dtype=torch.uint8) for _ in range(4)]     # spread the buffers over the address space
part = hip.conv1x1_stats_partial(M, N, dev)


def k3(i, j):
    hip.conv1x1(rows(xs[i]), ws[j], rows(ys[i]), conv3=(H, H, d), out_mode=2, partial=part)


# the "other kernels": 1x1 products in several epilogue forms, ABN passes, a weight gradient - all on small operands of their own
a1 = torch.randn(M, 1024, device=dev).bfloat16()
w1 = (torch.randn(256, 1024, device=dev) * 0.03).bfloat16()
y1 = torch.empty(M, 256, device=dev, dtype=torch.bfloat16)
w2 = (torch.randn(1024, 256, device=dev) * 0.03).bfloat16()
y2 = torch.empty(M, 1024, device=dev, dtype=torch.bfloat16)
res = torch.randn(M, 1024, device=dev).bfloat16()
v256, v1024 = torch.rand(256, device=dev) + 0.5, torch.rand(1024, device=dev) + 0.5
part1, part2 = hip.conv1x1_stats_partial(M, 256, dev), hip.conv1x1_stats_partial(M, 1024, dev)
dw = torch.empty(256, 1024, device=dev, dtype=torch.bfloat16)


def others():
    hip.conv1x1(a1, w1, y1)
    hip.conv1x1(a1, w1, y1, out_mode=2, partial=part1)
    hip.abn_apply(y1, 256, y1, 256, None, 0, M, 256, None, H * H, v256, v256, v256, hip.ACT_LEAKY_RELU, 0.01)
    hip.conv1x1(y1, w2, y2, out_mode=1, out_norm=(v1024, v1024, v1024, None, 1, 0.01), residual=res)
    hip.conv1x1(y1, w2, y2, out_mode=2, partial=part2)
    hip.abn_apply(y2, 1024, y2, 1024, res, 1024, M, 1024, None, H * H, v1024, v1024, v1024, hip.ACT_LEAKY_RELU, 0.01)
    hip.conv1x1(y2, w1, y1, accumulate=True)
    hip.conv_wgrad(y1, a1, dw=dw)
    torch.add(y2, res, out=y2)


def phase(fn, n=96):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(0)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for i in range(n):
                fn(i)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    time.sleep(0.05)


k3(0, 0)
others()
torch.cuda.synchronize()
time.sleep(0.05)
phase(lambda i: k3(0, 0))
phase(lambda i: k3(i % NB, 0))
phase(lambda i: k3(i % NB, i % NB))
phase(lambda i: (others(), k3(0, 0)))
phase(lambda i: (others(), k3(i % NB, i % NB)))
print("done")
