"""ucd_conv_wgrad (csrc/wgrad.hip) against what the step used before it - MIOpen's weight-gradient solvers (3x3 and narrow 1x1)
and eight batched split-M library products + a sum (wide 1x1) - on the student's layer shapes, B = 24 at 513^2.
usage: python tools/wgrad_probe2.py [target workgroups ...]   (UCD_WGRAD_TARGET sweep)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ucd_amd import hip  # noqa: E402

torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
LAYERS = [  # name, B, Cin, Cout, H, dilation (0 = 1x1), count per step
    ("mod4 3x3 256->256", 24, 256, 256, 33, 1, 22), ("mod4 1x1 1024->256", 24, 1024, 256, 33, 0, 22), ("mod4 1x1 256->1024", 24, 256, 1024, 33, 0, 23),
    ("mod5 3x3 512->512 d2", 24, 512, 512, 33, 2, 3), ("mod5 1x1 2048->512", 24, 2048, 512, 33, 0, 2), ("mod5 1x1 512->2048", 24, 512, 2048, 33, 0, 3),
    ("aspp 3x3 2048->256 d12", 24, 2048, 256, 33, 12, 3), ("aspp 1x1 2048->256", 24, 2048, 256, 33, 0, 1), ("red 1x1 1024->256", 24, 1024, 256, 33, 0, 1),
    ("mod3 3x3 128->128", 24, 128, 128, 65, 1, 3), ("mod3 1x1 512->128", 24, 512, 128, 65, 0, 3), ("mod3 1x1 128->512", 24, 128, 512, 65, 0, 4),
    ("mod2 3x3 64->64", 24, 64, 64, 129, 1, 3), ("mod2 1x1 256->64", 24, 256, 64, 129, 0, 2), ("mod2 1x1 64->256", 24, 64, 256, 129, 0, 4),
]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


targets = [int(v) for v in sys.argv[1:]] or [512]
tot_lib, tot_own = 0.0, {t: 0.0 for t in targets}
print("%-26s %10s | %s" % ("layer", "library us", "  ".join("own@%d us" % t for t in targets)))
for name, B, ci, co, H, d, cnt in LAYERS:
    x = torch.randn(B, ci, H, H, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    dz = torch.randn(B, co, H, H, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    k = 3 if d else 1
    w = torch.randn(co, ci, k, k, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    M = B * H * H
    xr, zr = x.permute(0, 2, 3, 1).reshape(M, ci), dz.permute(0, 2, 3, 1).reshape(M, co)
    pad = d if d else 0
    if d or min(ci, co) < 256:
        lib = lambda: torch.ops.aten.convolution_backward(dz, x, w, None, [1, 1], [pad, pad], [max(d, 1)] * 2, False, [0, 0], 1, [False, True, False])[1]
    else:
        S = 8
        lib = lambda: torch.bmm(zr.view(S, M // S, co).transpose(1, 2), xr.view(S, M // S, ci)).sum(0)
    t_lib = timeit(lib)
    dw = torch.empty(co, k * k * ci, device=dev, dtype=torch.bfloat16)
    row = []
    for t in targets:
        os.environ["UCD_WGRAD_TARGET"] = str(t)
        own = lambda: hip.conv_wgrad(zr, xr, dw, conv3=(H, H, d) if d else None)
        t_own = timeit(own)
        row.append(t_own)
        tot_own[t] += t_own * cnt
    ref = lib().float()
    ref = ref.permute(0, 2, 3, 1).reshape(co, -1) if ref.dim() == 4 else ref
    err = ((dw.float() - ref).norm() / ref.norm()).item()
    tot_lib += t_lib * cnt
    flop = 2.0 * M * k * k * ci * co
    print("%-26s %10.1f | %s   (best %.0f TF/s, vs library rel %.1e)" % (name, t_lib, "  ".join("%9.1f" % v for v in row), flop / min(row) / 1e6, err))
print("per step (counts applied): library %.2f ms | own %s" % (tot_lib / 1e3, "  ".join("@%d %.2f ms" % (t, v / 1e3) for t, v in tot_own.items())))
