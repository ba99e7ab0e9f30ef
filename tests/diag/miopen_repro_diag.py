"""Are the library convolutions the step still calls bit-reproducible from call to call (same process, same inputs, other work in
between)?  The global-pooling branch (1 x 1 maps), the stem weight gradient, the strided layers' input gradients.
usage: python tests/diag/miopen_repro_diag.py"""
import torch
import torch.nn.functional as F
dev = torch.device("cuda:0")
torch.backends.cudnn.deterministic = True
g = torch.Generator(dev).manual_seed(1)
cl = torch.channels_last
def rnd(*s): return torch.randn(*s, device=dev, generator=g).bfloat16().contiguous(memory_format=cl)
cases = {
    "pool conv 2048->256 on 1x1 maps (dx, dw)": (rnd(24, 2048, 1, 1), rnd(256, 2048, 1, 1), 1, 0, 1),
    "pool red conv 256->256 on 1x1 maps": (rnd(24, 256, 1, 1), rnd(256, 256, 1, 1), 1, 0, 1),
    "stem 7x7 s2 (dw)": (rnd(24, 3, 257, 257), rnd(64, 3, 7, 7), 2, 3, 1),
    "mod3 3x3 s2 128 (dx)": (rnd(24, 128, 129, 129), rnd(128, 128, 3, 3), 2, 1, 1),
    "mod3 proj 1x1 s2 256->512 (dx)": (rnd(24, 256, 129, 129), rnd(512, 256, 1, 1), 2, 0, 1),
}
junk = torch.empty(1 << 26, device=dev)
for name, (x, w, s, p, d) in cases.items():
    y = F.conv2d(x, w, None, s, p, d)
    dy = torch.randn(y.shape, device=dev, generator=g).bfloat16().contiguous(memory_format=cl)
    def once():
        return torch.ops.aten.convolution_backward(dy, x, w, None, [s, s], [p, p], [d, d], False, [0, 0], 1, [True, True, False])[:2]
    dx0, dw0 = once()
    y0 = F.conv2d(x, w, None, s, p, d)
    nx = nw = ny = 0
    for i in range(10):
        if i % 2:
            junk.normal_(); (junk[: 1 << (16 + i)] * 2).sum()
        dx1, dw1 = once()
        nx += int(not torch.equal(dx0, dx1)); nw += int(not torch.equal(dw0, dw1))
        ny += int(not torch.equal(y0, F.conv2d(x, w, None, s, p, d)))
    print(f"{name:45s} forward differs {ny}/10, dx differs {nx}/10, dw differs {nw}/10", flush=True)
