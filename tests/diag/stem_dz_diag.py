"""Where does the stem's d z (csrc/stem.hip: norm + max pool backward in one pass) leave a float64 reference computed from the SAME
stored bf16 convolution output?  (round 6: tests/test_conv1x1_fused_gpu.py::test_bench_shape_stem_against_float64 measured 3 %)
usage: python tests/diag/stem_dz_diag.py [B] [H]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from functools import partial
from ucd_amd import abn, synth
from oracle import model as OM

DEV = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
H = int(sys.argv[2]) if len(sys.argv) > 2 else 65
slope = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
bn = abn.InPlaceABNSync(64, activation="leaky_relu", activation_param=slope).to(DEV).train()
with torch.no_grad():
    bn.weight.copy_(torch.rand(64, device=DEV) + 0.5); bn.bias.copy_(torch.randn(64, device=DEV) * 0.1)
z = (synth.t_normal(3, (B, 64, H, H), stream=1) * 1.3 + 0.2).to(DEV).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
y = abn.stem_norm_pool(bn, z)
assert y is not None
dy = synth.t_normal(4, tuple(y.shape), stream=1).to(DEV).bfloat16().contiguous(memory_format=torch.channels_last)
y.backward(dy)
dz = z.grad.float()
P = {"bn.weight": bn.weight.detach().double().requires_grad_(True), "bn.bias": bn.bias.detach().double().requires_grad_(True),
     "bn.running_mean": torch.zeros(64, device=DEV, dtype=torch.double), "bn.running_var": torch.ones(64, device=DEV, dtype=torch.double)}
zin = z.detach().double().requires_grad_(True)
a = OM.abn(zin, P, "bn", True, slope=slope)
y64 = F.max_pool2d(a, 3, stride=2, padding=1)
y64.backward(dy.double())
ref = zin.grad.float()
rel = lambda u, v: ((u - v).norm() / v.norm()).item()
print("y rel", rel(y.float(), y64.float()), "dz rel", rel(dz, ref), "d weight", rel(bn.weight.grad.float(), P["bn.weight"].grad.float()),
      "d bias", rel(bn.bias.grad.float(), P["bn.bias"].grad.float()))
d = (dz - ref).abs()
thr = 4 * 2.0 ** -9 * ref.abs().clamp_min(ref.abs().mean())
bad = d > thr
print("elements off by more than 4 bf16 ulp: %d of %d (%.4f %%)" % (bad.sum().item(), bad.numel(), 100.0 * bad.float().mean().item()))
yy, xx = torch.meshgrid(torch.arange(H, device=DEV), torch.arange(H, device=DEV), indexing="ij")
border = ((yy == 0) | (xx == 0) | (yy == H - 1) | (xx == H - 1))[None, None].expand_as(bad)
print("  of them on the map's border:", (bad & border).sum().item(), " border elements:", border.sum().item())
# are the bad elements ones whose window maximum is tied (another element of a window with the same bf16 z)?
zz = z.detach().float()
mx = F.max_pool2d(zz, 3, stride=2, padding=1)
up = F.interpolate(mx, size=(H + (H % 2 == 0), H + (H % 2 == 0)), mode="nearest")[..., :H, :H] if False else None
cnt = F.avg_pool2d((F.pad(zz, (1, 1, 1, 1), value=float("-inf")).unfold(2, 3, 2).unfold(3, 3, 2) ==
                    mx[..., None, None]).float().sum((-1, -2)), 1)
print("windows with a tied maximum: %.4f %%" % (100.0 * (cnt > 1).float().mean().item()))
# contribution of the pooled-gradient routing alone: compare d a (gradient w.r.t. the normalised map) through the reference's max pool
big = d.flatten().topk(5)
for v, i in zip(big.values.tolist(), big.indices.tolist()):
    idx = torch.unravel_index(torch.tensor(i), dz.shape)
    print("   |diff| %.5f at (b, c, y, x) = %s: dz %.5f ref %.5f" % (v, [int(t) for t in idx], dz.flatten()[i].item(), ref.flatten()[i].item()))
# the windows of the largest differences: z values and who won
for v, i in list(zip(big.values.tolist(), big.indices.tolist()))[:2]:
    b, c, yy0, xx0 = [int(t) for t in torch.unravel_index(torch.tensor(i), dz.shape)]
    print("z around (%d, %d), channel %d:" % (yy0, xx0, c))
    print(zz[b, c, max(0, yy0 - 2):yy0 + 3, max(0, xx0 - 2):xx0 + 3])
    print("a64:"); print(a.detach()[b, c, max(0, yy0 - 2):yy0 + 3, max(0, xx0 - 2):xx0 + 3])
    print("dz ours:"); print(dz[b, c, max(0, yy0 - 2):yy0 + 3, max(0, xx0 - 2):xx0 + 3])
    print("dz ref:"); print(ref[b, c, max(0, yy0 - 2):yy0 + 3, max(0, xx0 - 2):xx0 + 3])
