import torch, sys
sys.path.insert(0, "/root/repo")
from ucd_amd import abn as _abn, hip
DEV = torch.device("cuda:0")
node = _abn._abn_node()
g = torch.Generator(DEV).manual_seed(5)
x = torch.randn(3, 256, 33, 33, device=DEV, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
ws = [(torch.randn(256, 256, 3, 3, device=DEV, generator=g) * 0.02).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_() for _ in range(3)]
def grads():
    y = x
    for w in ws:
        y = node.conv_stride1(y, w, 2, None, True, False, hip.stream(), True)
    return torch.autograd.grad(y.float().square().mean(), ws)
want = [t.clone() for t in grads()]
want2 = [t.clone() for t in grads()]
torch.cuda.synchronize()
print("mode0 repeat equal", [torch.equal(a, b) for a, b in zip(want, want2)], "flushes", node.pass_flushes())
for mode in (1, 3):
    hip.wgrad_defer(mode)
    got = grads()
    torch.cuda.synchronize()
    print("mode", mode, "flushes", node.pass_flushes(), [(torch.equal(a, b), float((a.float() - b.float()).abs().max()), bool(torch.isnan(a.float()).any())) for a, b in zip(got, want)])
    hip.wgrad_drop(); hip.wgrad_defer(0)
