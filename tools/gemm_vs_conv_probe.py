"""GPU probe (not part of the product): 1x1 convolutions of the ResNet-101/DeepLab-V3 step as MIOpen
convolutions vs plain GEMMs on the channels-last row matrix (hipBLASLt through torch.mm), bf16, fwd+bwd."""
import time, torch, torch.nn.functional as F
dev = torch.device("cuda:0")
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
B = 24
shapes = [(129, 64, 64), (129, 64, 256), (129, 256, 64), (65, 512, 128), (65, 128, 512), (33, 1024, 256), (33, 256, 1024),
          (33, 2048, 512), (33, 512, 2048), (33, 1024, 2048), (33, 2048, 256), (33, 1024, 256)]
for (hw, cin, cout) in shapes:
    x = torch.randn(B, cin, hw, hw, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(cout, cin, 1, 1, device=dev, dtype=torch.bfloat16) * 0.05).requires_grad_(True)
    g = torch.randn(B, cout, hw, hw, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    def conv():
        y = F.conv2d(x, w); y.backward(g); x.grad = None; w.grad = None
    xr = x.detach().permute(0, 2, 3, 1).reshape(-1, cin).requires_grad_(True)
    wr = w.detach().reshape(cout, cin).requires_grad_(True)
    gr = g.permute(0, 2, 3, 1).reshape(-1, cout)
    def gemm():
        y = xr @ wr.t(); y.backward(gr); xr.grad = None; wr.grad = None
    tc, tg = bench(conv), bench(gemm)
    flop = 3 * 2 * B * hw * hw * cin * cout
    print(f"hw={hw:4d} cin={cin:5d} cout={cout:5d}  conv {tc*1e3:7.3f} ms ({flop/tc/1e12:6.1f} TF/s)   gemm {tg*1e3:7.3f} ms ({flop/tg/1e12:6.1f} TF/s)", flush=True)
# 3x3 convs for reference
for (hw, c, dil, stride) in [(129, 64, 1, 1), (65, 128, 1, 1), (33, 256, 1, 1), (33, 512, 2, 1), (33, 2048, 12, 1)]:
    cout = c if c < 2048 else 256
    x = torch.randn(B, c, hw, hw, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(cout, c, 3, 3, device=dev, dtype=torch.bfloat16) * 0.02).requires_grad_(True)
    y0 = F.conv2d(x, w, padding=dil, dilation=dil)
    g = torch.randn_like(y0)
    def conv3():
        y = F.conv2d(x, w, padding=dil, dilation=dil); y.backward(g); x.grad = None; w.grad = None
    t = bench(conv3, 10)
    flop = 3 * 2 * B * hw * hw * c * cout * 9
    print(f"3x3 hw={hw} c={c}->{cout} dil={dil}: {t*1e3:7.3f} ms ({flop/t/1e12:6.1f} TF/s)", flush=True)
