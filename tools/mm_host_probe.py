"""Host cost per torch.mm / F.linear call (enqueue-bound loop) for the 1x1-conv GEMM shapes at the 8-GPU per-rank batch,
per BLAS backend."""
import time, torch
dev = torch.device("cuda:0")
M = 3 * 33 * 33
shapes = [(1024, 256), (256, 1024), (2048, 512), (512, 2048)]
def host_us(f, n=300):
    for _ in range(20): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    dt = time.perf_counter() - t
    torch.cuda.synchronize(); t2 = time.perf_counter() - t
    return dt / n * 1e6, t2 / n * 1e6
for lib in ("cublaslt", "cublas"):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
    except Exception as e:
        print(lib, "unavailable", e); continue
    for co, ci in shapes:
        x = torch.randn(M, ci, device=dev, dtype=torch.bfloat16); w = torch.randn(co, ci, device=dev, dtype=torch.bfloat16)
        dy = torch.randn(M, co, device=dev, dtype=torch.bfloat16)
        a = host_us(lambda: x @ w.t()); b = host_us(lambda: dy @ w); c = host_us(lambda: dy.t() @ x)
        print(f"{lib:9s} Co={co:4d} Ci={ci:4d}  fwd host {a[0]:5.1f} us (total {a[1]:5.1f})  dgrad {b[0]:5.1f} ({b[1]:5.1f})  wgrad {c[0]:5.1f} ({c[1]:5.1f})", flush=True)
x = torch.randn(3, 256, 33, 33, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = torch.randn(256, 256, 3, 3, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
torch.backends.cudnn.benchmark = True
print("conv2d 3x3 host/total us:", host_us(lambda: torch.nn.functional.conv2d(x, w, padding=1)))
print("empty kernel-ish (add_) host/total us:", host_us(lambda: x.add_(1.0)))
