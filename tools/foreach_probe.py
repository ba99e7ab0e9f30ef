"""How many kernels does torch._foreach_copy_ launch for bf16 -> fp32 lists (same strides, channels-last 4-D views)?"""
import torch
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
shapes = [(256, 1024, 1, 1), (256, 256, 3, 3), (1024, 256, 1, 1), (64, 3, 7, 7)] * 26
def cl(t): return t.to(memory_format=torch.channels_last)
src = [cl(torch.randn(s, device=dev, dtype=torch.bfloat16)) for s in shapes]
n = sum(t.numel() for t in src)
flat = torch.zeros(n, device=dev)
dst, off = [], 0
for s in shapes:
    co, ci, kh, kw = s; k = co * ci * kh * kw
    dst.append(flat[off:off + k].view(co, kh, kw, ci).permute(0, 3, 1, 2)); off += k
for name, f in (("foreach_copy bf16->f32", lambda: torch._foreach_copy_(dst, src)),
                ("foreach_copy f32->f32", lambda: torch._foreach_copy_(dst, [d.clone() for d in dst][:len(dst)])),):
    f(); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        f(); torch.cuda.synchronize()
    ks = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    print(name, "kernels:", len(ks), "gpu us:", sum(e.device_time for e in ks) if hasattr(ks[0], "device_time") else "?")
ok = all(torch.equal(d, s.float()) for d, s in zip(dst, src)) if True else None
torch._foreach_copy_(dst, src); print("values ok:", all(torch.equal(d, s.float()) for d, s in zip(dst, src)))
import time
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20): torch._foreach_copy_(dst, src)
torch.cuda.synchronize(); print("foreach_copy_ mixed: %.1f us per call (104 tensors)" % ((time.perf_counter() - t) / 20 * 1e6))
