"""ucd_gemm_bf16 (hipBLASLt, tuned once per shape) against torch for the 1x1-conv GEMM shapes: result, GPU time, host time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ucd_amd import hip
dev = torch.device("cuda:0")
assert hip.gemm_available()
lib = hip.load()
def gpu_us(f, n=20):
    for _ in range(3): f()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in evs:
        s.record(); f(); e.record()
    torch.cuda.synchronize()
    return sorted(s.elapsed_time(e) for s, e in evs)[n // 2] * 1e3
def host_us(f, n=300):
    for _ in range(10): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    dt = time.perf_counter() - t; torch.cuda.synchronize()
    return dt / n * 1e6
for B in (24, 3):
    Mr = B * 33 * 33
    for co, ci in ((1024, 256), (256, 1024), (2048, 512), (512, 2048), (2048, 1024), (256, 2048)):
        x = torch.randn(Mr, ci, device=dev, dtype=torch.bfloat16); w = torch.randn(co, ci, device=dev, dtype=torch.bfloat16) * 0.05
        dy = torch.randn(Mr, co, device=dev, dtype=torch.bfloat16)
        y = torch.empty(Mr, co, device=dev, dtype=torch.bfloat16); dx = torch.empty(Mr, ci, device=dev, dtype=torch.bfloat16)
        dw = torch.empty(co, ci, device=dev, dtype=torch.bfloat16)
        line = f"B={B:2d} Co={co:4d} Ci={ci:4d} |"
        for name, f_ucd, f_ref, out in (("fwd", lambda: hip.gemm_bf16(0, x, w, y), lambda: x @ w.t(), y),
                                        ("dgrad", lambda: hip.gemm_bf16(1, dy, w, dx), lambda: dy @ w, dx),
                                        ("wgrad", lambda: hip.gemm_bf16(2, dy, x, dw), lambda: dy.t() @ x, dw)):
            f_ucd(); cand = lib.ucd_gemm_last_candidates()
            ref = f_ref().float(); err = ((out.float() - ref).norm() / ref.norm()).item()
            assert err < 1e-2, (name, err)
            line += f" {name}: ucd {gpu_us(f_ucd):6.1f} us (host {host_us(f_ucd):4.1f}) torch {gpu_us(f_ref):6.1f} (host {host_us(f_ref):4.1f}) cand {cand:2d} |"
        print(line, flush=True)
