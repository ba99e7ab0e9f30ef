"""GPU micro-benchmark of the contrastive loss kernels at the SURVEY 8-d micro-shapes.
usage: python tools/pixcon_bench.py [f16|f16_split|f32] [dom]   (cfg2: BHW=26136; "dom": one teacher class dominates, like the
benchmark step / a trained teacher - 2/3 of the pairs are positives)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ucd_amd import synth
from ucd_amd.contrastive import pixcon_prepare, pixcon_loss_raw
prec = sys.argv[1] if len(sys.argv) > 1 else "f16"
dev = torch.device("cuda:0")
B, N, h, K, H = 24, 256, 33, 16, 513
torch.manual_seed(0)
f_n = torch.randn(B, N, h, h, device=dev).abs_().contiguous(memory_format=torch.channels_last)   # post-activation-like features
f_o = (f_n + 0.3 * torch.randn_like(f_n)).contiguous(memory_format=torch.channels_last)
l_po = 2 * torch.randn(B, K, h, h, device=dev)
if "dom" in sys.argv[2:]:
    l_po[:, 9] += 6.0
labels = synth.seg_labels(7, B, H, H, range(16, 21)).to(dev)
pb = pixcon_prepare(f_n, labels, l_po, f_o, sort_by_label=True, fp16=prec.startswith("f16"))
m = pb.meta_host()
print("A", m.A, "Co", m.Co, "Cpad", m.Cpad, "n_valid", m.n_valid)
def run():
    return pixcon_loss_raw(pb, 0.07, True, True, need_grad=True, precision=prec)
for _ in range(3): out = run()
torch.cuda.synchronize()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
for s, e in evs:
    s.record(); out = run(); e.record()
torch.cuda.synchronize()
ts = sorted(s.elapsed_time(e) for s, e in evs)
flop = float(m.A) * (m.A + m.Co) * (4 * N + 2 * K)
print(f"{prec}: median {ts[5]:.3f} ms  min {ts[0]:.3f} ms  -> {flop / ts[5] / 1e9:.1f} TFLOP/s algorithmic; loss {out[0][0].item():.5f}")
if prec == "f16":   # the plan the device built (header of the workspace, pixcon_loss_f16p.hip)
    from ucd_amd import hip
    ws = hip.workspace(hip.load().ucd_pixcon_loss_workspace_bytes(pb.BHW, pb.N, pb.K), dev, "pixloss")
    hdr = ws[:32].view(torch.int32).cpu().tolist()
    print("plan: counters", hdr[0:2], "units", hdr[2:4], "tiles per unit", hdr[4:6], "anchor blocks", hdr[6])
