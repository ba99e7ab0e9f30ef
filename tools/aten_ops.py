"""GPU diagnostic: which ATen operators still run inside the benchmark step (torch.profiler, grouped by operator and input
shapes, CUDA time).  usage: python tools/aten_ops.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity

sys.argv = [sys.argv[0]]
args = bench.parse()
dev = torch.device("cuda:0")
trainer, optimizer, scheduler, images, labels, classes = bench.build(args, dev, args.global_batch, 0)
for _ in range(5):
    trainer.train_step(images, labels, optimizer, scheduler)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(2):
        trainer.train_step(images, labels, optimizer, scheduler)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    t = getattr(e, "self_device_time_total", None)
    if t is None:
        t = e.self_cuda_time_total
    if t > 0 and e.key.startswith("aten::"):
        rows.append((t / 2e3, e.count // 2, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
print("ms/step  calls/step  op  shapes")
for r in rows[:45]:
    print("%7.3f %5d  %-34s %s" % r)
print("total aten self device time ms/step: %.3f" % sum(r[0] for r in rows))
