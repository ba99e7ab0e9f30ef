"""Operator-level CPU time of the step at a small per-rank batch (torch.profiler sees the autograd worker thread)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
sys.argv = ["bench.py", "--global_batch", "3"]
args = bench.parse()
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
torch.backends.cudnn.benchmark = True
trainer, optimizer, scheduler, images, labels, classes = bench.build(args, dev, 3, 0)
for _ in range(6): trainer.train_step(images, labels, optimizer, scheduler)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(3): trainer.train_step(images, labels, optimizer, scheduler)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=32, max_name_column_width=48))
