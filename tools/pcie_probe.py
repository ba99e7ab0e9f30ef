"""Step rate when the boundary hands over HOST buffers (pinned, non_blocking copies) instead of device-resident ones."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
sys.argv = ["bench.py"]
args = bench.parse()
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
torch.backends.cudnn.benchmark = True
trainer, optimizer, scheduler, images, labels, classes = bench.build(args, dev, 24, 0)
h_img = images.cpu().contiguous().pin_memory(); h_lab = labels.cpu().pin_memory()
def run(i, l, n):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): trainer.train_step(i, l, optimizer, scheduler)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
run(images, labels, 6)
print("device-resident inputs: %.2f ms/step" % run(images, labels, 15))
run(h_img, h_lab, 3)
print("pinned host inputs    : %.2f ms/step (%.0f MB per step over PCIe)" % (run(h_img, h_lab, 15),
      (h_img.numel() * 4 + h_lab.numel() * h_lab.element_size()) / 1e6))
print("peak memory allocated : %.1f GB" % (torch.cuda.max_memory_allocated() / 1e9))
