"""Does the loader feed the step?    A synthetic VOC-sized tree (JPEG 500 x 375 + PNG label maps) through
ucd_amd.dataset.DeviceLoader: decode in N DataLoader worker processes, crop + Pillow-exact resize + flip + normalise + label
table on the device (batch 24, crop 513) - images per second against what one benchmark step consumes.
usage: python tools/loader_bench.py [step img/s] -> profiles/rNN_loader_bench.txt"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ucd_amd import tasks  # noqa: E402
from ucd_amd.dataset import DeviceBatcher, DeviceLoader, VOCSegmentationIncremental  # noqa: E402

step_rate = float(sys.argv[1]) if len(sys.argv) > 1 else 634.0
N = 480
dev = torch.device("cuda:0")
with tempfile.TemporaryDirectory() as root:
    from PIL import Image
    rng = np.random.RandomState(0)
    for d in ("splits", "JPEGImages", "SegmentationClassAug"):
        os.makedirs(os.path.join(root, d))
    lines = []
    for k in range(N):
        H, W = (375, 500) if k % 3 else (500, 375)
        base = rng.randint(0, 256, size=(H // 8 + 1, W // 8 + 1, 3)).astype(np.uint8).repeat(8, 0).repeat(8, 1)[:H, :W]
        img = np.clip(base.astype(np.int16) + rng.randint(-12, 13, size=base.shape), 0, 255).astype(np.uint8)   # photo-like entropy
        lab = rng.choice([0, 16, 17, 18, 19, 20, 255], size=(H // 25 + 1, W // 25 + 1)).astype(np.uint8).repeat(25, 0).repeat(25, 1)[:H, :W]
        Image.fromarray(img).save(os.path.join(root, "JPEGImages", f"im{k}.jpg"), quality=90)
        Image.fromarray(lab).save(os.path.join(root, "SegmentationClassAug", f"im{k}.png"))
        lines.append(f"/JPEGImages/im{k}.jpg /SegmentationClassAug/im{k}.png\n")
    open(os.path.join(root, "splits", "train_aug.txt"), "w").write("".join(lines))
    labels, labels_old, _ = tasks.get_task_labels("voc", "15-5", 1)
    ds = VOCSegmentationIncremental(root, train=True, labels=list(labels), labels_old=list(labels_old), overlap=True)
    jpeg_kb = sum(os.path.getsize(os.path.join(root, "JPEGImages", f)) for f in os.listdir(os.path.join(root, "JPEGImages"))) / N / 1024
    print(f"# {len(ds)} of {N} synthetic VOC-sized images kept (mean JPEG {jpeg_kb:.0f} KiB), batch 24, crop 513, host cores {os.cpu_count()}")
    print(f"# the benchmark step consumes {step_rate:.0f} img/s on one MI355X")
    batcher = DeviceBatcher(dev, 513, ds.lut, train=True)
    for workers in (0, 4, 8, 16, 32):
        if workers > (os.cpu_count() or 8):
            continue
        loader = DeviceLoader(ds, 24, torch.utils.data.RandomSampler(ds), batcher, num_workers=workers, drop_last=True)
        for epoch in range(3):          # epoch 0: worker start-up + page cache
            torch.cuda.synchronize()
            t0 = time.time()
            n = 0
            for images, lab in loader:
                n += images.shape[0]
            torch.cuda.synchronize()
            dt = time.time() - t0
        print(f"workers {workers:2d}: {n / dt:8.1f} img/s  ({n / dt / step_rate:5.2f} x the step)")
        del loader
