"""Data-pipeline pixel work for one batch (24 VOC-sized images -> 513x513): device path vs the reference's host path
(Pillow crop/resize/flip + torch ToTensor/Normalize + the per-pixel label lambda), same random parameters."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import datapipe as OD
from ucd_amd import tasks
from ucd_amd.datapipe import DeviceImagePath, DeviceLabelPath, random_resized_crop_params, target_lut
dev = torch.device("cuda:0")
B, S = 24, 513
rng = np.random.RandomState(0); random.seed(0)
imgs = [rng.randint(0, 256, size=(375, 500, 3)).astype(np.uint8) for _ in range(B)]
labs = [rng.choice([0, 5, 15, 16, 18, 255], size=(375, 500)).astype(np.uint8) for _ in range(B)]
boxes = [random_resized_crop_params(375, 500) for _ in range(B)]
flips = [random.random() < 0.5 for _ in range(B)]
labels, labels_old, _ = tasks.get_task_labels("voc", "15-5", 1)
lut = target_lut(labels, labels_old)
d_imgs = [torch.from_numpy(a).to(dev) for a in imgs]; d_labs = [torch.from_numpy(a).to(dev) for a in labs]
ip, lp = DeviceImagePath(S), DeviceLabelPath(S, lut)
for _ in range(3): x = ip(d_imgs, boxes, flips); y = lp(d_labs, boxes, flips)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20): x = ip(d_imgs, boxes, flips); y = lp(d_labs, boxes, flips)
torch.cuda.synchronize(); gpu_ms = (time.perf_counter() - t) / 20 * 1e3
t = time.perf_counter()
for n in range(4):                                          # 4 images on the host, scaled to the batch
    xi = OD.image_path_pil(imgs[n], boxes[n], S, flips[n])
    yi = OD.label_path_pil(labs[n], boxes[n], S, flips[n], np.arange(256, dtype=np.uint8))
    t0 = torch.from_numpy(yi.astype(np.int64)); lam = {int(i): int(v) for i, v in enumerate(lut.tolist())}
    t0.apply_(lambda v: lam[v])                             # the reference's per-pixel Python lambda (voc.py:199-203)
cpu_ms = (time.perf_counter() - t) / 4 * B * 1e3
ok = np.array_equal(np.ascontiguousarray(x[0].cpu().numpy()), OD.image_path_pil(imgs[0], boxes[0], S, flips[0]))
print(f"device: {gpu_ms:.2f} ms per batch of {B} ({B / gpu_ms * 1e3:.0f} img/s); host (1 process, Pillow + per-pixel lambda): {cpu_ms:.0f} ms per batch ({B / cpu_ms * 1e3:.1f} img/s); first image bit-equal: {ok}")
