"""Golden vectors of the label path (crop + Pillow NEAREST resize + flip + step remapping), produced BY PILLOW in the build
container (torchvision's resized_crop / hflip are thin wrappers over these PIL calls) and by the literal lambda of
dataset/voc.py:176-203 on the task tables of the reference's tasks.py.  usage: python tests/golden/make_datapipe_golden.py"""
import os, sys, zlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, "/root/reference")
import tasks as ref_tasks                                  # the reference's own task tables
from oracle.datapipe import label_path_pil, image_path_pil

HERE = os.path.dirname(os.path.abspath(__file__))
out = {}
# ---- step remapping: the reference's construction, executed literally (voc.py:143-203) ----
for name, (dataset, task, step, dm) in {"voc_15-5_s1": ("voc", "15-5", 1, "current"), "voc_15-5s_s3": ("voc", "15-5s", 3, "current"),
                                        "voc_19-1_s1_old": ("voc", "19-1", 1, "current+old"), "voc_10-10_s1_new": ("voc", "10-10", 1, "new")}.items():
    labels, labels_old, _ = ref_tasks.get_task_labels(dataset, task, step)
    labels = [l for l in labels if l != 0]; labels_old = [l for l in labels_old if l != 0]     # voc.py:143-147
    L = [0] + labels; LO = [0] + labels_old
    order = [0] + labels_old + labels
    inverted_order = {label: order.index(label) for label in order}
    inverted_order[255] = 255
    masking_value = 0
    if dm == "current": tmp_labels = L + [255]
    elif dm == "current+old": tmp_labels = labels_old + L + [255]
    else: tmp_labels = L; masking_value = 255
    f = lambda x: inverted_order[x] if x in tmp_labels else masking_value
    out[f"lut::{name}"] = np.array([f(x) for x in range(256)], dtype=np.uint8)
    out[f"lutcfg::{name}"] = np.array([step, {"current": 0, "current+old": 1, "new": 2}[dm]])
# ---- crop + NEAREST resize + flip ----
rng = np.random.RandomState(2024)
lut = out["lut::voc_15-5_s1"]
cases = []
for k in range(12):
    H0, W0 = int(rng.randint(120, 501)), int(rng.randint(120, 501))
    lbl = rng.choice([0, 3, 7, 15, 16, 18, 20, 255], size=(H0 // 8 + 1, W0 // 8 + 1)).astype(np.uint8).repeat(8, 0).repeat(8, 1)[:H0, :W0]
    h, w = int(rng.randint(40, H0 + 1)), int(rng.randint(40, W0 + 1))
    i, j = int(rng.randint(0, H0 - h + 1)), int(rng.randint(0, W0 - w + 1))
    S = [33, 64, 129, 513][k % 4]
    flip = int(k % 3 == 0)
    res = label_path_pil(lbl, (i, j, h, w), S, flip, lut)
    cases.append([H0, W0, i, j, h, w, flip, S])
    out[f"case{k}::crc"] = np.array([zlib.crc32(res.tobytes())], dtype=np.int64)
    if S <= 64:
        out[f"case{k}::out"] = res.astype(np.uint8)
out["cases"] = np.array(cases, dtype=np.int64)
# ---- image half: crop + Pillow BILINEAR resize + flip + ToTensor + Normalize (torch) ----
rng = np.random.RandomState(4048)
icases = []
for k in range(10):
    H0, W0 = int(rng.randint(100, 501)), int(rng.randint(100, 501))
    img = (rng.randint(0, 256, size=(H0 // 4 + 1, W0 // 4 + 1, 3)).repeat(4, 0).repeat(4, 1)[:H0, :W0]
           + rng.randint(-9, 10, size=(H0, W0, 3))).clip(0, 255).astype(np.uint8)
    h, w = int(rng.randint(30, H0 + 1)), int(rng.randint(30, W0 + 1))
    i, j = int(rng.randint(0, H0 - h + 1)), int(rng.randint(0, W0 - w + 1))
    S = [33, 64, 129, 513, 48][k % 5]
    flip = int(k % 2 == 1)
    res = image_path_pil(img, (i, j, h, w), S, flip)
    icases.append([H0, W0, i, j, h, w, flip, S])
    out[f"img{k}::crc"] = np.array([zlib.crc32(np.ascontiguousarray(res).tobytes())], dtype=np.int64)
    if S <= 48:
        out[f"img{k}::out"] = res
out["icases"] = np.array(icases, dtype=np.int64)
# ---- crop parameters: the reference's own RandomResizedCrop.get_params (dataset/transform.py:505-540), imported with an
# empty stand-in for the torchvision module it pulls in at import time (not installed here; get_params does not use it)
import importlib.util, random, types
sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))
sys.modules.setdefault("torchvision.transforms", types.ModuleType("torchvision.transforms"))
sys.modules.setdefault("torchvision.transforms.functional", types.ModuleType("torchvision.transforms.functional"))
spec = importlib.util.spec_from_file_location("ref_transform", "/root/reference/dataset/transform.py")
ref_transform = importlib.util.module_from_spec(spec); spec.loader.exec_module(ref_transform)


class _Size:
    def __init__(self, w, h): self.size = (w, h)


crops = []
for seed in range(40):
    H0, W0 = [(375, 500), (500, 333), (281, 500), (442, 500)][seed % 4]
    random.seed(seed)
    i, j, h, w = ref_transform.RandomResizedCrop.get_params(_Size(W0, H0), (0.5, 2.0), (3. / 4., 4. / 3.))
    crops.append([seed, H0, W0, i, j, h, w])
out["crop_params"] = np.array(crops, dtype=np.int64)
np.savez_compressed(os.path.join(HERE, "datapipe.npz"), **out)
print("wrote datapipe.npz:", len(cases), "cases")
