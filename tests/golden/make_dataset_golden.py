"""Golden index lists of the reference's ``filter_images`` (dataset/utils.py:19-42), produced by importing that function by
file path (numpy + torch only) and running it on the synthetic label maps that ``tests/test_dataset.py::_make_tree`` writes.
Run in the build container:  python tests/golden/make_dataset_golden.py   ->  tests/golden/dataset_filter.npz
Only numbers are stored (the index lists); the reference's source is not copied."""
import importlib.util
import io
import contextlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))
from ucd_amd import tasks  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_dataset_utils", "/root/reference/dataset/utils.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

CLASS_SETS = [[0, 1, 5], [0, 16], [0, 3, 17, 255], [0, 20, 255], [0, 2], [0, 15, 19], [0, 18, 16, 4], [0, 6, 255], [0, 17], [0, 20, 1]]


def label_maps(n=10, seed=7):
    """The label maps of tests/test_dataset.py::_make_tree (same RandomState draws, images included)."""
    rng = np.random.RandomState(seed)
    maps = []
    for k in range(n):
        H, W = int(rng.randint(90, 180)), int(rng.randint(90, 180))
        rng.randint(0, 256, size=(H // 6 + 1, W // 6 + 1, 3))
        maps.append(rng.choice(CLASS_SETS[k % len(CLASS_SETS)], size=(H // 10 + 1, W // 10 + 1)).astype(np.uint8)
                    .repeat(10, 0).repeat(10, 1)[:H, :W])
    return maps


CASES = [("voc", "15-5", 0), ("voc", "15-5", 1), ("voc", "15-5s", 3), ("voc", "19-1", 1), ("voc", "10-10", 1)]

if __name__ == "__main__":
    maps = label_maps()
    dataset = [(None, m) for m in maps]
    out = {}
    for ds, task, step in CASES:
        labels, labels_old, _ = tasks.get_task_labels(ds, task, step)
        for overlap in (True, False):
            with contextlib.redirect_stdout(io.StringIO()):
                idx = ref.filter_images(dataset, list(labels), list(labels_old), overlap=overlap)
            out[f"{task}::{step}::{int(overlap)}"] = np.array(idx, dtype=np.int64)
    np.savez(os.path.join(HERE, "dataset_filter.npz"), **out)
    for k, v in out.items():
        print(k, v.tolist())


# ---- ADE20K / Cityscapes: the reference's own dataset CLASSES on synthetic trees ------------------------------------------------
# (dataset/ade.py, dataset/cityscape.py need torchvision only for transforms.Lambda / ToPILImage: replaced by two-line stand-ins
# in this process).  Stored: the indices (ADE: sorted listing) or file names (Cityscapes: os.walk order is the file system's) the
# classes keep, their target transform evaluated on every label value 0 .. 255 (= the label table), and the raw-id -> class table
# of CitySegmentation._class_to_index.   ->  tests/golden/dataset_ade_city.npz
from dataset_trees import make_ade_tree, make_city_tree  # noqa: E402  (tests/dataset_trees.py: shared with tests/test_dataset.py)


def _import_reference_datasets():
    import types
    import torch
    from PIL import Image
    tv = types.ModuleType("torchvision"); tvt = types.ModuleType("torchvision.transforms")
    tvf = types.ModuleType("torchvision.transforms.functional")

    class Lambda:
        def __init__(self, fn):
            self.fn = fn

        def __call__(self, x):
            return self.fn(x)

    class ToPILImage:
        def __call__(self, arr):
            return Image.fromarray(np.asarray(arr))
    tvt.Lambda, tvt.ToPILImage, tvt.functional, tv.transforms = Lambda, ToPILImage, tvf, tvt
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.transforms.functional": tvf})
    sys.path.insert(0, "/root/reference")
    import dataset as ref_dataset
    return ref_dataset, torch


def gold_ade_city():
    import tempfile
    ref_dataset, torch = _import_reference_datasets()
    out = {}
    values = torch.arange(256, dtype=torch.long)
    with tempfile.TemporaryDirectory() as tmp:
        make_ade_tree(tmp)
        make_city_tree(tmp)
        for task, step in (("100-50", 0), ("100-50", 1), ("100-10", 2), ("50", 1)):
            labels, labels_old, _ = tasks.get_task_labels("ade", task, step)
            for train in (True, False):
                for overlap in (True, False):
                    with contextlib.redirect_stdout(io.StringIO()):
                        ds = ref_dataset.AdeSegmentationIncremental(tmp, train=train, labels=list(labels), labels_old=list(labels_old),
                                                                    idxs_path=None, masking=True, overlap=overlap)
                    key = f"ade::{task}::{step}::{int(train)}::{int(overlap)}"
                    out[key + "::idx"] = np.array(ds.dataset.indices, dtype=np.int64)
                    out[key + "::lut"] = ds.dataset.target_transform(values.clone()).numpy().astype(np.int64)
        for task, step in (("13-6", 0), ("13-6", 1), ("13-1", 3)):
            try:
                labels, labels_old, _ = tasks.get_task_labels("city", task, step)
            except (KeyError, AssertionError):
                continue
            for train in (True, False):
                for overlap in (True, False):
                    with contextlib.redirect_stdout(io.StringIO()):
                        ds = ref_dataset.CitySegmentationIncremental(tmp, train=train, labels=list(labels), labels_old=list(labels_old),
                                                                     idxs_path=None, masking=True, overlap=overlap)
                    key = f"city::{task}::{step}::{int(train)}::{int(overlap)}"
                    full = ds.dataset.dataset
                    kept = sorted(os.path.basename(full.images[i]) for i in ds.dataset.indices)
                    out[key + "::names"] = np.array([int(n.split("_")[1]) for n in kept], dtype=np.int64)   # the k of the file name
                    out[key + "::lut"] = ds.dataset.target_transform(values.clone()).numpy().astype(np.int64)
        with contextlib.redirect_stdout(io.StringIO()):
            full = ref_dataset.CitySegmentation(tmp, True)
        out["city::class_of_raw"] = full._class_to_index(np.arange(0, 34)).astype(np.int64)
    np.savez(os.path.join(HERE, "dataset_ade_city.npz"), **out)
    print("dataset_ade_city.npz:", len(out), "arrays;", {k: v.tolist() for k, v in out.items() if k.endswith("1::1::idx") or k.endswith("1::1::names")})


if __name__ == "__main__" and (len(sys.argv) < 2 or sys.argv[1] == "ade_city"):
    gold_ade_city()
