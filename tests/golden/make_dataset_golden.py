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
