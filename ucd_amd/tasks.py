"""Class-split tables of the incremental-segmentation tasks.

Data table only: the hot path needs it to size the per-step 1x1 classifier heads
(VOC 15-5 -> [16, 5]; 15-5s -> [16, 1, 1, 1, 1, 1]; ADE 100-50 -> [101, 50];
Cityscapes 13-6 -> [14, 6]).  Same lookups as the reference's ``tasks.py:179-210``
(``get_task_list``, ``get_task_labels``, ``get_per_task_classes``), but the splits are
generated from (first-step size, increment) rules or from the primitive shuffled
groups instead of being spelled out step by step.
"""
from __future__ import annotations


def _contiguous(first: int, sizes, start: int = 0):
    """{step: [ids]} for contiguous class ids: step 0 gets ``first`` ids, later steps ``sizes``."""
    out, lo = {0: list(range(start, start + first))}, start + first
    for s, n in enumerate(sizes, 1):
        out[s] = list(range(lo, lo + n))
        lo += n
    return out


# ---- Pascal VOC (21 ids incl. background 0), reference tasks.py:1-55 -----------------
_VOC = {
    "offline": _contiguous(21, []),
    "19-1": _contiguous(20, [1]),
    "15-5": _contiguous(16, [5]),
    "15-5s": _contiguous(16, [1] * 5),
    "10-10": _contiguous(11, [10]),
    "10-10s": _contiguous(11, [1] * 10),
    "10-5-5": _contiguous(11, [5, 5]),
}
# 19-1b holds out class 5 instead of class 20 (reference tasks.py:11-15)
_VOC["19-1b"] = {0: [c for c in range(21) if c != 5], 1: [5]}

# ---- Cityscapes (20 ids), reference tasks.py:56-81 ------------------------------------
_CITY = {
    "offline": _contiguous(20, []),
    "17-2": _contiguous(18, [2]),
    "13-6": _contiguous(14, [6]),
    "13-6s": _contiguous(14, [1] * 6),
}

# ---- ADE20K (151 ids incl. 0), reference tasks.py:82-175 ------------------------------
# The "b"/"c" orders are two fixed shuffles of the 150 classes, published as groups of ten
# (steps 1..5 of 100-10b / 100-10c) and of fifty (steps 0..1 of 50b / 50c; the last fifty
# are the union of the five groups of ten).  Everything else derives from these groups.
_ADE_B10 = [
    [11, 16, 50, 64, 66, 73, 89, 92, 145, 146],
    [30, 37, 51, 52, 72, 85, 98, 114, 115, 138],
    [2, 35, 65, 97, 110, 111, 112, 118, 124, 141],
    [4, 7, 15, 41, 67, 78, 79, 88, 108, 139],
    [17, 20, 59, 68, 83, 94, 102, 122, 127, 137],
]
_ADE_C10 = [
    [3, 4, 7, 18, 39, 64, 73, 101, 113, 137],
    [47, 51, 55, 60, 62, 80, 116, 127, 140, 148],
    [22, 42, 49, 58, 59, 89, 91, 92, 108, 125],
    [2, 38, 53, 100, 104, 117, 130, 131, 141, 145],
    [15, 21, 72, 75, 88, 93, 103, 107, 122, 150],
]
_ADE_B50_0 = [0, 1, 9, 14, 18, 22, 24, 25, 27, 28, 29, 32, 38, 42, 45, 46, 47, 48, 49, 54, 56, 58,
              61, 62, 63, 69, 74, 75, 76, 77, 81, 82, 84, 90, 93, 96, 100, 103, 104, 109, 117, 119,
              121, 123, 128, 129, 130, 134, 135, 136, 144]
_ADE_C50_0 = [0, 5, 10, 11, 12, 13, 16, 17, 19, 20, 23, 27, 28, 30, 31, 32, 33, 37, 43, 46, 52, 56,
              57, 65, 66, 69, 70, 74, 76, 77, 79, 82, 83, 86, 87, 105, 109, 110, 111, 119, 128, 129,
              132, 133, 134, 138, 142, 143, 144, 146, 147]


def _ade_shuffled(groups10, first50):
    last50 = sorted(c for g in groups10 for c in g)
    mid50 = sorted(set(range(151)) - set(first50) - set(last50))
    first101 = sorted(first50 + mid50)
    return {
        "100-50": {0: first101, 1: last50},
        "100-10": {0: first101, **{i + 1: list(g) for i, g in enumerate(groups10)}},
        "50": {0: list(first50), 1: mid50, 2: last50},
    }


_ADE = {
    "offline": _contiguous(151, []),
    "100-50": _contiguous(101, [50]),
    "100-10": _contiguous(101, [10] * 5),
    "50": _contiguous(51, [50, 50]),
}
for _sfx, _g10, _f50 in (("b", _ADE_B10, _ADE_B50_0), ("c", _ADE_C10, _ADE_C50_0)):
    for _name, _split in _ade_shuffled(_g10, _f50).items():
        _ADE[_name + _sfx] = _split

_TABLES = {"voc": _VOC, "ade": _ADE, "city": _CITY}
# kept under the reference's names for callers that index the dicts directly
tasks_voc, tasks_ade, tasks_city = _VOC, _ADE, _CITY


def get_task_list():
    return [name for ds in ("voc", "ade", "city") for name in _TABLES[ds]]


def _split(dataset, name, step):
    if dataset not in _TABLES:
        raise NotImplementedError(dataset)
    split = _TABLES[dataset][name]
    assert step in split, f"You should provide a valid step! [{step} is out of range]"
    return split


def get_task_labels(dataset, name, step):
    """(new labels of ``step``, labels of all earlier steps, index directory)."""
    split = _split(dataset, name, step)
    old = [c for s in range(step) for c in split[s]]
    return list(split[step]), old, f"data/{dataset}/{name}"


def get_per_task_classes(dataset, name, step):
    """Number of classes introduced at each step 0..step (sizes of the classifier heads)."""
    split = _split(dataset, name, step)
    return [len(split[s]) for s in range(step + 1)]
