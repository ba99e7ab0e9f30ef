"""TEST INFRASTRUCTURE - restatement of the reference's incremental-dataset bookkeeping on plain arrays.

``filter_images`` (dataset/utils.py:19-42): an image is kept when it holds at least one of the step's labels and - in the
disjoint setting - no label outside ``labels + labels_old + {0, 255}``;  ``target_transform`` (dataset/voc.py:176-203): the
per-pixel lambda that maps stored labels to the ids the step sees (inverted order of ``[0] + labels_old + labels``, 255 kept,
everything else -> the masking value).  PARITY: ``filter_images`` is pinned by index lists the reference's own function
produced on the label maps of tests/test_dataset.py (tests/golden/make_dataset_golden.py, dataset_filter.npz: five task /
step pairs, overlap and disjoint); the label table is pinned by the goldens of tests/golden/make_datapipe_golden.py (the
reference's lambda executed on its own task tables).  The class ``VOCSegmentationIncremental`` itself needs torchvision and
real VOC files and is followed by reading the cited lines."""
import numpy as np


def filter_images(label_maps, labels, labels_old=None, overlap=True):
    labels = [l for l in labels if l != 0]                                   # :23-24
    labels_cum = labels + list(labels_old or []) + [0, 255]                  # :29
    idxs = []
    for i, lab in enumerate(label_maps):
        cls = np.unique(np.asarray(lab))                                     # :37
        keep = any(x in labels for x in cls)                                 # :32
        if not overlap:
            keep = keep and all(x in labels_cum for x in cls)                # :34
        if keep:
            idxs.append(i)
    return idxs


def target_transform(label, labels, labels_old, data_masking="current"):
    labels = [0] + [l for l in labels if l != 0]
    labels_old = [0] + [l for l in labels_old if l != 0]
    order = [0] + labels_old[1:] + labels[1:]                                # voc.py:160
    inverted = {l: order.index(l) for l in order}
    inverted[255] = 255
    masking_value = 0
    if data_masking == "current":
        tmp = labels + [255]
    elif data_masking == "current+old":
        tmp = labels_old + labels + [255]
    elif data_masking == "new":
        tmp, masking_value = labels, 255
    else:
        raise NotImplementedError(data_masking)
    out = np.full_like(np.asarray(label), masking_value)
    for v in np.unique(label):
        if v in tmp:
            out[np.asarray(label) == v] = inverted[v]
    return out
