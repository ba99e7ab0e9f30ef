"""TEST INFRASTRUCTURE - CPU restatement of the label path of the reference's training data pipeline.

* incremental-step remapping: dataset/voc.py:143-203 (`order`, `inverted_order`, `tmp_labels`, `masking_value`, and
  the per-pixel lambda `inverted_order[x] if x in tmp_labels else masking_value`);
* RandomResizedCrop on the label: dataset/transform.py:481-553 -> torchvision.transforms.functional.resized_crop(lbl,
  i, j, h, w, size, NEAREST) = PIL crop((j, i, j+w, i+h)) + PIL resize(size, NEAREST); torchvision is not installed
  here and Pillow is a third-party dependency of the reference (requirements.txt pins Pillow==6.2.1; this container has
  12.2): the resize is executed BY PILLOW itself in the golden generator, and `nearest_indices` below restates its
  index rule (double accumulation), pinned against it;
* RandomHorizontalFlip: dataset/transform.py:300-318 (PIL FLIP_LEFT_RIGHT).
"""
import numpy as np


def target_lut(labels, labels_old, masking=True, data_masking="current"):
    """uint8 [256] table of the target transform (voc.py:143-203) for a step with new `labels` and `labels_old`."""
    labels = [0] + [l for l in labels if l != 0]
    labels_old = [0] + [l for l in labels_old if l != 0]
    order = [0] + labels_old[1:] + labels[1:]                       # voc.py:155 (`[0] + labels_old + labels`, 0 removed before)
    inverted_order = {label: order.index(label) for label in order}  # :181
    inverted_order[255] = 255                                        # :182
    masking_value = 0                                                # :180
    if not masking:
        raise AssertionError("the reference asserts False on this branch (voc.py:206)")
    if data_masking == "current":
        tmp_labels = labels + [255]                                  # :192
    elif data_masking == "current+old":
        tmp_labels = labels_old[1:] + labels + [255]                 # :194
    elif data_masking == "new":
        tmp_labels = labels                                          # :200
        masking_value = 255                                          # :201
    else:
        raise NotImplementedError(data_masking)
    return np.array([inverted_order[x] if x in tmp_labels else masking_value for x in range(256)], dtype=np.uint8)


def nearest_indices(extent, size):
    """Source index of every output position of Pillow's NEAREST resize from `extent` to `size` pixels."""
    a = extent / size
    xo = a * 0.5
    out = np.empty(size, dtype=np.int64)
    for x in range(size):
        out[x] = min(int(xo), extent - 1)
        xo += a
    return out


def label_path(lbl, box, size, flip, lut):
    """lbl uint8 [H0, W0]; box = (i, j, h, w); -> int64 [size, size] (the pipeline's label tensor after remapping)."""
    i, j, h, w = box
    ys, xs = nearest_indices(h, size), nearest_indices(w, size)
    out = lbl[i + ys][:, j + xs]
    if flip:
        out = out[:, ::-1]
    return lut[out].astype(np.int64)


def label_path_pil(lbl, box, size, flip, lut):
    """The same through Pillow (what torchvision's resized_crop / hflip do) - used by the golden generator."""
    from PIL import Image
    i, j, h, w = box
    im = Image.fromarray(lbl).crop((j, i, j + w, i + h)).resize((size, size), Image.NEAREST)
    if flip:
        im = im.transpose(Image.FLIP_LEFT_RIGHT)
    return lut[np.array(im)].astype(np.int64)
