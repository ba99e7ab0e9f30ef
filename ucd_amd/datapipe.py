"""Label path of the training data pipeline on the device (first piece).

Host-side mirror of what the reference does per image on the CPU: the target transform of
``dataset/voc.py:143-203`` (``VOCSegmentationIncremental``: re-ordering of the label ids for the incremental step and
masking of the classes that are not visible, executed there as a Python lambda per pixel) and, for the label map,
``RandomResizedCrop`` + ``RandomHorizontalFlip`` of ``dataset/transform.py:300-318,481-553``.  The random parameters are
drawn exactly like the reference draws them (``RandomResizedCrop.get_params``, ``random.random() < 0.5``); the pixel work
of the whole batch is one ``ucd_label_path`` call on label maps that stay resident in HBM.

``DeviceImagePath`` is the image half: crop + Pillow's BILINEAR resize (its 8-bit separable resampler with 22-bit
fixed-point coefficients, bit-exact) + flip + ``ToTensor`` + ``Normalize`` (``transform.py:37-86``, ``run.py:49-55``) for
decoded uint8 RGB images resident in HBM; JPEG decoding itself stays on the host.
"""
from __future__ import annotations

import math
import random

import torch

from . import hip


def target_lut(labels, labels_old, masking=True, data_masking="current"):
    """uint8 [256] table: stored label id -> id seen by the step (``voc.py:143-203``)."""
    labels = [0] + [l for l in labels if l != 0]
    labels_old = [0] + [l for l in labels_old if l != 0]
    order = [0] + labels_old[1:] + labels[1:]
    inverted_order = {label: order.index(label) for label in order}
    inverted_order[255] = 255
    masking_value = 0                       # future classes are background
    if not masking:
        raise AssertionError("masking=False is not a path of the reference (voc.py:206 asserts)")
    if data_masking == "current":
        tmp_labels = labels + [255]
    elif data_masking == "current+old":
        tmp_labels = labels_old[1:] + labels + [255]
    elif data_masking == "new":
        tmp_labels = labels
        masking_value = 255
    else:
        raise NotImplementedError(f"data_masking={data_masking} not yet implemented sorry not sorry.")   # voc.py:196-198
    return torch.tensor([inverted_order[x] if x in tmp_labels else masking_value for x in range(256)], dtype=torch.uint8)


def ade_target_lut(labels, labels_old, masking=True, ignore_test_bg=False):
    """uint8 [256] table of ``AdeSegmentationIncremental`` (dataset/ade.py:103-147): order ``[0] + labels_old + labels`` with the
    zeros stripped from both lists, 255 kept by the re-ordering table; with ``masking`` only the step's own labels keep their
    id - the background 0 and the ignore label 255 included go to the masking value (ade.py:139-141 tests ``x in self.labels``,
    which holds neither), 0 or 255 with ``ignore_test_bg``."""
    labels = [l for l in labels if l != 0]
    labels_old = [l for l in labels_old if l != 0]
    order = [0] + labels_old + labels
    inverted = {label: order.index(label) for label in order}
    masking_value = 255 if ignore_test_bg else 0
    if ignore_test_bg:
        inverted[0] = masking_value
    inverted[255] = 255
    keep = labels if masking else list(inverted)
    return torch.tensor([inverted[x] if x in keep else masking_value for x in range(256)], dtype=torch.uint8)


# raw Cityscapes labelIds -> the 19 training classes + void 0 (dataset/cityscape.py:49-66: np.digitize against -1 .. 33, then
# the `_key` table): entry r is the class of raw id r; ids outside 0 .. 33 do not occur (the reference asserts)
CITY_KEY = (0, 0, 0, 0, 0, 0, 0, 1, 2, 0, 0, 3, 4, 5, 0, 0, 0, 6, 0, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 0, 0, 17, 18, 19)


def city_class_lut():
    """uint8 [256]: raw labelId -> class index 0 .. 19 (``CitySegmentation._class_to_index``); the reference's table starts at
    raw id -1 (``_mapping = range(-1, 34)``, digitize with right=True), so raw id r reads ``_key[r + 1]``... which is what
    CITY_KEY[r] holds for r >= 0 after dropping the -1 entry."""
    lut = torch.zeros(256, dtype=torch.uint8)
    lut[:len(CITY_KEY)] = torch.tensor(CITY_KEY, dtype=torch.uint8)
    return lut


def city_target_lut(labels, labels_old, masking=True, train=True):
    """uint8 [256] table of ``CitySegmentationIncremental`` over CLASS indices (dataset/cityscape.py:123-152): labels get a
    leading 0, order ``[0] + labels_old + labels``, the masking value is 0 in training and 255 otherwise and 255 itself maps to
    it; with ``masking`` only ``[0] + labels + [255]`` keep their (re-ordered) id."""
    labels = [l for l in labels if l != 0]
    labels_old = [l for l in labels_old if l != 0]
    order = [0] + labels_old + labels
    masking_value = 0 if train else 255
    inverted = {label: order.index(label) for label in order}
    inverted[255] = masking_value
    keep = ([0] + labels + [255]) if masking else list(inverted)
    return torch.tensor([inverted[x] if x in keep else masking_value for x in range(256)], dtype=torch.uint8)


def random_resized_crop_params(height, width, scale=(0.5, 2.0), ratio=(3. / 4., 4. / 3.)):
    """``RandomResizedCrop.get_params`` (transform.py:505-540): (i, j, h, w), same draws from ``random`` in the same order."""
    area = width * height
    for _ in range(10):
        target_area = random.uniform(*scale) * area
        log_ratio = (math.log(ratio[0]), math.log(ratio[1]))
        aspect_ratio = math.exp(random.uniform(*log_ratio))
        w = int(round(math.sqrt(target_area * aspect_ratio)))
        h = int(round(math.sqrt(target_area / aspect_ratio)))
        if w <= width and h <= height:
            i = random.randint(0, height - h)
            j = random.randint(0, width - w)
            return i, j, h, w
    in_ratio = width / height
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


class DeviceLabelPath:
    """labels of a batch: ``__call__(label_maps, boxes, flips) -> int64 [B, S, S]`` on the device.
    ``label_maps``: list of uint8 device tensors [H0_b, W0_b]; ``boxes``: list of (i, j, h, w); ``flips``: list of bool."""

    def __init__(self, size, lut):
        self.size = int(size)
        self.lut = lut

    def __call__(self, label_maps, boxes, flips):
        B, S = len(label_maps), self.size
        dev = label_maps[0].device
        if dev.type != "cuda":
            raise RuntimeError("ucd_amd.datapipe runs on the GPU only (there is no CPU fallback)")
        maps = [m if (m.dtype == torch.uint8 and m.is_contiguous()) else m.to(torch.uint8).contiguous() for m in label_maps]
        lut = self.lut.to(dev)
        desc = torch.tensor([[m.shape[0], m.shape[1], b[0], b[1], b[2], b[3], int(bool(f)), 0]
                             for m, b, f in zip(maps, boxes, flips)], dtype=torch.int32).to(dev, non_blocking=True)
        for m, (i, j, h, w) in zip(maps, boxes):
            if not (0 <= i and 0 <= j and h > 0 and w > 0 and i + h <= m.shape[0] and j + w <= m.shape[1]):
                raise ValueError(f"crop box {(i, j, h, w)} outside a {tuple(m.shape)} label map")
        ptrs = torch.tensor([m.data_ptr() for m in maps], dtype=torch.int64).to(dev, non_blocking=True)
        tables = torch.empty(B * 2 * S, dtype=torch.int32, device=dev)
        out = torch.empty(B, S, S, dtype=torch.int64, device=dev)
        hip._check(hip.load().ucd_label_path(hip.ptr(ptrs), hip.ptr(desc), B, S, hip.ptr(lut), hip.ptr(tables), hip.ptr(out),
                                             hip.stream()), "ucd_label_path")
        return out


class DeviceImagePath:
    """images of a batch: ``__call__(images, boxes, flips) -> float32 [B, 3, S, S]`` (channels-last storage) on the
    device.  ``images``: list of uint8 device tensors [H0_b, W0_b, 3] (decoded RGB)."""

    def __init__(self, size, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
        self.size, self.mean, self.std = int(size), tuple(float(v) for v in mean), tuple(float(v) for v in std)

    def __call__(self, images, boxes, flips):
        B, S = len(images), self.size
        dev = images[0].device
        if dev.type != "cuda":
            raise RuntimeError("ucd_amd.datapipe runs on the GPU only (there is no CPU fallback)")
        imgs = [m if (m.dtype == torch.uint8 and m.is_contiguous()) else m.to(torch.uint8).contiguous() for m in images]
        for m, (i, j, h, w) in zip(imgs, boxes):
            if m.dim() != 3 or m.shape[2] != 3:
                raise ValueError("images must be [H, W, 3] uint8 RGB")
            if not (0 <= i and 0 <= j and h > 0 and w > 0 and i + h <= m.shape[0] and j + w <= m.shape[1]):
                raise ValueError(f"crop box {(i, j, h, w)} outside a {tuple(m.shape)} image")
        hmax = max(b[2] for b in boxes)
        extent = max(max(b[2], b[3]) for b in boxes)
        kmax = int(math.ceil(max(1.0, extent / S))) * 2 + 1
        desc = torch.tensor([[m.shape[0], m.shape[1], b[0], b[1], b[2], b[3], int(bool(f)), 0]
                             for m, b, f in zip(imgs, boxes, flips)], dtype=torch.int32).to(dev, non_blocking=True)
        ptrs = torch.tensor([m.data_ptr() for m in imgs], dtype=torch.int64).to(dev, non_blocking=True)
        coeff = torch.empty(B * 2 * S * (kmax + 2), dtype=torch.int32, device=dev)
        tmp = torch.empty(B * hmax * S * 3, dtype=torch.uint8, device=dev)
        out = torch.empty(B, S, S, 3, dtype=torch.float32, device=dev)
        hip._check(hip.load().ucd_image_path(hip.ptr(ptrs), hip.ptr(desc), B, S, kmax, hmax, *self.mean, *self.std,
                                             hip.ptr(coeff), hip.ptr(tmp), hip.ptr(out), hip.stream()), "ucd_image_path")
        return out.permute(0, 3, 1, 2)            # [B, 3, S, S] with channels-last strides
