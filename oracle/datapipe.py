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


# ---- image half: crop + Pillow BILINEAR resize + flip + ToTensor + Normalize -------------------------------------------
# Pillow's ImagingResample for 8-bit images (src/libImaging/Resample.c): separable, horizontal pass first into an 8-bit
# intermediate, coefficients in 22-bit fixed point (PRECISION_BITS = 32 - 8 - 2), support widened by the down-scaling
# factor (anti-aliasing).  Restated here and pinned bit-for-bit against Pillow by the golden generator.
PRECISION_BITS = 22


def bilinear_coeffs(in_size, out_size):
    """(bounds [out, 2] = (xmin, count), kk [out, ksize] int64) of precompute_coeffs + normalize_coeffs_8bpc."""
    import math
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale                                    # BILINEAR: support 1.0
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int64)
    kk = np.zeros((out_size, ksize), dtype=np.int64)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.zeros(ksize)
        ww = 0.0
        for x in range(xmax):
            v = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - v if v < 1.0 else 0.0
            ww += w[x]
        if ww != 0.0:
            w[:xmax] /= ww
        for x in range(ksize):
            kk[xx, x] = int(-0.5 + w[x] * (1 << PRECISION_BITS)) if w[x] < 0 else int(0.5 + w[x] * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _resample_rows(img, out_size):
    """Resample axis 1 of uint8 [H, W, C] to out_size."""
    b, kk = bilinear_coeffs(img.shape[1], out_size)
    out = np.empty((img.shape[0], out_size, img.shape[2]), dtype=np.uint8)
    for xx in range(out_size):
        xmin, cnt = b[xx]
        acc = (1 << (PRECISION_BITS - 1)) + (img[:, xmin:xmin + cnt, :].astype(np.int64) * kk[xx, :cnt][None, :, None]).sum(1)
        out[:, xx, :] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return out


def resize_bilinear(img, size):
    """uint8 [H, W, C] -> uint8 [size, size, C], Pillow's Image.resize((size, size), BILINEAR)."""
    t = img
    if img.shape[1] != size:
        t = _resample_rows(t, size)
    if img.shape[0] != size:
        t = _resample_rows(t.transpose(1, 0, 2), size).transpose(1, 0, 2)
    return t


MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)     # run.py:53-54
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def image_path(img, box, size, flip):
    """uint8 RGB [H0, W0, 3] -> float32 [3, size, size]: RandomResizedCrop + RandomHorizontalFlip + ToTensor + Normalize
    (dataset/transform.py:481-553, 300-318, 37-58, 61-86; run.py:49-55)."""
    i, j, h, w = box
    t = resize_bilinear(np.ascontiguousarray(img[i:i + h, j:j + w]), size)
    if flip:
        t = t[:, ::-1]
    x = t.transpose(2, 0, 1).astype(np.float32) / np.float32(255)         # ToTensor: float().div(255)
    return (x - MEAN[:, None, None]) / STD[:, None, None]                  # Normalize: sub_(mean).div_(std)


def image_path_pil(img, box, size, flip):
    """The same through Pillow + torch (what torchvision's resized_crop / hflip / to_tensor / normalize do)."""
    import torch
    from PIL import Image
    i, j, h, w = box
    im = Image.fromarray(img).crop((j, i, j + w, i + h)).resize((size, size), Image.BILINEAR)
    if flip:
        im = im.transpose(Image.FLIP_LEFT_RIGHT)
    t = torch.from_numpy(np.array(im)).permute(2, 0, 1).contiguous().float().div(255)
    mean = torch.as_tensor([0.485, 0.456, 0.406], dtype=torch.float32)
    std = torch.as_tensor([0.229, 0.224, 0.225], dtype=torch.float32)
    t.sub_(mean[:, None, None]).div_(std[:, None, None])
    return t.numpy()
