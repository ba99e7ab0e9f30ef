"""Closed-form synthetic inputs (no RNG state, no libm): every value is a pure function of
(seed, flat index), built from 64-bit integer mixing, so the same tensors can be regenerated
bit-for-bit in this container (where the goldens are captured from the reference), on the GPU box
and inside ``bench.py``.  Shapes and value distributions follow SURVEY.md section 8(d).
"""
from __future__ import annotations

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(z: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def hash_u64(seed: int, n: int, stream: int = 0) -> np.ndarray:
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        key = _mix(np.uint64(seed & 0xFFFFFFFF) * np.uint64(0x1000193) + np.uint64(stream))
        return _mix(idx ^ key)


def uniform(seed: int, shape, stream: int = 0) -> np.ndarray:
    """float64 in [0, 1) with 24 random bits (exactly representable in float32)."""
    n = int(np.prod(shape))
    return ((hash_u64(seed, n, stream) >> np.uint64(40)).astype(np.float64) / float(1 << 24)).reshape(shape)


def normal(seed: int, shape, stream: int = 0) -> np.ndarray:
    """Approximately N(0, 1): centred sum of four 16-bit uniforms (Irwin-Hall, variance 4/12) scaled
    by sqrt(3).  Only exact integer and IEEE add/mul operations, so platform independent."""
    n = int(np.prod(shape))
    h = hash_u64(seed, n, stream)
    s = np.zeros(n, dtype=np.float64)
    for k in range(4):
        s += ((h >> np.uint64(16 * k)) & np.uint64(0xFFFF)).astype(np.float64)
    s = (s / 65536.0 - 2.0) * 1.7320508075688772
    return s.reshape(shape)


def randint(seed: int, shape, lo: int, hi: int, stream: int = 0) -> np.ndarray:
    """integers in [lo, hi)."""
    n = int(np.prod(shape))
    return (lo + (hash_u64(seed, n, stream) >> np.uint64(33)) % np.uint64(hi - lo)).astype(np.int64).reshape(shape)


def t_normal(seed, shape, stream=0, scale=1.0, dtype=torch.float32) -> torch.Tensor:
    return torch.from_numpy((normal(seed, shape, stream) * scale).astype(np.float32)).to(dtype)


def seg_labels(seed: int, B: int, H: int, W: int, new_ids, rects: int = 3, border: int = 4,
               cover: float = 0.25) -> torch.Tensor:
    """int64 [B, H, W] label maps as the incremental datasets deliver them at step >= 1: background
    0 (old classes are already remapped to 0, reference dataset/voc.py:182-208), ``rects`` axis-aligned
    rectangles of new-class ids per image covering about ``cover`` of the pixels, and a ``border``-pixel
    frame of the ignore label 255."""
    new_ids = list(new_ids)
    lab = np.zeros((B, H, W), dtype=np.int64)
    side = max(2, int(round((cover * H * W / rects) ** 0.5)))
    r = randint(seed, (B, rects, 3), 0, 1 << 20, stream=11)
    for b in range(B):
        for k in range(rects):
            hh = min(H, max(2, side + int(r[b, k, 2] % max(1, side // 2)) - side // 4))
            ww = min(W, max(2, (side * side) // hh))
            y0 = int(r[b, k, 0] % max(1, H - hh + 1))
            x0 = int(r[b, k, 1] % max(1, W - ww + 1))
            lab[b, y0:y0 + hh, x0:x0 + ww] = new_ids[int(r[b, k, 0] >> 7) % len(new_ids)]
    if border > 0:
        lab[:, :border] = 255
        lab[:, -border:] = 255
        lab[:, :, :border] = 255
        lab[:, :, -border:] = 255
    return torch.from_numpy(lab)


def contrastive_case(seed: int, B: int, N: int, h: int, w: int, K: int, H: int, W: int, new_ids,
                     logit_scale: float = 2.0):
    """Inputs of ``pre_contractive_pixel``: student/teacher pre-logit maps [B, N, h, w], teacher
    low-resolution logits [B, K, h, w] ~ N(0, logit_scale^2) and full-resolution labels [B, H, W]."""
    f_n = t_normal(seed, (B, N, h, w), stream=1)
    f_o = t_normal(seed, (B, N, h, w), stream=2)
    l_po = t_normal(seed, (B, K, h, w), stream=3, scale=logit_scale)
    labels = seg_labels(seed, B, H, W, new_ids)
    return f_n, f_o, l_po, labels


def images(seed: int, B: int, S: int) -> torch.Tensor:
    """[B, 3, S, S] ~ N(0, 1): statistics of a normalised crop (reference run.py:53-54)."""
    return t_normal(seed, (B, 3, S, S), stream=5)


CAL_BN3 = 0.1       # see fill_state_dict(calibrated=True)
CAL_CLS = 1.0


def fill_state_dict(state: dict, seed: int = 42, calibrated: bool = False) -> dict:
    """Deterministic stand-in for a trained checkpoint (there is no network for the real
    ``pretrained/resnet101_iabn_sync.pth.tar``): He-scaled conv weights, norm scales near 1 (positive,
    as in the pretrained ABN files), small biases / running means, running variances near 1.  Values
    depend only on (seed, key name, shape) - not on which other keys are present.

    ``calibrated``: the same values with the scale of every block's LAST norm (``convs.bn3.weight``: the one in front of the
    residual sum) multiplied by ``CAL_BN3`` and the classifier rows by ``CAL_CLS``.  With unit scales everywhere the residual stream of the frozen,
    evaluation-mode teacher doubles its variance in each of the 33 blocks - logits of 1e5 that turn every rounding difference
    into percent on the losses - whereas a trained ``iabn_sync`` checkpoint keeps evaluation activations O(1).
    The calibrated fill has teacher logits of order 10 and is what the bf16 (--opt_level O1) tests are held against."""
    import zlib
    out = {}
    for k in sorted(state):
        i = zlib.crc32(k.encode()) & 0x3FFFFFFF
        v = state[k]
        if not torch.is_floating_point(v):
            out[k] = v.clone()
            continue
        shp = tuple(v.shape)
        z = normal(seed, shp if shp else (1,), stream=1000 + i).reshape(shp)
        if k.endswith("running_var"):
            z = 1.0 + 0.1 * np.abs(z)
        elif k.endswith("running_mean"):
            z = 0.1 * z
        elif v.dim() == 4:                       # conv weight [out, in, kh, kw]
            z = z * (2.0 / (shp[1] * shp[2] * shp[3])) ** 0.5
            if calibrated and ".cls." in "." + k:
                z = z * CAL_CLS
        elif k.endswith("weight"):               # norm scale
            z = 1.0 + 0.1 * z
            if calibrated and k.endswith("convs.bn3.weight"):
                z = z * CAL_BN3
        else:                                    # biases
            z = 0.1 * z
        out[k] = torch.from_numpy(np.asarray(z, dtype=np.float32)).to(v.dtype)
    return out
