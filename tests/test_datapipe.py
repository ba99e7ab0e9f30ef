"""Label path of the data pipeline (first piece): integer work, bit-exact.
CPU: the oracle's index rule against golden outputs produced by Pillow itself; the step-remapping table against the
literal lambda of the reference executed on its own task tables.  GPU: ucd_label_path against the same goldens."""
import random
import zlib

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import datapipe as OD
from ucd_amd import tasks
from ucd_amd.datapipe import DeviceImagePath, DeviceLabelPath, random_resized_crop_params, target_lut

LUTS = {"voc_15-5_s1": ("voc", "15-5", 1, "current"), "voc_15-5s_s3": ("voc", "15-5s", 3, "current"),
        "voc_19-1_s1_old": ("voc", "19-1", 1, "current+old"), "voc_10-10_s1_new": ("voc", "10-10", 1, "new")}


def _label_map(rng, H0, W0):
    return rng.choice([0, 3, 7, 15, 16, 18, 20, 255], size=(H0 // 8 + 1, W0 // 8 + 1)).astype(np.uint8).repeat(8, 0).repeat(8, 1)[:H0, :W0]


def _cases(g):
    rng = np.random.RandomState(2024)
    for k, (H0, W0, i, j, h, w, flip, S) in enumerate(g["cases"].tolist()):
        a, b = int(rng.randint(120, 501)), int(rng.randint(120, 501))
        assert (a, b) == (H0, W0)
        lbl = _label_map(rng, H0, W0)
        rng.randint(40, H0 + 1); rng.randint(40, W0 + 1); rng.randint(0, H0 - h + 1); rng.randint(0, W0 - w + 1)   # replay the draws
        yield k, lbl, (i, j, h, w), bool(flip), S


@pytest.mark.parametrize("name", list(LUTS))
def test_step_remapping_table_matches_reference_lambda(name):
    g = load_golden("datapipe.npz")
    dataset, task, step, dm = LUTS[name]
    labels, labels_old, _ = tasks.get_task_labels(dataset, task, step)
    assert np.array_equal(OD.target_lut(labels, labels_old, True, dm), g[f"lut::{name}"])
    assert np.array_equal(target_lut(labels, labels_old, True, dm).numpy(), g[f"lut::{name}"])


def test_oracle_label_path_matches_pillow_golden():
    g = load_golden("datapipe.npz")
    lut = g["lut::voc_15-5_s1"]
    for k, lbl, box, flip, S in _cases(g):
        res = OD.label_path(lbl, box, S, flip, lut)
        assert zlib.crc32(res.tobytes()) == int(g[f"case{k}::crc"][0]), k
        if S <= 64:
            assert np.array_equal(res.astype(np.uint8), g[f"case{k}::out"])


def test_crop_parameters_follow_the_reference_draws():
    """random_resized_crop_params against (i, j, h, w) produced by the reference's own RandomResizedCrop.get_params
    (transform.py:505-540, imported by the golden generator) from the same `random` seeds, incl. the fall-back branch."""
    g = load_golden("datapipe.npz")
    for seed, H0, W0, i, j, h, w in g["crop_params"].tolist():
        random.seed(seed)
        assert random_resized_crop_params(H0, W0, (0.5, 2.0), (3. / 4., 4. / 3.)) == (i, j, h, w), seed
        assert 0 <= i and 0 <= j and i + h <= H0 and j + w <= W0


@pytest.mark.gpu
def test_device_label_path_bit_exact():
    g = load_golden("datapipe.npz")
    dev = torch.device("cuda:0")
    lut = torch.from_numpy(g["lut::voc_15-5_s1"])
    by_size = {}
    for k, lbl, box, flip, S in _cases(g):
        by_size.setdefault(S, []).append((k, lbl, box, flip))
    for S, items in by_size.items():                         # one batched call per output size, ragged source sizes
        path = DeviceLabelPath(S, lut)
        out = path([torch.from_numpy(l).to(dev) for _, l, _, _ in items], [b for _, _, b, _ in items], [f for _, _, _, f in items])
        assert out.dtype == torch.int64 and out.shape == (len(items), S, S)
        for n, (k, lbl, box, flip) in enumerate(items):
            res = out[n].cpu().numpy()
            assert zlib.crc32(res.tobytes()) == int(g[f"case{k}::crc"][0]), (k, S)
            assert np.array_equal(res, OD.label_path(lbl, box, S, flip, g["lut::voc_15-5_s1"]))


def _image(rng, H0, W0):
    return (rng.randint(0, 256, size=(H0 // 4 + 1, W0 // 4 + 1, 3)).repeat(4, 0).repeat(4, 1)[:H0, :W0]
            + rng.randint(-9, 10, size=(H0, W0, 3))).clip(0, 255).astype(np.uint8)


def _icases(g):
    rng = np.random.RandomState(4048)
    for k, (H0, W0, i, j, h, w, flip, S) in enumerate(g["icases"].tolist()):
        a, b = int(rng.randint(100, 501)), int(rng.randint(100, 501))
        assert (a, b) == (H0, W0)
        img = _image(rng, H0, W0)
        rng.randint(30, H0 + 1); rng.randint(30, W0 + 1); rng.randint(0, H0 - h + 1); rng.randint(0, W0 - w + 1)
        yield k, img, (i, j, h, w), bool(flip), S


def test_oracle_image_path_matches_pillow_torch_golden():
    """Pillow's 8-bit BILINEAR resampler (fixed-point coefficients, anti-aliased down-scaling) + ToTensor + Normalize,
    restated in numpy: bit-identical float32 output to Pillow + torch on ten crops (up- and down-scaling, flips)."""
    g = load_golden("datapipe.npz")
    for k, img, box, flip, S in _icases(g):
        res = np.ascontiguousarray(OD.image_path(img, box, S, flip))
        assert zlib.crc32(res.tobytes()) == int(g[f"img{k}::crc"][0]), k
        if S <= 48:
            assert np.array_equal(res, g[f"img{k}::out"])


@pytest.mark.gpu
def test_device_image_path_bit_exact():
    g = load_golden("datapipe.npz")
    dev = torch.device("cuda:0")
    by_size = {}
    for k, img, box, flip, S in _icases(g):
        by_size.setdefault(S, []).append((k, img, box, flip))
    for S, items in by_size.items():
        out = DeviceImagePath(S)([torch.from_numpy(im).to(dev) for _, im, _, _ in items], [b for _, _, b, _ in items],
                                 [f for _, _, _, f in items])
        assert out.shape == (len(items), 3, S, S) and out.is_contiguous(memory_format=torch.channels_last)
        for n, (k, img, box, flip) in enumerate(items):
            res = np.ascontiguousarray(out[n].cpu().numpy())
            assert zlib.crc32(res.tobytes()) == int(g[f"img{k}::crc"][0]), (k, S)
