"""Synthetic ADE20K / Cityscapes directory trees (tiny images, block-constant label maps) shared by tests/test_dataset.py and
tests/golden/make_dataset_golden.py - the golden index lists and label tables were captured from the reference's dataset classes
on exactly these trees."""
import os

import numpy as np

ADE_SETS = [[0, 1, 5, 120], [0, 101], [0, 3, 150, 255], [0, 99, 100], [0, 2, 140], [0, 15, 19], [0, 118, 16, 4], [0, 6, 255], [0, 107], [0, 150, 1]]
CITY_RAW_SETS = [[7, 8, 11], [26, 7], [0, 24, 33], [27, 28, 21], [7, 23], [31, 32, 11], [26, 27, 28, 8], [6, 12, 4], [33, 21], [25, 7, 1]]


def make_ade_tree(root, n=10, seed=11):
    from PIL import Image
    rng = np.random.RandomState(seed)
    for split in ("training", "validation"):
        os.makedirs(os.path.join(root, "ADEChallengeData2016", "images", split))
        os.makedirs(os.path.join(root, "ADEChallengeData2016", "annotations", split))
    maps = []
    for k in range(n):
        H, W = int(rng.randint(90, 180)), int(rng.randint(90, 180))
        img = rng.randint(0, 256, size=(H // 6 + 1, W // 6 + 1, 3)).astype(np.uint8).repeat(6, 0).repeat(6, 1)[:H, :W]
        lab = rng.choice(ADE_SETS[k % len(ADE_SETS)], size=(H // 10 + 1, W // 10 + 1)).astype(np.uint8).repeat(10, 0).repeat(10, 1)[:H, :W]
        for split in ("training", "validation"):
            if split == "validation" and k >= n // 2:
                continue
            Image.fromarray(img).save(os.path.join(root, "ADEChallengeData2016", "images", split, f"ADE_{k:08d}.jpg"), quality=95)
            Image.fromarray(lab).save(os.path.join(root, "ADEChallengeData2016", "annotations", split, f"ADE_{k:08d}.png"))
        maps.append(lab)
    return maps


def make_city_tree(root, n=10, seed=13):
    from PIL import Image
    rng = np.random.RandomState(seed)
    names = []
    for k in range(n):
        city = ["aachen", "bochum", "ulm"][k % 3]
        for split in ("train", "val"):
            os.makedirs(os.path.join(root, "Cityscapes", "leftImg8bit", split, city), exist_ok=True)
            os.makedirs(os.path.join(root, "Cityscapes", "gtFine", split, city), exist_ok=True)
        H, W = int(rng.randint(90, 150)), int(rng.randint(120, 200))
        img = rng.randint(0, 256, size=(H // 6 + 1, W // 6 + 1, 3)).astype(np.uint8).repeat(6, 0).repeat(6, 1)[:H, :W]
        raw = rng.choice(CITY_RAW_SETS[k % len(CITY_RAW_SETS)], size=(H // 10 + 1, W // 10 + 1)).astype(np.uint8).repeat(10, 0).repeat(10, 1)[:H, :W]
        name = f"{city}_{k:06d}_000019"
        for split in ("train", "val"):
            if split == "val" and k >= n // 2:
                continue
            Image.fromarray(img).save(os.path.join(root, "Cityscapes", "leftImg8bit", split, city, name + "_leftImg8bit.png"))
            Image.fromarray(raw).save(os.path.join(root, "Cityscapes", "gtFine", split, city, name + "_gtFine_labelIds.png"))
        names.append(name)
    return names


