"""Pascal-VOC incremental-segmentation dataset on top of the device data pipeline.

Host-side mirror of the reference's ``dataset/voc.py`` (``VOCSegmentation`` :38-119, ``VOCSegmentationIncremental`` :122-237),
``dataset/utils.py`` (``filter_images`` :19-42, ``Subset`` :45-87) and of the transform stack ``run.py:49-73`` builds - same
constructor arguments, same file layout (``splits/train_aug.txt`` / ``val.txt`` lines ``/JPEGImages/x.jpg
/SegmentationClassAug/x.png``), same index files (``data/voc/<task>[-ov]/train-<step>.npy``), same label masking /
re-ordering - split where the hardware wants it:

* the host decodes (Pillow) and hands out the RAW uint8 image / label pair of a sample;
* ``DeviceBatcher`` (the ``collate_fn``) draws the reference's random parameters on the host (``RandomResizedCrop.get_params``,
  the flip coin: same ``random`` draws in the same order as ``dataset/transform.py``) and runs crop + Pillow-exact resize + flip
  + ToTensor + Normalize (images) and crop + NEAREST resize + flip + the step's label re-mapping (labels) on the GPU
  (``ucd_amd.datapipe``: bit-exact against Pillow / the reference's per-pixel lambda).  The reference does all of it per sample
  on the host with ``num_workers=0`` (37 img/s measured against 142 846 on the device, tests/diag/datapipe_bench.py).

Validation with ``--crop_val`` (``Resize`` + ``CenterCrop``, run.py:58-65) resizes on the host with Pillow (exact by
construction) and normalises on the device.  ADE20K (dataset/ade.py) and Cityscapes (dataset/cityscape.py) follow the same
split: their listings, filters and label tables are mirrored here (the Cityscapes raw-id -> class table is composed with the
step's table into ONE device gather).  ``DeviceLoader`` puts the decode into DataLoader worker processes.
"""
from __future__ import annotations

import os
import random

import numpy as np
import torch
import torch.utils.data as data

from . import datapipe


def _open_rgb(path):
    from PIL import Image
    return np.array(Image.open(path).convert("RGB"), dtype=np.uint8)


def _open_label(path):
    from PIL import Image
    return np.array(Image.open(path), dtype=np.uint8)


class VOCSegmentation(data.Dataset):
    """File listing of the reference's ``VOCSegmentation`` (dataset/voc.py:49-93); items are decoded uint8 arrays
    ``(image [H, W, 3], label [H, W])`` - no transform here."""

    def __init__(self, root, image_set="train", is_aug=True, transform=None):
        self.root = os.path.expanduser(root)
        self.image_set = image_set
        splits_dir = os.path.join(self.root, "splits")
        if not os.path.isdir(self.root):
            raise RuntimeError(f"Dataset not found or corrupted. at location = {self.root}")
        if is_aug and image_set == "train":
            mask_dir = os.path.join(self.root, "SegmentationClassAug")
            assert os.path.exists(mask_dir), "SegmentationClassAug not found"
            split_f = os.path.join(splits_dir, "train_aug.txt")
        else:
            split_f = os.path.join(splits_dir, image_set.rstrip("\n") + ".txt")
        if not os.path.exists(split_f):
            raise ValueError(f'Wrong image_set entered! Please use image_set="train" or image_set="trainval" or image_set="val" {split_f}')
        with open(split_f, "r") as f:
            file_names = [x[:-1].split(" ") for x in f.readlines()]
        self.images = [(os.path.join(self.root, x[0][1:]), os.path.join(self.root, x[1][1:])) for x in file_names]

    def __getitem__(self, index):
        return _open_rgb(self.images[index][0]), _open_label(self.images[index][1])

    def label(self, index):
        return _open_label(self.images[index][1])

    def __len__(self):
        return len(self.images)


def filter_images(dataset, labels, labels_old=None, overlap=True):
    """Indices of the images that carry at least one of ``labels`` (overlap) and - disjoint setting - nothing outside
    ``labels + labels_old + {0, 255}`` (dataset/utils.py:19-42, on the labels as stored)."""
    labels = [l for l in labels if l != 0]
    labels_cum = labels + list(labels_old or []) + [0, 255]
    idxs = []
    for i in range(len(dataset)):
        cls = np.unique(dataset.label(i) if hasattr(dataset, "label") else np.array(dataset[i][1]))
        if any(x in labels for x in cls) and (overlap or all(x in labels_cum for x in cls)):
            idxs.append(i)
    return idxs


def _save_index_file(path, idxs):
    """Write an index file atomically: the other ranks of a first multi-rank run test ``os.path.exists`` and ``np.load`` the same
    path while rank 0 writes it (dataset/voc.py:160-166 has that race); with a temporary file + ``os.replace`` they see either
    no file (and filter themselves: same list) or the complete one."""
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    tmp = f"{path}.tmp{os.getpid()}.npy"
    np.save(tmp, np.array(idxs, dtype=int))
    os.replace(tmp, path)


class VOCSegmentationIncremental(data.Dataset):
    """``VOCSegmentationIncremental(root, train, transform, labels, labels_old, idxs_path, masking, overlap, data_masking,
    test_on_val)`` of the reference (dataset/voc.py:124-216).  ``transform`` is accepted and ignored: the transform of a batch
    runs on the device (``DeviceBatcher``); ``self.lut`` is the step's label table (inverted order + masking)."""

    def __init__(self, root, train=True, transform=None, labels=None, labels_old=None, idxs_path=None, masking=True,
                 overlap=True, data_masking="current", test_on_val=False, **kwargs):
        full_voc = VOCSegmentation(root, "train" if train else "val", is_aug=True, transform=None)
        self.full, self.train = full_voc, train
        self.labels, self.labels_old = [], []
        if labels is not None:
            labels_old = labels_old if labels_old is not None else []
            labels = [l for l in labels if l != 0]
            labels_old = [l for l in labels_old if l != 0]
            assert not any(l in labels_old for l in labels), "labels and labels_old must be disjoint sets"
            self.labels, self.labels_old = [0] + labels, [0] + labels_old
            self.order = [0] + labels_old + labels
            if idxs_path is not None and os.path.exists(idxs_path):
                idxs = np.load(idxs_path).tolist()
            else:
                idxs = filter_images(full_voc, labels, labels_old, overlap=overlap)
                if idxs_path is not None and (not torch.distributed.is_initialized() or torch.distributed.get_rank() == 0):
                    _save_index_file(idxs_path, idxs)
            if test_on_val:
                rnd = np.random.RandomState(1)
                rnd.shuffle(idxs)
                train_len = int(0.8 * len(idxs))
                idxs = idxs[:train_len] if train else idxs[train_len:]
            self.indices = idxs
            self.lut = datapipe.target_lut(labels, labels_old, masking=masking, data_masking=data_masking)
        else:
            self.indices = list(range(len(full_voc)))
            self.lut = torch.arange(256, dtype=torch.uint8)

    def __getitem__(self, index):
        img, lab = self.full[self.indices[index]]
        return torch.from_numpy(img), torch.from_numpy(lab)

    def __len__(self):
        return len(self.indices)


class _Incremental(data.Dataset):
    """What the three incremental datasets share (dataset/{voc,ade,cityscape}.py): filter the full listing by the step's labels
    (index file when present), keep the step's label table for the device, hand out raw uint8 pairs."""

    def _select(self, full, labels, labels_old, idxs_path, overlap):
        if idxs_path is not None and os.path.exists(idxs_path):
            return np.load(idxs_path).tolist()
        idxs = filter_images(full, labels, labels_old, overlap=overlap)
        if idxs_path is not None and (not torch.distributed.is_initialized() or torch.distributed.get_rank() == 0):
            _save_index_file(idxs_path, idxs)
        return idxs

    def __getitem__(self, index):
        img, lab = self.full[self.indices[index]]
        return torch.from_numpy(img), torch.from_numpy(lab)

    def __len__(self):
        return len(self.indices)


class AdeSegmentation(data.Dataset):
    """File listing of the reference's ``AdeSegmentation`` (dataset/ade.py:38-58): every file of
    ``ADEChallengeData2016/images/{training,validation}`` in sorted order, annotation = same name with ``png``."""

    def __init__(self, root, train=True, transform=None):
        ade_root = os.path.join(os.path.expanduser(root), "ADEChallengeData2016")
        split = "training" if train else "validation"
        image_folder = os.path.join(ade_root, "images", split)
        annotation_folder = os.path.join(ade_root, "annotations", split)
        if not os.path.isdir(image_folder):
            raise RuntimeError(f"Dataset not found or corrupted. at location = {image_folder}")
        self.images = [(os.path.join(image_folder, x), os.path.join(annotation_folder, x[:-3] + "png"))
                       for x in sorted(os.listdir(image_folder))]

    def __getitem__(self, index):
        return _open_rgb(self.images[index][0]), _open_label(self.images[index][1])

    def label(self, index):
        return _open_label(self.images[index][1])

    def __len__(self):
        return len(self.images)


class AdeSegmentationIncremental(_Incremental):
    """``AdeSegmentationIncremental(root, train, transform, labels, labels_old, idxs_path, masking, overlap, data_masking,
    ignore_test_bg)`` (dataset/ade.py:78-175); ``transform`` accepted and ignored (device batch transform), ``self.lut`` the
    step's label table (``datapipe.ade_target_lut``)."""

    def __init__(self, root, train=True, transform=None, labels=None, labels_old=None, idxs_path=None, masking=True,
                 overlap=True, data_masking="current", ignore_test_bg=False, **kwargs):
        self.full, self.train = AdeSegmentation(root, train), train
        self.labels, self.labels_old = [], []
        if labels is not None:
            labels = [l for l in labels if l != 0]
            labels_old = [l for l in (labels_old or []) if l != 0]
            assert not any(l in labels_old for l in labels), "labels and labels_old must be disjoint sets"
            self.labels, self.labels_old, self.order = labels, labels_old, [0] + labels_old + labels
            self.indices = self._select(self.full, labels, labels_old, idxs_path, overlap)
            self.lut = datapipe.ade_target_lut(labels, labels_old, masking=masking, ignore_test_bg=ignore_test_bg)
        else:
            self.indices = list(range(len(self.full)))
            self.lut = torch.arange(256, dtype=torch.uint8)


class CitySegmentation(data.Dataset):
    """File listing of ``CitySegmentation`` (dataset/cityscape.py:34-47, get_city_pairs :166-203): ``os.walk`` over
    ``Cityscapes/leftImg8bit/<split>``, every ``*.png`` whose ``gtFine/<split>/<city>/*_gtFine_labelIds.png`` exists, in walk
    order.  Items are the raw uint8 image and the RAW labelIds map; ``label(i)`` returns the class-index map the reference's
    ``_class_to_index`` makes (what ``filter_images`` looks at)."""

    def __init__(self, root, train=True):
        city_root = os.path.join(os.path.expanduser(root), "Cityscapes")
        split = "train" if train else "val"
        img_folder, mask_folder = os.path.join(city_root, "leftImg8bit/" + split), os.path.join(city_root, "gtFine/" + split)
        self.images, self.mask_paths = [], []
        for r, _dirs, files in os.walk(img_folder):
            for filename in files:
                if filename.endswith(".png"):
                    imgpath = os.path.join(r, filename)
                    maskpath = os.path.join(mask_folder, os.path.basename(os.path.dirname(imgpath)),
                                            filename.replace("leftImg8bit", "gtFine_labelIds"))
                    if os.path.isfile(imgpath) and os.path.isfile(maskpath):
                        self.images.append(imgpath)
                        self.mask_paths.append(maskpath)
        if len(self.images) == 0:
            raise RuntimeError("Found 0 images in subfolders of: " + city_root + "\n")
        self._class_lut = datapipe.city_class_lut().numpy()

    def __getitem__(self, index):
        return _open_rgb(self.images[index]), _open_label(self.mask_paths[index])

    def label(self, index):
        raw = _open_label(self.mask_paths[index])
        assert raw.max() <= 33, "Cityscapes labelIds are 0 .. 33 (dataset/cityscape.py:58-62 asserts the same)"
        return self._class_lut[raw]

    def __len__(self):
        return len(self.images)


class CitySegmentationIncremental(_Incremental):
    """``CitySegmentationIncremental(root, train, transform, labels, labels_old, idxs_path, masking, overlap)``
    (dataset/cityscape.py:103-163).  ``self.lut`` maps RAW labelIds straight to the step's ids: the class table composed with
    the step table, one gather on the device."""

    def __init__(self, root, train=True, transform=None, labels=None, labels_old=None, idxs_path=None, masking=True,
                 overlap=True, **kwargs):
        self.full, self.train = CitySegmentation(root, train), train
        self.labels, self.labels_old = [], []
        if labels is not None:
            labels = [l for l in labels if l != 0]
            labels_old = [l for l in (labels_old or []) if l != 0]
            assert not any(l in labels_old for l in labels), "labels and labels_old must be disjoint sets"
            self.labels, self.labels_old, self.order = [0] + labels, [0] + labels_old, [0] + labels_old + labels
            self.indices = self._select(self.full, labels, labels_old, idxs_path, overlap)
            step = datapipe.city_target_lut(labels, labels_old, masking=masking, train=train)
            self.lut = step[datapipe.city_class_lut().long()]
        else:
            self.indices = list(range(len(self.full)))
            self.lut = datapipe.city_class_lut()


def _identity(samples):
    return samples


class DeviceLoader:
    """The train / validation loader: ``torch.utils.data.DataLoader`` worker PROCESSES decode (Pillow) and hand raw uint8 pairs
    through shared, pinned memory; the batch transform runs in the main process on the device (``DeviceBatcher``) - no HIP call
    ever happens in a worker.  The random crop / flip parameters are drawn in the main process in sample order, i.e. the same
    ``random`` sequence as the reference's ``num_workers=0`` loader (argparser.py:53).  Iterating yields what the trainer
    takes; ``sampler`` / ``__len__`` as the reference's loader exposes them (train.py:93, run.py:189)."""

    def __init__(self, dataset, batch_size, sampler, batcher, num_workers=0, drop_last=False, prefetch_factor=4):
        kw = dict(persistent_workers=True, prefetch_factor=prefetch_factor) if num_workers > 0 else {}
        self.loader = data.DataLoader(dataset, batch_size=batch_size, sampler=sampler, num_workers=num_workers, drop_last=drop_last,
                                      collate_fn=_identity, pin_memory=num_workers > 0, **kw)
        self.sampler, self.batcher, self.batch_size = sampler, batcher, batch_size

    def __iter__(self):
        for samples in self.loader:
            yield self.batcher(samples)

    def __len__(self):
        return len(self.loader)


class DeviceBatcher:
    """``collate_fn`` of the loader: a list of raw ``(image u8 [H, W, 3], label u8 [H, W])`` pairs -> the batch the train step
    takes, ``(images float32 [B, 3, S, S] channels-last, labels int64 [B, S, S])`` on ``device``.

    train:  RandomResizedCrop(S, (0.5, 2.0)) + RandomHorizontalFlip + ToTensor + Normalize   (run.py:49-55) - parameters
            drawn on the host exactly like dataset/transform.py (``random`` module, per sample: crop box, then flip coin),
            pixels on the device;
    val:    Resize(S) + CenterCrop(S) + ToTensor + Normalize when ``crop`` (run.py:58-65; Pillow on the host, then the device
            path with the identity box), else ToTensor + Normalize on the full image (batch size 1, run.py:66-73)."""

    def __init__(self, device, size, lut, train=True, crop=True, scale=(0.5, 2.0), mean=(0.485, 0.456, 0.406),
                 std=(0.229, 0.224, 0.225)):
        self.device, self.size, self.train, self.crop = torch.device(device), int(size), train, crop
        self.scale = scale
        self.images = datapipe.DeviceImagePath(self.size, mean, std)
        self.labels = datapipe.DeviceLabelPath(self.size, lut)
        self.mean, self.std, self.lut = mean, std, lut

    def __call__(self, samples):
        dev, S = self.device, self.size
        if self.train:
            boxes, flips = [], []
            for img, _ in samples:
                boxes.append(datapipe.random_resized_crop_params(img.shape[0], img.shape[1], scale=self.scale))
                flips.append(random.random() < 0.5)
            imgs = [s[0].to(dev, non_blocking=True) for s in samples]
            labs = [s[1].to(dev, non_blocking=True) for s in samples]
            return self.images(imgs, boxes, flips), self.labels(labs, boxes, flips)
        if self.crop:
            from PIL import Image
            imgs, labs = [], []
            for img, lab in samples:
                pi, pl = Image.fromarray(img.numpy()), Image.fromarray(lab.numpy())
                w, h = pi.size
                if w <= h:
                    ow, oh = S, int(S * h / w)
                else:
                    oh, ow = S, int(S * w / h)                       # transform.Resize(size): smaller edge -> size
                pi, pl = pi.resize((ow, oh), Image.BILINEAR), pl.resize((ow, oh), Image.NEAREST)
                i, j = int(round((oh - S) / 2.0)), int(round((ow - S) / 2.0))          # transform.CenterCrop
                imgs.append(torch.from_numpy(np.asarray(pi.crop((j, i, j + S, i + S)), dtype=np.uint8).copy()).to(dev))
                labs.append(torch.from_numpy(np.asarray(pl.crop((j, i, j + S, i + S)), dtype=np.uint8).copy()).to(dev))
            boxes, flips = [(0, 0, S, S)] * len(samples), [False] * len(samples)
            return self.images(imgs, boxes, flips), self.labels(labs, boxes, flips)
        out_i, out_l = [], []
        for img, lab in samples:                                     # full-size validation: per image
            x = img.to(dev).permute(2, 0, 1).float().div_(255.0)
            m = torch.tensor(self.mean, device=dev).view(3, 1, 1)
            s = torch.tensor(self.std, device=dev).view(3, 1, 1)
            out_i.append(((x - m) / s).unsqueeze(0).contiguous(memory_format=torch.channels_last))
            out_l.append(self.lut.to(dev)[lab.to(dev).long()].long().unsqueeze(0))
        return torch.cat(out_i), torch.cat(out_l)
