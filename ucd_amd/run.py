"""Driver: ``python -m torch.distributed.run --nproc_per_node=N run.py --method UCD ...`` (reference
run.py:116-400; launcher README.md:35).  Keeps the reference's control flow, optimiser / scheduler
semantics (run.py:175-189) and checkpoint layout (run.py:32-43: ``checkpoints/step/{task}-{dataset}_
{name}_{step}.pth`` with keys epoch, model_state [``module.``-prefixed], optimizer_state, scheduler_state,
best_score, trainer_state).  Data: with a VOC tree under ``--data_root`` (``splits/``, ``JPEGImages/``,
``SegmentationClassAug/``) the reference's incremental dataset is used - host decode, batch transform on the device
(``ucd_amd/dataset.py``); ``--data_root synthetic`` - and only that - trains on the closed-form
synthetic batches the benchmark uses; a data root without ``splits/`` raises like the reference's dataset classes do.
"""
from __future__ import annotations

import os

from . import switches as _switches
import random

import numpy as np
import torch
import torch.distributed as dist

from . import argparser, synth, tasks
from .ddp import DistributedDataParallel
from .logger import Logger
from .metrics import StreamSegMetrics
from .scheduler import PolyLR
from .segmentation_module import make_model
from .train import Trainer


def save_ckpt(path, model, trainer, optimizer, scheduler, epoch, best_score):
    """Same dictionary as the reference (run.py:32-43)."""
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    torch.save({"epoch": epoch, "model_state": model.state_dict(), "optimizer_state": optimizer.state_dict(),
                "scheduler_state": scheduler.state_dict(), "best_score": best_score,
                "trainer_state": trainer.state_dict()}, path)


def make_optimizer(opts, model):
    """Three parameter groups (body unless --freeze, head, cls), trainable tensors only, SGD with
    momentum 0.9 and Nesterov (run.py:175-186)."""
    net = model.module if hasattr(model, "module") else model
    groups = []
    if not opts.freeze:
        groups.append({"params": [p for p in net.body.parameters() if p.requires_grad], "weight_decay": opts.weight_decay})
    groups.append({"params": [p for p in net.head.parameters() if p.requires_grad], "weight_decay": opts.weight_decay})
    groups.append({"params": [p for p in net.cls.parameters() if p.requires_grad], "weight_decay": opts.weight_decay})
    if next(net.parameters()).is_cuda and _switches.get("UCD_SGD", "hip") != "torch":
        # a torch.optim.SGD subclass (same groups, state_dict, hooks, schedulers) whose step is ONE launch (csrc/sgd.hip) that
        # also writes the bf16 working weights; bit-exact against the rule in float64 (tests/test_optim.py).  UCD_SGD=torch
        # selects torch's fused step (the A/B reference of the tests).
        from .optim import SGD
        return SGD(groups, lr=opts.lr, momentum=0.9, nesterov=True)
    kw = {"fused": True} if next(net.parameters()).is_cuda else {}
    return torch.optim.SGD(groups, lr=opts.lr, momentum=0.9, nesterov=True, **kw)


class SyntheticSegmentation(torch.utils.data.Dataset):
    """Closed-form stand-in for {Voc,Ade,City}SegmentationIncremental at a given step: normalised images,
    labels in {0, new ids, 255} (old classes already mapped to background, dataset/voc.py:182-208)."""

    def __init__(self, n, crop, new_ids, seed=0):
        self.n, self.crop, self.new_ids, self.seed = n, crop, list(new_ids), seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        img = synth.images(self.seed * 100003 + i, 1, self.crop)[0]
        lab = synth.seg_labels(self.seed * 100003 + i, 1, self.crop, self.crop, self.new_ids)[0]
        return img, lab


def build_models(opts, device, classes):
    """Student (all heads) and, for step > 0, the teacher (previous heads) on ``device``."""
    model = make_model(opts, classes=classes).to(device)
    model_old = make_model(opts, classes=classes[:-1]).to(device) if opts.step > 0 else None
    if opts.fix_bn:
        model.fix_bn()
    if device.type == "cuda":
        # activations are channels-last; weights in the same format spare MIOpen one layout copy per 3x3
        # convolution call and let gradients be adopted without a re-striding copy
        model = model.to(memory_format=torch.channels_last)
        if model_old is not None:
            model_old = model_old.to(memory_format=torch.channels_last)
    return model, model_old


def load_step_checkpoint(opts, model, model_old, state, device):
    """run.py:207-233: previous-step weights into student and teacher (strict=False because of the new
    head), balanced init of the new head, teacher frozen in eval mode."""
    target = model.module if hasattr(model, "module") else model
    strip = lambda sd: {k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()}
    target.load_state_dict(strip(state), strict=False)
    if opts.init_balanced:
        target.init_new_classifier(device)
    old = model_old.module if hasattr(model_old, "module") else model_old
    old.load_state_dict(strip(state), strict=False)
    for p in model_old.parameters():
        p.requires_grad = False
    model_old.eval()


def _incremental_datasets():
    from .dataset import AdeSegmentationIncremental, CitySegmentationIncremental, VOCSegmentationIncremental
    return {"voc": VOCSegmentationIncremental, "ade": AdeSegmentationIncremental, "city": CitySegmentationIncremental}


class _Lazy(dict):
    def __missing__(self, key):
        self.update(_incremental_datasets())
        if key not in self:
            raise NotImplementedError(key)                       # run.py:84-85
        return self[key]


DATASETS = _Lazy()


def loader_workers(opts, world_size):
    """Decode processes per rank.  The reference's default is 0 (argparser.py:53: decode + transform in the training process,
    ~37 img/s - it cannot feed one GPU); ``--num_workers N`` is honoured, and 0 with real data means "enough": the host's
    cores shared between the ranks, at most 16 (profiles/r03_loader_bench.txt: 8 workers decode ~3x what the step consumes)."""
    if opts.num_workers > 0:
        return opts.num_workers
    return max(1, min(16, (os.cpu_count() or 8) // max(1, world_size)))


def get_dataset(opts, device, world_size, rank, labels, labels_old):
    """The reference's ``get_dataset`` + loader construction (run.py:46-113, 150-164) on the device data pipeline: returns
    (train_loader, val_dst or None, val_loader_of(dataset) -> loader, real).  ``--data_root synthetic`` - and only that - gives
    the closed-form synthetic batches; a root without the dataset's directory raises like the reference's classes."""
    real = opts.data_root != "synthetic"
    if not real:
        train_dst = SyntheticSegmentation(24 * 8, opts.crop_size, [l for l in labels if l != 0] or [1], seed=opts.step)
        sampler = torch.utils.data.distributed.DistributedSampler(train_dst, num_replicas=world_size, rank=rank)
        train_loader = torch.utils.data.DataLoader(train_dst, batch_size=opts.batch_size, sampler=sampler,
                                                   num_workers=opts.num_workers, drop_last=True)

        def val_loader_of(dst):
            return torch.utils.data.DataLoader(
                dst, batch_size=opts.batch_size if opts.crop_val else 1, num_workers=opts.num_workers,
                sampler=torch.utils.data.distributed.DistributedSampler(dst, num_replicas=world_size, rank=rank, shuffle=False))
        return train_loader, None, val_loader_of, False
    marker = {"voc": "splits", "ade": "ADEChallengeData2016", "city": "Cityscapes"}.get(opts.dataset)
    if marker is None:
        raise NotImplementedError(opts.dataset)
    if not os.path.isdir(os.path.join(opts.data_root, marker)):
        # dataset/voc.py:58-59: a mistyped or unmounted root must not train (and checkpoint) on synthetic batches
        raise RuntimeError(f"Dataset not found or corrupted. at location = {opts.data_root} (no {marker}/ directory; pass "
                           "--data_root synthetic for the closed-form synthetic batches)")
    if device.type != "cuda":
        raise RuntimeError("the device data pipeline needs a GPU (there is no CPU fallback)")
    from .dataset import DeviceBatcher, DeviceLoader
    dataset = DATASETS[opts.dataset]
    _, _, path_base = tasks.get_task_labels(opts.dataset, opts.task, opts.step)
    path_base += "-ov" if opts.overlap else ""
    os.makedirs(path_base, exist_ok=True)
    train_dst = dataset(opts.data_root, train=True, labels=list(labels), labels_old=list(labels_old),
                        idxs_path=path_base + f"/train-{opts.step}.npy", masking=not opts.no_mask, overlap=opts.overlap)
    train_lut = train_dst.lut
    workers = loader_workers(opts, world_size)
    collate_train = DeviceBatcher(device, opts.crop_size, train_lut, train=True)
    val_dst = None
    if not opts.no_cross_val:
        # run.py:95-97: the validation subset is a random split of the TRAINING set, so it goes through the training transform
        # (random crop to crop_size + flip) - batches of opts.batch_size crops
        train_len = int(0.8 * len(train_dst))
        train_dst, val_dst = torch.utils.data.random_split(train_dst, [train_len, len(train_dst) - train_len])
        cross_val_batcher = collate_train
    else:
        val_dst = dataset(opts.data_root, train=False, labels=list(labels), labels_old=list(labels_old),
                          idxs_path=path_base + f"/val-{opts.step}.npy", masking=not opts.no_mask, overlap=True)
        cross_val_batcher = None
    sampler = torch.utils.data.distributed.DistributedSampler(train_dst, num_replicas=world_size, rank=rank)
    train_loader = DeviceLoader(train_dst, opts.batch_size, sampler, collate_train, num_workers=workers, drop_last=True)

    def val_loader_of(dst):
        lut = dst.dataset.lut if hasattr(dst, "dataset") else dst.lut
        batcher = cross_val_batcher if (cross_val_batcher is not None and hasattr(dst, "dataset")) else \
            DeviceBatcher(device, opts.crop_size, lut, train=False, crop=opts.crop_val)
        full_size = batcher is not cross_val_batcher and not opts.crop_val
        return DeviceLoader(dst, 1 if full_size else opts.batch_size,
                            torch.utils.data.distributed.DistributedSampler(dst, num_replicas=world_size, rank=rank, shuffle=False),
                            batcher, num_workers=min(workers, 4))
    return train_loader, val_dst, val_loader_of, True


def main(opts):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", opts.MASTER_PORT)
    use_cuda = torch.cuda.is_available()
    local_rank = int(os.environ.get("LOCAL_RANK", opts.local_rank))
    if "RANK" in os.environ or "WORLD_SIZE" in os.environ:
        dist.init_process_group(backend="nccl" if use_cuda else "gloo")
    else:
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl" if use_cuda else "gloo", rank=0, world_size=1)
    device = torch.device("cuda", local_rank) if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(device)
    rank, world_size = dist.get_rank(), dist.get_world_size()
    task_name = f"{opts.task}-{opts.dataset}"
    logger = Logger(f"{opts.logdir}/{task_name}/{opts.name}/", rank=rank, debug=opts.debug, step=opts.step)
    logger.info(f"Device: {device}; total batch size {opts.batch_size * world_size}")
    torch.manual_seed(opts.random_seed); np.random.seed(opts.random_seed); random.seed(opts.random_seed)

    classes = tasks.get_per_task_classes(opts.dataset, opts.task, opts.step)
    labels, labels_old, _ = tasks.get_task_labels(opts.dataset, opts.task, opts.step)
    train_loader, val_dst, val_loader_of, real = get_dataset(opts, device, world_size, rank, labels, labels_old)
    model, model_old = build_models(opts, device, classes)
    optimizer = make_optimizer(opts, model)
    scheduler = PolyLR(optimizer, max_iters=opts.epochs * len(train_loader), power=opts.lr_power)
    model = DistributedDataParallel(model, delay_allreduce=True,
                                    bf16_weights=getattr(opts, "opt_level", "O0") != "O0" and getattr(opts, "bf16_weights", True))
    if opts.step > 0:
        path = opts.step_ckpt or f"checkpoints/step/{task_name}_{opts.name}_{opts.step - 1}.pth"
        if os.path.exists(path):
            ckpt = torch.load(path, map_location="cpu")
            load_step_checkpoint(opts, model, model_old, ckpt["model_state"], device)
            logger.info(f"[!] Previous model loaded from {path}")
        elif not opts.debug:
            raise FileNotFoundError(path)
        for p in model_old.parameters():
            p.requires_grad = False
        model_old.eval()
    trainer = Trainer(model, model_old, device=device, opts=opts, classes=classes)
    # validation set + streaming metrics (run.py:161-164, 304-338): synthetic images carrying every class seen so far
    n_classes = sum(classes)
    seen = [l for l in (list(labels_old) + list(labels)) if l != 0] or [1]
    if val_dst is None:
        val_dst = SyntheticSegmentation(max(2 * opts.batch_size, 2), opts.crop_size, seen, seed=1000 + opts.step)
    val_loader = val_loader_of(val_dst)
    val_metrics = StreamSegMetrics(n_classes)

    cur_epoch, best_score = 0, 0.0
    if opts.ckpt is not None and os.path.isfile(opts.ckpt):
        ckpt = torch.load(opts.ckpt, map_location="cpu")
        model.load_state_dict(ckpt["model_state"], strict=True)
        optimizer.load_state_dict(ckpt["optimizer_state"])
        scheduler.load_state_dict(ckpt["scheduler_state"])
        cur_epoch, best_score = ckpt["epoch"] + 1, ckpt["best_score"]
    ckpt_path = f"checkpoints/step/{task_name}_{opts.name}_{opts.step}.pth"
    while cur_epoch < opts.epochs and not opts.test:
        epoch_loss = trainer.train(cur_epoch=cur_epoch, optim=optimizer, train_loader=train_loader,
                                   scheduler=scheduler, print_int=opts.print_interval, logger=logger)
        logger.info(f"End of Epoch {cur_epoch}/{opts.epochs}, Average Loss={float(epoch_loss[0]) + float(epoch_loss[1])}")
        if (cur_epoch + 1) % opts.val_interval == 0:                     # run.py:304-316
            logger.info("validate on val set...")
            model.eval()
            val_loss, val_score, _ = trainer.validate(loader=val_loader, metrics=val_metrics, logger=logger)
            logger.info(f"End of Validation {cur_epoch}/{opts.epochs}, Validation Loss={float(val_loss[0]) + float(val_loss[1])},"
                        f" Class Loss={float(val_loss[0])}, Reg Loss={float(val_loss[1])}")
            if rank == 0:
                logger.info(val_metrics.to_str(val_score))
                best_score = val_score["Mean IoU"]
            model.train()
        if rank == 0 and (cur_epoch + 1) % opts.ckpt_interval == 0:
            save_ckpt(ckpt_path, model, trainer, optimizer, scheduler, cur_epoch, best_score)
        dist.barrier()
        cur_epoch += 1
    if rank == 0 and not opts.test:
        save_ckpt(ckpt_path, model, trainer, optimizer, scheduler, cur_epoch, best_score)
    dist.barrier()
    if real:
        # final pass over the test split with every class seen so far (run.py:108-111, 340-372)
        image_set = "train" if opts.val_on_trainset else "val"
        _, _, path_base = tasks.get_task_labels(opts.dataset, opts.task, opts.step)
        path_base += "-ov" if opts.overlap else ""
        test_dst = DATASETS[opts.dataset](opts.data_root, train=opts.val_on_trainset, labels=list(labels_old) + list(labels),
                                          idxs_path=path_base + f"/test_on_{image_set}-{opts.step}.npy")
        test_loader = val_loader_of(test_dst)
        model.eval()
        val_loss, val_score, _ = trainer.validate(loader=test_loader, metrics=val_metrics, logger=logger)
        logger.info(f"*** End of Test, Total Loss={float(val_loss[0]) + float(val_loss[1])}, Class Loss={float(val_loss[0])},"
                    f" Reg Loss={float(val_loss[1])}")
        if rank == 0:
            logger.info(val_metrics.to_str(val_score))
    dist.destroy_process_group()


def cli():
    opts = argparser.modify_command_options(argparser.get_argparser().parse_args())
    os.makedirs("checkpoints/step", exist_ok=True)
    main(opts)


if __name__ == "__main__":
    cli()
