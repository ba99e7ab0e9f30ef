"""Command line of the reference (argparser.py:46-203): same flag names, defaults and ``--method``
presets, declared from a table.  Differences, both documented in SURVEY.md section 0: ``UCD`` is an
accepted ``--method`` (the reference's parser rejects the preset its own code handles), and
``--opt_level`` selects the activation precision of this build (O0 = fp32 like apex O0, O1..O3 = bf16
autocast with fp32 master weights) instead of an apex mode."""
from __future__ import annotations

import argparse

from . import tasks

# method -> option overrides (reference argparser.py:15-39)
METHOD_PRESETS = {
    "FT": {},
    "LWF": {"loss_kd": 100},
    "LWF-MC": {"icarl": True, "icarl_importance": 10},
    "ILT": {"loss_kd": 100, "loss_de": 100},
    "EWC": {"regularizer": "ewc", "reg_importance": 500},
    "RW": {"regularizer": "rw", "reg_importance": 100},
    "PI": {"regularizer": "pi", "reg_importance": 500},
    "MiB": {},
    "att": {},
    "UCD": {"loss_kd": 10, "unce": True, "unkd": True, "init_balanced": True},
}
NUM_CLASSES = {"voc": 21, "ade": 150, "city": 20}


def modify_command_options(opts):
    if opts.dataset in NUM_CLASSES:
        opts.num_classes = NUM_CLASSES[opts.dataset]
    if not opts.visualize:
        opts.sample_num = 0
    for key, value in METHOD_PRESETS.get(opts.method, {}).items():
        setattr(opts, key, value)
    opts.no_overlap = not opts.overlap
    opts.no_cross_val = not opts.cross_val
    return opts


def _flag(name, **kw):
    return (name, kw)


_ON, _OFF = dict(action="store_true", default=False), dict(action="store_false", default=True)
_ARGS = [
    # performance
    _flag("--local_rank", type=int, default=0), _flag("--random_seed", type=int, default=42),
    _flag("--num_workers", type=int, default=0),
    # dataset
    _flag("--data_root", type=str, default="data"),
    _flag("--dataset", type=str, default="voc", choices=["voc", "ade", "city"]),
    _flag("--num_classes", type=int, default=None),
    # method (overrides other parameters)
    _flag("--method", type=str, default=None, choices=list(METHOD_PRESETS)),
    # train
    _flag("--epochs", type=int, default=30), _flag("--fix_bn", **_ON),
    _flag("--batch_size", type=int, default=4), _flag("--crop_size", type=int, default=512),
    _flag("--lr", type=float, default=0.007), _flag("--momentum", type=float, default=0.9),
    _flag("--weight_decay", type=float, default=1e-4),
    _flag("--lr_policy", type=str, default="poly", choices=["poly", "step"]),
    _flag("--lr_decay_step", type=int, default=5000), _flag("--lr_decay_factor", type=float, default=0.1),
    _flag("--lr_power", type=float, default=0.9), _flag("--bce", **_ON),
    # validation
    _flag("--val_on_trainset", **_ON), _flag("--cross_val", **_ON), _flag("--crop_val", **_OFF),
    # logging
    _flag("--logdir", type=str, default="./logs"), _flag("--name", type=str, default="Experiment"),
    _flag("--sample_num", type=int, default=0), _flag("--debug", **_ON), _flag("--visualize", **_OFF),
    _flag("--print_interval", type=int, default=10), _flag("--val_interval", type=int, default=1),
    _flag("--ckpt_interval", type=int, default=1),
    # model
    _flag("--backbone", type=str, default="resnet101", choices=["resnet50", "resnet101"]),
    _flag("--output_stride", type=int, default=16, choices=[8, 16]), _flag("--no_pretrained", **_ON),
    _flag("--norm_act", type=str, default="iabn_sync", choices=["iabn_sync", "iabn", "abn", "std"]),
    _flag("--fusion-mode", metavar="NAME", type=str, choices=["mean", "voting", "max"], default="mean"),
    _flag("--pooling", type=int, default=32), _flag("--temperature", type=float, default=0.07),
    # test / checkpoints
    _flag("--test", **_ON), _flag("--ckpt", default=None, type=str),
    # distillation (ILT)
    _flag("--freeze", **_ON), _flag("--loss_de", type=float, default=0.), _flag("--loss_kd", type=float, default=0.),
    # regularisers (EWC / RW / PI): accepted for CLI compatibility, not on the UCD path
    _flag("--regularizer", default=None, type=str, choices=["ewc", "rw", "pi"]),
    _flag("--reg_importance", type=float, default=1.), _flag("--reg_alpha", type=float, default=0.9),
    _flag("--reg_no_normalize", **_ON), _flag("--reg_iterations", type=int, default=10),
    # iCaRL
    _flag("--icarl", **_ON), _flag("--icarl_importance", type=float, default=1.),
    _flag("--icarl_disjoint", **_ON), _flag("--icarl_bkg", **_ON),
    # MiB / UCD switches
    _flag("--init_balanced", **_ON), _flag("--unkd", **_ON), _flag("--alpha", default=1., type=float),
    _flag("--unce", **_ON),
    # incremental protocol
    _flag("--task", type=str, default="19-1", choices=tasks.get_task_list()),
    _flag("--step", type=int, default=0), _flag("--no_mask", **_ON), _flag("--overlap", **_ON),
    _flag("--step_ckpt", default=None, type=str),
    _flag("--opt_level", type=str, choices=["O0", "O1", "O2", "O3"], default="O0"),
    _flag("--MASTER_PORT", type=str, default="29501"),
]


def get_argparser():
    parser = argparse.ArgumentParser(description="UCD incremental segmentation (MI355X-native hot path)")
    for name, kw in _ARGS:
        parser.add_argument(name, **kw)
    return parser
