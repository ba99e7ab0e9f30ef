"""Oracle (test infrastructure, not product): one UCD training iteration on CPU.

Restates the *intended* composition of ``Trainer.train`` for ``--method UCD`` at step >= 1
(train.py:95-151 with the unpacking at :115-116 repaired as SURVEY.md section 0 describes):

    teacher (eval, no grad)  ->  student (train)  ->
    loss     = mean(UnbiasedCE(out, labels)) + PixelConLossV2(pre_contractive_pixel(...)) / 100
    lkd      = loss_kd * UnbiasedKD(out, out_old)          (loss_kd = 10, argparser.py:35-39)
    loss_tot = loss + lkd ; backward ; SGD(momentum 0.9, nesterov, wd 1e-4) ; PolyLR per iteration
                                                                    (run.py:175-189, scheduler.py:3-10)
Pinned by tests/golden/ucd_step.npz (hand-composed from the reference's own classes).
"""
from __future__ import annotations

import torch

from . import contrastive, losses, model


def make_params(state, requires_grad=True):
    """Float leaves for every parameter, plain tensors for the running statistics."""
    P = {}
    for k, v in state.items():
        t = v.detach().clone()
        if requires_grad and t.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            t.requires_grad_(True)
        P[k] = t
    return P


def ucd_losses(Ps, Pt, images, labels, classes, temperature=0.07, loss_kd=10.0, max_label=20,
               pooling_size=32):
    """Returns dict(loss, lkd, ce, con, A, C) for student params ``Ps`` / teacher params ``Pt``."""
    n_heads = len(classes)
    old_cl = sum(classes[:-1])
    with torch.no_grad():
        out_old, feat_old = model.segmentation_forward(images, Pt, n_heads - 1, training=False,
                                                       pooling_size=pooling_size)
    out, feat = model.segmentation_forward(images, Ps, n_heads, training=True, pooling_size=pooling_size)
    prep = contrastive.pre_contrastive_pixel(feat["pre_logits"], labels, feat_old["sem"],
                                             feat_old["pre_logits"], max_label=max_label)
    con = contrastive.pixcon_loss(prep["a"], prep["c"], prep["la"], prep["lc"], prep["P"], temperature)
    ce = losses.unbiased_cross_entropy(out, labels, old_cl).mean()
    loss = ce + con / 100                                                       # train.py:116
    lkd = loss_kd * losses.unbiased_kd(out, out_old)                            # train.py:131-133
    return {"loss": loss, "lkd": lkd, "ce": ce, "con": con, "A": prep["a"].shape[0],
            "C": prep["c"].shape[0], "logits": out, "logits_old": out_old}


def poly_lr(base_lr, it, max_iters, power=0.9):
    return base_lr * (1 - it / max_iters) ** power                              # scheduler.py:9-10
