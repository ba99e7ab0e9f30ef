"""Losses of the ``--method UCD`` step, reference interface (utils/loss.py).

``UnbiasedCrossEntropy`` (utils/loss.py:89-109) and ``UnbiasedKnowledgeDistillationLoss``
(utils/loss.py:139-184) keep the reference's module interface on full-resolution logits (the unfused path:
``--alpha`` != 1, plain KD, tests).  The training step itself calls ``fused_seg_losses``: bilinear x16
up-sampling + UnbiasedCE + UnbiasedKD + the gradient w.r.t. the LOW-resolution logits in one HIP kernel
(``ucd_seg_losses``, csrc/seglogit_loss.hip) - the ``[B, Ctot, H, W]`` tensors never exist.
The contrastive loss lives in :mod:`ucd_amd.contrastive`.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hip
from .contrastive import PixelConLossV2, pre_contractive_pixel, ucd_contrastive_loss  # noqa: F401


class _FusedSegLosses(torch.autograd.Function):
    """total = ce_weight * mean(UnbiasedCE) + kd_weight * mean(UnbiasedKD) from the LOW-resolution logits;
    the up-sampled [B, Ctot, H, W] tensors never exist (ucd_seg_losses)."""

    @staticmethod
    def forward(ctx, sem, sem_old, labels, old_cl, ce_weight, kd_weight, ignore_index):
        lib = hip.load()
        B, Ctot, h, w = sem.shape
        H, W = labels.shape[-2:]
        s = sem.detach().permute(0, 2, 3, 1).reshape(B * h * w, Ctot).float().contiguous()
        t, K = None, int(old_cl)
        if sem_old is not None:
            K = sem_old.shape[1]
            t = sem_old.detach().permute(0, 2, 3, 1).reshape(B * h * w, K).float().contiguous()
        labels = labels.contiguous()
        out = torch.empty(2, dtype=torch.float32, device=sem.device)
        d = torch.empty(B * h * w, Ctot, dtype=torch.float32, device=sem.device)
        nbytes = lib.ucd_seg_losses_workspace_bytes(B, H, W)
        ws = hip.workspace(nbytes, sem.device, "seglosses")
        with hip._timed("ucd_seg_losses", B * H * W * 8 + 2 * B * h * w * (2 * Ctot + K) * 4):
            hip._check(lib.ucd_seg_losses(hip.ptr(s), Ctot, hip.ptr(t), K, hip.ptr(labels), B, H, W, h, w, Ctot, max(K, 1),
                                          int(ignore_index), float(ce_weight), float(kd_weight), hip.ptr(out), hip.ptr(d),
                                          Ctot, hip.ptr(ws), nbytes, hip.stream()), "ucd_seg_losses")
        ctx.save_for_backward(d)
        ctx.meta = (B, Ctot, h, w, sem.dtype)
        ce, kd = out[0], out[1]
        total = ce_weight * ce + kd_weight * kd
        ctx.mark_non_differentiable(ce, kd)
        return total, ce, kd

    @staticmethod
    def backward(ctx, g, _gce, _gkd):
        (d,) = ctx.saved_tensors
        B, Ctot, h, w, dtype = ctx.meta
        grad = (d * g).view(B, h, w, Ctot).permute(0, 3, 1, 2).to(dtype)
        return grad, None, None, None, None, None, None


def fused_seg_losses(sem, sem_old, labels, old_cl, ce_weight=1.0, kd_weight=0.0, ignore_index=255):
    """Returns (ce_weight*CE + kd_weight*KD [differentiable w.r.t. ``sem``], CE, KD) where CE / KD are the
    reference's ``UnbiasedCrossEntropy(...)(up(sem), labels).mean()`` and
    ``UnbiasedKnowledgeDistillationLoss()(up(sem), up(sem_old))`` (``up`` = bilinear to the label size)."""
    if not sem.is_cuda:
        raise RuntimeError("ucd_amd.loss.fused_seg_losses runs on the GPU only (there is no CPU fallback)")
    return _FusedSegLosses.apply(sem, sem_old, labels, old_cl, ce_weight, kd_weight, ignore_index)


class UnbiasedCrossEntropy(nn.Module):
    """Cross entropy in which the background competes as the pooled old classes:
    ``log p(bkg) = LSE(x[:, :old_cl]) - LSE(x)``; labels below ``old_cl`` count as background."""

    def __init__(self, old_cl=None, reduction="mean", ignore_index=255):
        super().__init__()
        self.reduction, self.ignore_index, self.old_cl = reduction, ignore_index, old_cl

    def forward(self, inputs, targets):
        old_cl = self.old_cl
        inputs = inputs.float()
        den = torch.logsumexp(inputs, dim=1)
        log_bkg = torch.logsumexp(inputs[:, :old_cl], dim=1) - den
        # gather instead of materialising the [B, Ctot, H, W] log-probability tensor (loss.py:99-102)
        labels = torch.where(targets < old_cl, torch.zeros_like(targets), targets)   # loss.py:104-105
        ignore = labels == self.ignore_index
        idx = torch.where(ignore, torch.zeros_like(labels), labels)
        picked = inputs.gather(1, idx.unsqueeze(1)).squeeze(1) - den
        logp = torch.where(idx == 0, log_bkg, picked)
        loss = torch.where(ignore, torch.zeros_like(logp), -logp)
        if self.reduction == "none":
            return loss
        if self.reduction == "sum":
            return loss.sum()
        return loss.sum() / (~ignore).sum()      # nll_loss 'mean': over the non-ignored pixels


class KnowledgeDistillationLoss(nn.Module):
    """Plain soft-target distillation on the old classes (utils/loss.py:112-136)."""

    def __init__(self, reduction="mean", alpha=1.):
        super().__init__()
        self.reduction, self.alpha = reduction, alpha

    def forward(self, inputs, targets, mask=None):
        inputs = inputs.narrow(1, 0, targets.shape[1]).float()
        loss = (torch.log_softmax(inputs, dim=1) * torch.softmax(targets.float() * self.alpha, dim=1)).mean(dim=1)
        if mask is not None:
            loss = loss * mask.float()
        if self.reduction == "mean":
            return -loss.mean()
        if self.reduction == "sum":
            return -loss.sum()
        return -loss


class UnbiasedKnowledgeDistillationLoss(nn.Module):
    """The student's background is compared with the teacher's as ``p(bkg or any new class)``
    (utils/loss.py:162-174).  The reference also evaluates an unused ``gamma`` from a global average pool
    (:155-156); it never reaches the output and is dropped."""

    def __init__(self, reduction="mean", alpha=1.):
        super().__init__()
        self.reduction, self.alpha = reduction, alpha

    def forward(self, inputs, targets, mask=None):
        K = targets.shape[1]
        inputs, targets = inputs.float(), targets.float() * self.alpha
        den = torch.logsumexp(inputs, dim=1)
        out_old = inputs[:, 1:K] - den.unsqueeze(1)
        # LSE over {background} U {new classes}: index_select-free
        bkg_new = torch.cat((inputs[:, :1], inputs[:, K:]), dim=1)
        out_bkg = torch.logsumexp(bkg_new, dim=1) - den
        q = torch.softmax(targets, dim=1)
        loss = (q[:, 0] * out_bkg + (q[:, 1:] * out_old).sum(dim=1)) / K
        if mask is not None:
            loss = loss * mask.float()
        if self.reduction == "mean":
            return -loss.mean()
        if self.reduction == "sum":
            return -loss.sum()
        return -loss
