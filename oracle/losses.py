"""Oracle (test infrastructure, not product): the full-resolution logit losses of the UCD step.

CPU restatements of
  * ``UnbiasedCrossEntropy.forward``                      utils/loss.py:96-109
  * ``UnbiasedKnowledgeDistillationLoss.forward``         utils/loss.py:148-184
Pinned by tests/golden/logit_losses.npz (outputs of the reference classes).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def unbiased_cross_entropy(logits, targets, old_cl: int, ignore_index: int = 255, reduction: str = "none"):
    """logits [B, Ctot, H, W], targets [B, H, W] int64.  Background/old probability mass is pooled:
    ``out[:, 0] = LSE(x[:, :old_cl]) - LSE(x)``, new classes keep ``x - LSE(x)`` (loss.py:99-102);
    labels below ``old_cl`` count as background (loss.py:104-105; the reference rewrites ``targets``
    in place - this restatement leaves the caller's tensor alone)."""
    den = torch.logsumexp(logits, dim=1)
    out = torch.zeros_like(logits)
    out[:, 0] = torch.logsumexp(logits[:, :old_cl], dim=1) - den
    out[:, old_cl:] = logits[:, old_cl:] - den.unsqueeze(1)
    labels = torch.where(targets < old_cl, torch.zeros_like(targets), targets)
    return F.nll_loss(out, labels, ignore_index=ignore_index, reduction=reduction)


def unbiased_kd(logits, teacher_logits, alpha: float = 1.0, reduction: str = "mean", mask=None):
    """logits [B, Ctot, H, W] (student), teacher_logits [B, K, H, W].  The student's background
    competes with its new classes: ``out_bkg = LSE(x[:, {0} U new]) - LSE(x)``; old classes
    ``x[:, 1:K] - LSE(x)``; targets ``softmax(alpha * teacher)``; per-pixel loss averaged over the K
    teacher classes (loss.py:162-174).  The reference also computes an unused ``gamma`` from a global
    average pool (loss.py:155-156) - not restated, it does not reach the output."""
    K = teacher_logits.shape[1]
    Ctot = logits.shape[1]
    t = teacher_logits * alpha
    den = torch.logsumexp(logits, dim=1)
    out_old = logits[:, 1:K] - den.unsqueeze(1)
    idx = torch.tensor([0] + list(range(K, Ctot)), dtype=torch.long)
    out_bkg = torch.logsumexp(torch.index_select(logits, 1, idx), dim=1) - den
    q = torch.softmax(t, dim=1)
    loss = (q[:, 0] * out_bkg + (q[:, 1:] * out_old).sum(dim=1)) / K
    if mask is not None:
        loss = loss * mask.float()
    if reduction == "mean":
        return -loss.mean()
    if reduction == "sum":
        return -loss.sum()
    return -loss
