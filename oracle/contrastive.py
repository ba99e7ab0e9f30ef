"""Oracle (test infrastructure, not product): uncertainty-weighted pixel-contrastive distillation.

CPU restatement of the wired contrastive path of the reference:
  * ``pre_contractive_pixel`` live branch ``version == 'v2'``   utils/utils.py:256-393
    (byte-identical twin ``pre_contrastive_pixel``               utils/loss.py:258-395)
  * ``PixelConLossV2.forward``                                     utils/loss.py:412-466
  * the dead-file v1 losses ``PixelConLoss`` / ``SupConLoss``      utils/loss_new.py:264-400
Pinned against the reference's own outputs: tests/golden/pixcon_*.npz (tests/test_oracle_golden.py).

Everything is dense [A, C] fp32 arithmetic in the same op order as the reference, so the forward
agrees with it bit for bit on CPU; ``pixcon_loss_backward`` is the closed-form gradient of
SURVEY.md section 8-a3 (checked against autograd in the tests).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------
# label down-sampling                                                         utils/utils.py:264-268
# ----------------------------------------------------------------------------------------------
def bilinear_source(in_size: int, out_size: int):
    """Source indices and weights of a bilinear resize with ``align_corners=False`` in float32:
    ``src = scale*(dst+0.5)-0.5`` clamped at 0, ``i0 = floor(src)``, ``i1 = i0+1`` clamped,
    ``l1 = src-i0``, ``l0 = 1-l1`` with ``scale = float(in)/out``."""
    f32 = np.float32
    scale = f32(in_size) / f32(out_size)
    src = (scale * (np.arange(out_size, dtype=f32) + f32(0.5))).astype(f32) - f32(0.5)
    src = np.maximum(src, f32(0)).astype(f32)
    i0 = np.minimum(np.floor(src).astype(np.int64), in_size - 1)
    i1 = i0 + (i0 < in_size - 1)
    l1 = np.clip((src - i0.astype(f32)).astype(f32), f32(0), f32(1)).astype(f32)
    l0 = (f32(1) - l1).astype(f32)
    return i0, i1, l0, l1


def bilinear_labels_formula(labels: torch.Tensor, h: int, w: int) -> np.ndarray:
    """float32 [B, h, w]: the exact arithmetic of ``F.interpolate(labels.float()[:, None], (h, w),
    'bilinear', align_corners=False)`` as this torch build executes it for a one-channel map
    (product weights ``w_ab = lh_a*lw_b``; ``acc = v01*w01; acc = fma(v00, w00, acc);
    acc = fma(v10, w10, acc); acc = fma(v11, w11, acc)``).  This is the formula the HIP prep kernel
    implements; tests pin it to ``F.interpolate`` element for element."""
    f32, f64 = np.float32, np.float64
    x = labels.numpy().astype(f32)
    B, H, W = x.shape
    h0, h1, lh0, lh1 = bilinear_source(H, h)
    w0, w1, lw0, lw1 = bilinear_source(W, w)

    def fma(a, b, c):  # a*b is exact in float64 for float32 operands
        return (a.astype(f64) * b.astype(f64) + c.astype(f64)).astype(f32)

    w00 = (lh0[:, None] * lw0[None]).astype(f32)
    w01 = (lh0[:, None] * lw1[None]).astype(f32)
    w10 = (lh1[:, None] * lw0[None]).astype(f32)
    w11 = (lh1[:, None] * lw1[None]).astype(f32)
    out = np.empty((B, h, w), f32)
    for b in range(B):
        v00, v01 = x[b][h0][:, w0], x[b][h0][:, w1]
        v10, v11 = x[b][h1][:, w0], x[b][h1][:, w1]
        acc = (v01 * w01).astype(f32)
        acc = fma(v00, w00, acc)
        acc = fma(v10, w10, acc)
        out[b] = fma(v11, w11, acc)
    return out


def downsample_labels(labels: torch.Tensor, h: int, w: int, max_label: int = 20) -> torch.Tensor:
    """int64 [B, h, w].  Reference: bilinear (not nearest) resize of the float label map, cast to
    int8, then ``<0 -> 0`` and ``>20 -> 0`` (utils/utils.py:264,267-268).  The int8 cast truncates
    toward zero and wraps 128..255 to negatives, so after the two clamps the result is simply
    ``trunc(v)`` when that lies in [0, max_label] and 0 otherwise; ``max_label`` generalises the
    hard-coded VOC bound 20 to datasets with more classes (deviation 4)."""
    v = F.interpolate(labels.to(torch.float32).unsqueeze(1), size=(h, w), mode="bilinear",
                      align_corners=False)[:, 0]
    t = torch.trunc(v).to(torch.int64)
    return torch.where((t >= 0) & (t <= max_label), t, torch.zeros_like(t))


# ----------------------------------------------------------------------------------------------
# anchor / contrast construction                                             utils/utils.py:349-393
# ----------------------------------------------------------------------------------------------
def pre_contrastive_pixel(f_n, l_n, l_po, f_o, max_label: int = 20):
    """Returns a dict with the reference's 5-tuple (``a, c, la, lc, P``) plus the intermediate
    masks the HIP path is checked against.

    f_n, f_o : [B, N, h, w] student / teacher pre-logit features;  l_n : [B, H, W] int64 labels;
    l_po : [B, K, h, w] teacher low-resolution logits.
    """
    B, N, h, w = f_n.shape
    K = l_po.shape[1]
    label_n = downsample_labels(l_n, h, w, max_label).reshape(B * h * w)          # :264-268
    new = label_n > 0                                                             # :352
    if not bool(new.any()):
        raise RuntimeError("no new-class pixel in the batch (reference: min() of an empty tensor, "
                           "utils/utils.py:353)")
    min_new = int(label_n[new].min())                                             # :353
    teacher_arg = l_po.argmax(dim=1).reshape(B * h * w)                           # :355
    mix = torch.where(label_n == 0, teacher_arg, label_n)                         # :356
    keep = mix > 0                                                                # :358
    keep_o = keep & ~new                                                          # :359

    fn = f_n.permute(0, 2, 3, 1).reshape(B * h * w, N)                            # :273-274
    fo = f_o.detach().permute(0, 2, 3, 1).reshape(B * h * w, N)                   # :361-362
    a = F.normalize(fn[keep], dim=1)                                              # :363
    c = torch.cat((a, F.normalize(fo[keep_o], dim=1)), dim=0).detach()            # :364
    la = mix[keep]                                                                # :358
    lc = torch.cat((la, mix[keep_o]))                                             # :359

    p = torch.softmax(l_po.permute(0, 2, 3, 1), dim=-1).reshape(B * h * w, K)     # :367-371
    pa, pc = p[keep], torch.cat((p[keep], p[keep_o]))                             # :372-375
    P = pa @ pc.T                                                                 # :376
    gt_a = (la >= min_new)                                                        # :378-381
    gt_c = (lc >= min_new)                                                        # :383-386
    P = torch.where(gt_a[:, None] & gt_c[None, :], torch.ones_like(P), P)         # :388-391
    return {"a": a, "c": c, "la": la, "lc": lc, "P": P.detach(), "keep": keep, "keep_o": keep_o,
            "min_new": min_new, "label_ds": label_n, "mix": mix, "pa": pa, "pc": pc}


# ----------------------------------------------------------------------------------------------
# PixelConLossV2                                                               utils/loss.py:412-466
# ----------------------------------------------------------------------------------------------
def pixcon_loss(a, c, la, lc, P=None, temperature: float = 0.07, shift: bool = True):
    """Scalar loss; differentiable w.r.t. ``a`` through autograd.  Same op order as the reference.
    ``shift=False`` drops the row-max subtraction of loss.py:455-456 (which the reference applies to
    the positive logit but NOT to the negative sum - the "inconsistent stabilisation"); that variant
    with ``P=None, c=a`` is the v1 ``PixelConLoss`` of loss_new.py."""
    A = a.shape[0]
    R = (la.view(-1, 1) == lc.view(1, -1)).to(a.dtype)                            # :435
    pos = R.clone()
    pos[:, :A] -= torch.eye(A, dtype=a.dtype)                                     # :437 (self pair)
    negm = 1 - R                                                                  # :439
    S = (a @ c.T) / temperature                                                   # :445-447
    neg = (torch.exp(S) * negm).sum(dim=1, keepdim=True)                          # :449 un-shifted
    if shift:
        S = S - S.max(dim=1, keepdim=True)[0].detach()                            # :455-456
    if P is None:
        term = torch.log(torch.exp(S)) * pos - torch.log(torch.exp(S) + neg) * pos        # :458-459
    else:
        term = torch.log(torch.exp(S)) * pos * P - torch.log(torch.exp(S) + neg) * pos * P  # :461-462
    num = pos.sum(dim=1)                                                          # :464
    valid = num != 0
    return (-(term.sum(dim=1)[valid] / num[valid])).mean()                        # :465-466


def pixcon_loss_backward(a, c, la, lc, P=None, temperature: float = 0.07, dtype=torch.float64,
                         shift: bool = True):
    """Closed-form ``d loss / d a`` (no autograd), evaluated in ``dtype``:
    with ``w_ij = pos_ij P_ij / (num_i * R)`` (R = number of rows with a positive),
    ``D_ij = exp(S'_ij) + neg_i`` and ``G_i = sum_j w_ij / D_ij``:
        dL/dS_ij = -w_ij * neg_i / D_ij            (positive pairs)
                 +  exp(S_ij) * negm_ij * G_i      (negative pairs)
    (the row max is detached, so the shift contributes no gradient)
        dL/da    = (dL/dS) @ c / T
    Returns (loss, dL/da, row vectors neg, G, num)."""
    a, c = a.detach().to(dtype), c.detach().to(dtype)
    A = a.shape[0]
    R = (la.view(-1, 1) == lc.view(1, -1)).to(dtype)
    pos = R.clone()
    pos[:, :A] -= torch.eye(A, dtype=dtype)
    negm = 1 - R
    Pm = torch.ones_like(R) if P is None else P.to(dtype)
    S = (a @ c.T) / temperature
    E = torch.exp(S)
    neg = (E * negm).sum(dim=1, keepdim=True)
    m = S.max(dim=1, keepdim=True)[0] if shift else torch.zeros_like(neg)
    Sp = S - m
    D = torch.exp(Sp) + neg
    num = pos.sum(dim=1, keepdim=True)
    valid = (num != 0).to(dtype)
    nrows = valid.sum()
    w = pos * Pm / torch.clamp(num, min=1) / nrows * valid
    loss = -(w * (Sp - torch.log(D))).sum()
    G = (w / D).sum(dim=1, keepdim=True)
    dS = -w * neg / D + E * negm * G
    da = dS @ c / temperature
    return loss, da, neg[:, 0], G[:, 0], num[:, 0]


# ----------------------------------------------------------------------------------------------
# dead-file v1 losses (named by north_star)                              utils/loss_new.py:264-400
# ----------------------------------------------------------------------------------------------
def pixcon_loss_v1(features, labels, temperature: float = 1.0):
    """``PixelConLoss.forward`` (loss_new.py:359-400): one feature set (contrast == anchors), no
    joint-probability weight, no max shift; ``features`` [n, 1, d], ``labels`` [n].  Quirk kept:
    the negative sum is ``repeat``-ed along rows (loss_new.py:396-397), so element (i, j) is paired
    with the negatives of pixel j.  Because positives share a class (``num_i == num_j``) and ``S`` is
    symmetric, the mean equals the V2 form with ``P = 1`` and ``c = a`` in exact arithmetic - the
    special case the HIP kernel is checked on."""
    f = torch.cat(torch.unbind(features.reshape(features.shape[0], features.shape[1], -1), dim=1), dim=0)
    n = f.shape[0]
    R = (labels.view(1, -1) == labels.view(-1, 1)).to(f.dtype)
    pos = R - torch.eye(n, dtype=f.dtype)
    negm = 1 - R
    S = (f @ f.T) / temperature
    neg = (torch.exp(S) * negm).sum(dim=1)
    term = torch.log(torch.exp(S)) * pos - torch.log(torch.exp(S) + neg.repeat(n, 1)) * pos
    num = pos.sum(dim=1)
    valid = num != 0
    return (-(term.sum(dim=1)[valid] / num[valid])).mean()


def supcon_loss(features, labels=None, mask=None, temperature: float = 0.07,
                contrast_mode: str = "all", base_temperature: float = 0.07):
    """``SupConLoss.forward`` (loss_new.py:274-350): supervised contrastive loss over
    ``features`` [bsz, n_views, d]."""
    bsz, n_views = features.shape[0], features.shape[1]
    features = features.reshape(bsz, n_views, -1)
    if labels is not None and mask is not None:
        raise ValueError("Cannot define both `labels` and `mask`")
    if labels is None and mask is None:
        mask = torch.eye(bsz, dtype=torch.float32)
    elif labels is not None:
        labels = labels.contiguous().view(-1, 1)
        if labels.shape[0] != bsz:
            raise ValueError("Num of labels does not match num of features")
        mask = (labels == labels.T).float()
    else:
        mask = mask.float()
    contrast = torch.cat(torch.unbind(features, dim=1), dim=0)
    if contrast_mode == "one":
        anchor, anchor_count = features[:, 0], 1
    elif contrast_mode == "all":
        anchor, anchor_count = contrast, n_views
    else:
        raise ValueError(f"Unknown mode: {contrast_mode}")
    logits = (anchor @ contrast.T) / temperature
    logits = logits - logits.max(dim=1, keepdim=True)[0].detach()
    mask = mask.repeat(anchor_count, n_views)
    logits_mask = torch.ones_like(mask)
    logits_mask[torch.arange(bsz * anchor_count), torch.arange(bsz * anchor_count)] = 0
    mask = mask * logits_mask
    exp_logits = torch.exp(logits) * logits_mask
    log_prob = logits - torch.log(exp_logits.sum(1, keepdim=True) + 1e-6)
    mean_log_prob_pos = (mask * log_prob).sum(1) / (mask.sum(1) + 1e-8)
    loss = -(temperature / base_temperature) * mean_log_prob_pos
    return loss.view(anchor_count, bsz).mean()
