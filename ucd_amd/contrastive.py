"""Uncertainty-weighted pixel-contrastive distillation on the HIP kernels.

Host-side mirror of the reference's wired contrastive path:
  * ``pre_contractive_pixel(f_n, l_n, l_po, f_o)``          utils/utils.py:256-393 (twin utils/loss.py:258-395)
  * ``PixelConLossV2(temperature).forward(a, c, la, lc, P)``  utils/loss.py:403-466
and of the step the reference's trainer composes from them (train.py:115-116, as intended - SURVEY.md
section 0): :func:`ucd_contrastive_loss` fuses prep + loss + gradient and never builds the [A, C]
matrices; it is what :class:`ucd_amd.train.Trainer` calls.  The two reference-shaped entry points are
kept for drop-in use and for the parity tests: ``pre_contractive_pixel`` returns the reference's
5-tuple (``P`` materialised on request only), and ``PixelConLossV2`` accepts that tuple.

All sizes that depend on the data (anchor count A, contrast count C) stay on the device; the host only
allocates worst-case buffers (A <= BHW, C <= 2 BHW) - there is no ``.cpu()`` round trip
(the reference syncs at utils/utils.py:356).
"""
from __future__ import annotations

import weakref

import torch
import torch.nn as nn

from . import hip


class PixconBatch:
    """Device-resident anchor / contrast sets of one batch (outputs of the prep + gather kernels)."""

    __slots__ = ("BHW", "N", "K", "anchor_pix", "old_pix", "row_label", "prob", "meta", "chat", "pcat",
                 "inv_norm", "ldp", "sorted", "f_dtype", "ch16", "p16")

    def meta_host(self):
        """Copy the meta record to the host (tests / logging only: this synchronises)."""
        raw = self.meta.cpu().numpy().tobytes()
        return hip.PixconMeta.from_buffer_copy(raw)


def _rows2d(t, what):
    """[B, C, h, w] map -> ([B*h*w, C] row view sharing memory, ld)."""
    t2, M, Cc, HW, ld = hip.rows_view(t)
    return t2, M, Cc, ld


def pixcon_prepare(f_n, labels, l_po, f_o, max_label=20, sort_by_label=True, fp16=False):
    """Run ucd_pixcon_prep + ucd_pixcon_gather.  ``f_n``/``f_o``: [B, N, h, w] student / teacher
    pre-logits (any strides; channels-last is zero-copy), ``labels``: [B, H, W] int64,
    ``l_po``: [B, K, h, w] teacher low-resolution logits."""
    lib = hip.load()
    if not f_n.is_cuda:
        raise RuntimeError("ucd_amd.contrastive runs on the GPU only (there is no CPU fallback)")
    B, N, h, w = f_n.shape
    K = l_po.shape[1]
    H, W = labels.shape[-2:]
    if N > hip.PIXCON_LD:
        raise RuntimeError(f"feature dimension {N} > {hip.PIXCON_LD} is not supported")
    dev = f_n.device
    BHW = B * h * w
    if f_o.dtype != f_n.dtype:
        f_o = f_o.to(f_n.dtype)
    fn2, _, _, ld_n = _rows2d(f_n, "f_n")
    fo2, _, _, ld_o = _rows2d(f_o.detach(), "f_o")
    # the teacher logits row matrix [BHW, K]; K is small, so a packed copy is cheap
    t = l_po.detach().permute(0, 2, 3, 1).reshape(BHW, K)
    if t.dtype not in (torch.float32, torch.bfloat16):
        t = t.float()
    t = t.contiguous()
    labels = labels.contiguous()
    if labels.dtype != torch.int64:
        labels = labels.long()

    pb = PixconBatch()
    pb.BHW, pb.N, pb.K, pb.sorted, pb.f_dtype = BHW, N, K, bool(sort_by_label), f_n.dtype
    rows = 2 * BHW + 2 * hip.PIX_TILE
    pb.anchor_pix = torch.empty(BHW, dtype=torch.int32, device=dev)
    pb.old_pix = torch.empty(BHW, dtype=torch.int32, device=dev)
    pb.row_label = torch.empty(rows, dtype=torch.uint8, device=dev)
    pb.prob = torch.empty(BHW, K, dtype=torch.float32, device=dev)
    pb.meta = torch.empty(hip.META_BYTES, dtype=torch.uint8, device=dev)
    nbytes = lib.ucd_pixcon_prep_workspace_bytes(BHW, K)
    ws = hip.workspace(nbytes, dev, "pixprep")
    with hip._timed("ucd_pixcon_prep", BHW * (4 * 8 + K * t.element_size())):
        hip._check(lib.ucd_pixcon_prep(hip.ptr(labels), B, H, W, h, w, int(max_label), hip.ptr(t), K, hip.dtype_code(t),
                                       K, int(bool(sort_by_label)), hip.ptr(pb.anchor_pix), hip.ptr(pb.old_pix),
                                       hip.ptr(pb.row_label), hip.ptr(pb.prob), hip.ptr(pb.meta), hip.ptr(ws), nbytes,
                                       hip.stream()), "ucd_pixcon_prep")
    pb.ldp = (K + 1) & ~1
    pb.chat = torch.empty(rows, hip.PIXCON_LD, dtype=torch.float32, device=dev)
    pb.pcat = torch.empty(rows, pb.ldp, dtype=torch.float32, device=dev)
    pb.inv_norm = torch.empty(BHW, dtype=torch.float32, device=dev)
    pb.ch16 = pb.p16 = None
    if fp16:
        pb.ch16 = torch.empty(rows, hip.PIXCON_LD, dtype=torch.float16, device=dev)
        pb.p16 = torch.empty(rows, 2 * ((K + 15) // 16 * 16), dtype=torch.float16, device=dev)
    with hip._timed("ucd_pixcon_gather", 2 * BHW * N * (fn2.element_size() + 4)):
        hip._check(lib.ucd_pixcon_gather(hip.ptr(fn2), ld_n, hip.ptr(fo2), ld_o, hip.dtype_code(fn2), BHW, N,
                                         hip.ptr(pb.anchor_pix), hip.ptr(pb.old_pix), hip.ptr(pb.prob), K,
                                         hip.ptr(pb.meta), hip.ptr(pb.chat), hip.PIXCON_LD, hip.ptr(pb.pcat), pb.ldp,
                                         hip.ptr(pb.ch16), hip.ptr(pb.p16), hip.ptr(pb.inv_norm), hip.stream()),
                   "ucd_pixcon_gather")
    return pb


def pixcon_loss_raw(pb, temperature=0.07, shift_pos=True, use_prob=True, need_grad=True, row_stats=False,
                    precision="f32"):
    """ucd_pixcon_loss on a prepared batch: returns (loss_out[2], grad_a or None, row_stats or None).
    precision "f32" = exact-fp32 MFMA parity mode, "f16" = fp16-operand performance mode (needs a batch
    prepared with fp16=True)."""
    lib = hip.load()
    prec = hip.PIXCON_PRECISION[precision]
    if prec != hip.PIXCON_F32 and pb.ch16 is None:
        raise RuntimeError("precision='f16' needs pixcon_prepare(..., fp16=True)")
    dev = pb.chat.device
    loss_out = torch.empty(2, dtype=torch.float32, device=dev)
    grad_a = torch.empty(pb.BHW, hip.PIXCON_LD, dtype=torch.float32, device=dev) if need_grad else None
    stats = torch.zeros(3, pb.BHW, dtype=torch.float32, device=dev) if row_stats else None
    nbytes = lib.ucd_pixcon_loss_workspace_bytes(pb.BHW, pb.N, pb.K)
    ws = hip.workspace(nbytes, dev, "pixloss")
    work = 0
    if hip._timing is not None:     # instrumented bench pass only: algorithmic flops A*C*(4N+2K) need the counts
        m = pb.meta_host()
        work = float(m.A) * float(m.A + m.Co) * (4 * pb.N + 2 * pb.K)
    with hip._timed("ucd_pixcon_loss[%s]" % precision, work):
        hip._check(lib.ucd_pixcon_loss(hip.ptr(pb.chat), hip.PIXCON_LD, pb.N, hip.ptr(pb.row_label), hip.ptr(pb.pcat),
                                       pb.ldp, pb.K, hip.ptr(pb.ch16), hip.ptr(pb.p16), prec, hip.ptr(pb.meta), pb.BHW,
                                       float(temperature), int(bool(shift_pos)), int(bool(use_prob)),
                                       hip.ptr(loss_out), hip.ptr(grad_a), hip.PIXCON_LD, hip.ptr(stats), hip.ptr(ws),
                                       nbytes, hip.stream()), "ucd_pixcon_loss")
    return loss_out, grad_a, stats


class _FusedContrastive(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f_n, labels, l_po, f_o, temperature, max_label, precision):
        pb = pixcon_prepare(f_n, labels, l_po, f_o, max_label=max_label, sort_by_label=True,
                            fp16=hip.PIXCON_PRECISION[precision] != hip.PIXCON_F32)
        loss_out, grad_a, _ = pixcon_loss_raw(pb, temperature, True, True, need_grad=ctx.needs_input_grad[0],
                                              precision=precision)
        ctx.pb, ctx.grad_a, ctx.shape, ctx.dtype = pb, grad_a, f_n.shape, f_n.dtype
        return loss_out[0].clone()

    @staticmethod
    def backward(ctx, g):
        lib = hip.load()
        pb, grad_a = ctx.pb, ctx.grad_a
        B, N, h, w = ctx.shape
        d = torch.empty((B, N, h, w), dtype=ctx.dtype, device=g.device, memory_format=torch.channels_last)
        gs = g.reshape(1).float().contiguous()
        hip._check(lib.ucd_pixcon_scatter_grad(hip.ptr(grad_a), hip.ptr(pb.chat), hip.PIXCON_LD, hip.ptr(pb.inv_norm),
                                               hip.ptr(pb.anchor_pix), hip.ptr(pb.meta), hip.ptr(gs), hip.ptr(d), N,
                                               hip.dtype_code(d), pb.BHW, N, hip.stream()), "ucd_pixcon_scatter_grad")
        ctx.pb = ctx.grad_a = None
        return d, None, None, None, None, None, None


def ucd_contrastive_loss(f_n, labels, l_po, f_o, temperature=0.07, max_label=20, precision="f32"):
    """``PixelConLossV2(T)(*pre_contractive_pixel(f_n, labels, l_po, f_o))`` as one fused, differentiable
    device-side operation (train.py:115-116 as intended).  ``precision``: "f32" (exact-fp32 MFMA, parity
    mode) or "f16" (fp16 operands / fp32 accumulation, performance mode)."""
    return _FusedContrastive.apply(f_n, labels, l_po, f_o, float(temperature), int(max_label), precision)


# ---------------------------------------------------------------------------------------------
# reference-shaped entry points
# ---------------------------------------------------------------------------------------------
class _NormalizedAnchors(torch.autograd.Function):
    """a = normalize(f_n rows kept as anchors); backward scatters through the normalisation."""

    @staticmethod
    def forward(ctx, f_n, pb, A):
        ctx.pb, ctx.shape, ctx.dtype, ctx.A = pb, f_n.shape, f_n.dtype, A
        return pb.chat[:A, :pb.N].clone()

    @staticmethod
    def backward(ctx, g):
        lib = hip.load()
        pb = ctx.pb
        B, N, h, w = ctx.shape
        ga = torch.zeros(pb.BHW, hip.PIXCON_LD, dtype=torch.float32, device=g.device)
        ga[:ctx.A, :N] = g
        one = torch.ones(1, dtype=torch.float32, device=g.device)
        d = torch.empty((B, N, h, w), dtype=ctx.dtype, device=g.device, memory_format=torch.channels_last)
        hip._check(lib.ucd_pixcon_scatter_grad(hip.ptr(ga), hip.ptr(pb.chat), hip.PIXCON_LD, hip.ptr(pb.inv_norm),
                                               hip.ptr(pb.anchor_pix), hip.ptr(pb.meta), hip.ptr(one), hip.ptr(d), N,
                                               hip.dtype_code(d), pb.BHW, N, hip.stream()), "ucd_pixcon_scatter_grad")
        return d, None, None


class PixconTuple(tuple):
    """The reference's 5-tuple ``(a, c, la, lc, P)`` plus the prepared batch it came from (``.batch``), so
    that :class:`PixelConLossV2` can run the fused kernel instead of re-deriving everything from ``P``."""
    batch = None


# anchors tensor -> (weak reference to it, prepared batch, the P it was returned with): lets
# ``PixelConLossV2()(a, c, la, lc, P)`` - the reference's literal call, train.py:115-116 unpacks the tuple first - find the
# device batch behind plain tensors.  Entries die with their anchors tensor.
_batches_by_anchor = {}


def _remember(a, pb, P):
    key = id(a)

    def _drop(_ref, key=key):
        _batches_by_anchor.pop(key, None)
    _batches_by_anchor[key] = (weakref.ref(a, _drop), pb, None if P is None else weakref.ref(P))


def _recall(a, P):
    """(prepared batch, use_prob) when ``a`` (and ``P``, if given) are the very tensors pre_contractive_pixel returned."""
    ent = _batches_by_anchor.get(id(a))
    if ent is None or ent[0]() is not a:
        return None
    if P is None:
        return ent[1], False
    if ent[2] is not None and ent[2]() is P:
        return ent[1], True
    return None


def pre_contractive_pixel(f_n, l_n, l_po=None, f_o=None, max_label=20, materialize_P=True):
    """Reference signature and return order (utils/utils.py:256,393): anchors ``a`` [A, N] (grad flows to
    ``f_n``), contrast ``c`` [C, N], labels ``la`` [A] / ``lc`` [C] (int8 like the reference) and the
    joint-probability weight ``P`` [A, C].  Rows are in the reference's pixel order.  This entry point
    returns host-sized tensors, so it reads the counts back (one sync) - the trainer uses
    :func:`ucd_contrastive_loss`, which does not."""
    if l_po is None or f_o is None:
        raise NotImplementedError("only the wired teacher/student form (utils/utils.py:316-393) is on the hot path")
    pb = pixcon_prepare(f_n, l_n, l_po, f_o, max_label=max_label, sort_by_label=False)
    m = pb.meta_host()
    A, Co, Apad = m.A, m.Co, m.Apad
    if m.n_new == 0:
        raise RuntimeError("no new-class pixel in the batch (the reference fails at utils/utils.py:353)")
    a = _NormalizedAnchors.apply(f_n, pb, A)
    c = torch.cat((pb.chat[:A, :pb.N], pb.chat[Apad:Apad + Co, :pb.N]), dim=0)
    la = pb.row_label[:A].to(torch.int8)
    lc = torch.cat((pb.row_label[:A], pb.row_label[Apad:Apad + Co])).to(torch.int8)
    P = None
    if materialize_P:
        pa = pb.pcat[:A, :pb.K]
        pc = torch.cat((pa, pb.pcat[Apad:Apad + Co, :pb.K]), dim=0)
        P = pa @ pc.T
        gt_a, gt_c = la >= m.min_new, lc >= m.min_new
        P = torch.where(gt_a[:, None] & gt_c[None, :], torch.ones_like(P), P)
    out = PixconTuple((a, c, la, lc, P))
    out.batch = pb
    _remember(a, pb, P)
    return out


class _LossOnBatch(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, pb, temperature, use_prob, shift_pos):
        loss_out, grad_a, _ = pixcon_loss_raw(pb, temperature, shift_pos, use_prob, need_grad=True)
        ctx.grad_a, ctx.A, ctx.N = grad_a, a.shape[0], a.shape[1]
        return loss_out[0].clone()

    @staticmethod
    def backward(ctx, g):
        return ctx.grad_a[:ctx.A, :ctx.N] * g, None, None, None, None


def _meta_from_labels(A, Co, la, lc):
    """Device-built ``ucd_pixcon_meta`` for plain tensors: the sizes are host-known shapes, the label histograms and the
    number of anchors with a positive are computed on the device (no synchronisation)."""
    dev = la.device
    tile = hip.PIX_TILE
    Apad = (A + tile - 1) // tile * tile
    Cpad = (Apad + Co + tile - 1) // tile * tile
    la64, lc64 = la.long(), lc.long()
    count_a = torch.bincount(la64, minlength=256)[:256]
    count_c = torch.bincount(lc64, minlength=256)[:256]
    n_valid = (count_c[la64] > 1).sum().reshape(1)
    head = torch.tensor([A, Co, 256, 0, Apad, Cpad], dtype=torch.int64, device=dev)
    z = torch.zeros
    meta = torch.cat((head, n_valid, z(1 + 257 + 257, dtype=torch.int64, device=dev), count_a, count_c,
                      z(2, dtype=torch.int64, device=dev))).to(torch.int32)
    assert meta.numel() * 4 == hip.META_BYTES
    return meta, Apad, Cpad


class _LossOnTensors(torch.autograd.Function):
    """PixelConLossV2 on plain ``(a, c, la, lc, P)`` tensors through ``ucd_pixcon_loss_given_p`` (exact-fp32 MFMA sweeps;
    the [A, C] score matrices of utils/loss.py:435-462 are still never built, only the caller's P is read)."""

    @staticmethod
    def forward(ctx, a, c, la, lc, P, temperature, shift_pos, one_sided):
        lib = hip.load()
        A, N = a.shape
        Ct = c.shape[0]
        if a.dim() != 2 or c.dim() != 2 or c.shape[1] != N or Ct < A or la.numel() != A or lc.numel() != Ct:
            raise RuntimeError("PixelConLossV2: expected a [A, N], c [C >= A, N] (c[:A] = the anchors), la [A], lc [C]")
        if N > hip.PIXCON_LD:
            raise RuntimeError(f"feature dimension {N} > {hip.PIXCON_LD} is not supported")
        if P is not None and tuple(P.shape) != (A, Ct):
            raise RuntimeError(f"PixelConLossV2: P must be [A, C] = [{A}, {Ct}], got {tuple(P.shape)}")
        dev = a.device
        Co = Ct - A
        meta, Apad, Cpad = _meta_from_labels(A, Co, la.reshape(-1), lc.reshape(-1))
        chat = torch.zeros(Cpad, hip.PIXCON_LD, dtype=torch.float32, device=dev)
        chat[:A, :N] = a.detach()
        chat[Apad:Apad + Co, :N] = c.detach()[A:]
        row_label = torch.full((Cpad,), 255, dtype=torch.uint8, device=dev)
        row_label[:A] = la.reshape(-1).to(torch.uint8)
        row_label[Apad:Apad + Co] = lc.reshape(-1)[A:].to(torch.uint8)
        if P is not None:
            P = P.detach().float()
            if P.stride(1) != 1:
                P = P.contiguous()
        maxA = max(A, 1)
        loss_out = torch.empty(2, dtype=torch.float32, device=dev)
        grad_a = torch.empty(maxA, hip.PIXCON_LD, dtype=torch.float32, device=dev)
        nbytes = lib.ucd_pixcon_loss_workspace_bytes(maxA, N, 0)
        ws = hip.workspace(nbytes, dev, "pixloss")
        hip._check(lib.ucd_pixcon_loss_given_p(hip.ptr(chat), hip.PIXCON_LD, N, hip.ptr(row_label), hip.ptr(P),
                                               P.stride(0) if P is not None else 0, hip.ptr(meta), maxA, float(temperature),
                                               int(bool(shift_pos)), hip.ptr(loss_out), hip.ptr(grad_a), hip.PIXCON_LD, None,
                                               hip.ptr(ws),
                                               nbytes, hip.stream()), "ucd_pixcon_loss_given_p")
        ctx.grad_a, ctx.A, ctx.N, ctx.dtype, ctx.one_sided = grad_a, A, N, a.dtype, one_sided
        return loss_out[0].clone()

    @staticmethod
    def backward(ctx, g):
        if not ctx.one_sided:
            raise NotImplementedError("PixelConLoss (utils/loss_new.py:359-400, never imported by the reference) differentiates "
                                      "through both sides of f f^T; the HIP sweeps produce the anchor-side gradient only "
                                      "(the wired PixelConLossV2 detaches the contrast side)")
        return (ctx.grad_a[:ctx.A, :ctx.N] * g).to(ctx.dtype), None, None, None, None, None, None, None


class PixelConLossV2(nn.Module):
    """``forward(anchor_features, contrast_feature, anchor_labels, contrast_labels, P=None)`` like the reference
    (utils/loss.py:412).  Three ways in, all on the HIP kernels:

    * the tuple returned by this package's :func:`pre_contractive_pixel` (or ``batch=tuple.batch``): the fused kernel on
      the prepared device batch, P formed in-tile from the teacher probabilities;
    * the five tensors of that tuple unpacked, as the reference's trainer passes them (train.py:115-116): the batch is
      found through the anchors tensor - same kernel;
    * any other plain tensors: ``ucd_pixcon_loss_given_p`` reads the given ``P`` ([A, C]); ``contrast_feature[:A]`` must be
      the anchors (the reference's construction, utils/utils.py:362).  Gradient flows to ``anchor_features`` only, as in the
      reference (``contrast_feature`` and ``P`` arrive detached)."""

    def __init__(self, sample_method="none", temperature=0.07):
        super().__init__()
        self.temperature, self.sample_method = temperature, sample_method

    def forward(self, anchor_features, contrast_feature=None, anchor_labels=None, contrast_labels=None, P=None,
                batch=None):
        if isinstance(anchor_features, PixconTuple):
            tup = anchor_features
            anchor_features, batch, P = tup[0], tup.batch, tup[4]
        if not anchor_features.is_cuda:
            raise RuntimeError("ucd_amd.contrastive runs on the GPU only (there is no CPU fallback)")
        use_prob = P is not None
        if batch is None:
            hit = _recall(anchor_features, P)
            if hit is not None:
                batch, use_prob = hit
        if batch is not None:
            return _LossOnBatch.apply(anchor_features, batch, float(self.temperature), use_prob, True)
        if contrast_feature is None or anchor_labels is None or contrast_labels is None:
            raise RuntimeError("PixelConLossV2: pass (anchor_features, contrast_feature, anchor_labels, contrast_labels[, P])")
        return _LossOnTensors.apply(anchor_features, contrast_feature, anchor_labels, contrast_labels, P,
                                    float(self.temperature), True, True)


class PixelConLoss(nn.Module):
    """The older single-set loss of the reference's dead file (utils/loss_new.py:352-400; nothing imports it):
    ``forward(features [n, 1, d], labels [n])`` = the V2 form with contrast == anchors, ``P = 1`` and no row-max shift,
    which is how it runs here (``ucd_pixcon_loss_given_p(shift_pos = 0, P = NULL)``).  Forward value only: the reference
    differentiates through both factors of ``f f^T``, the kernels through the anchor side (backward raises).
    ``SupConLoss`` of the same file (loss_new.py:264-350) has no HIP path: it is pinned in ``oracle/`` only."""

    def __init__(self, temperature=1.0):
        super().__init__()
        self.temperature = temperature

    def forward(self, features, labels):
        f = torch.cat(torch.unbind(features.reshape(features.shape[0], features.shape[1], -1), dim=1), dim=0)
        if not f.is_cuda:
            raise RuntimeError("ucd_amd.contrastive runs on the GPU only (there is no CPU fallback)")
        if features.shape[1] != 1:
            raise NotImplementedError("one view per pixel ([n, 1, d]), as loss_new.py:359-400 is written for")
        lab = labels.reshape(-1)
        return _LossOnTensors.apply(f, f, lab, lab, None, float(self.temperature), False, False)
