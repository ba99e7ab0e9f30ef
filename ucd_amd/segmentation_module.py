"""``make_model`` and ``IncrementalSegmentationModule``: the student / teacher network of the UCD step.

Mirror of the reference's ``segmentation_module.py:14-143`` (same call signatures, module names and
``state_dict`` keys), built on this package's HIP ABN layers and channels-last activations:

* ``forward(x, x_b_old=None, x_pl_old=None, ret_intermediate=False) -> (logits [B, Ctot, H, W],
  {"body", "pre_logits", "sem"})`` as at ``segmentation_module.py:125-136``;
* the per-step 1x1 classifiers (``cls``; ``cls[0]`` frozen, ``:72-78``) run as ONE 1x1 convolution over
  the concatenated weights instead of a conv per step + ``cat`` (``:102-105``);
* the attention-weighted feature maps of ``att_map`` (``:86-94``) are produced lazily: the UCD loss
  normalises every pixel's feature vector, which cancels the positive per-pixel attention factor, so the
  contrastive kernels read the raw maps (``features.raw(...)``) and the 50+ MB attention passes only run
  if somebody indexes ``features["body"]`` / ``features["pre_logits"]`` (the ``--loss_de`` path).
"""
from __future__ import annotations

import os
from functools import partial, reduce

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import backbone as models
from . import hip
from . import switches as _switches
from .abn import ABN, InPlaceABN, InPlaceABNSync
from .blocks import DeeplabV3

_NORMS = {"iabn_sync": InPlaceABNSync, "iabn": InPlaceABN, "abn": ABN}


class _HeadProduct(torch.autograd.Function):
    """All classifier heads as ONE product on the own GEMM kernels: ``logits[M, Ct] = x[M, C] . w[Ct, C]^T + b``.

    The heads are off the 64-channel grid of the kernels (16 + 5 classes at VOC 15-5, 101 + 50 at ADE), so the library served
    them (a CK convolution forward, ``convolution_backward`` + a strided ``sum`` for the bias gradient: 0.3 ms per step).  Here the
    weight is zero-padded to 64 rows: forward = ``ucd_conv1x1`` out_mode 1 with mean 0, scale 1, shift = bias (``(v - 0) * 1 + b``:
    exactly the bias add in fp32 before the one rounding to bf16) into an [M, 64] buffer whose first Ct columns are the logits;
    backward = the input gradient as a K = 64 product on the transposed padded weight, the weight gradient through
    ``ucd_conv_wgrad`` straight into fp32, the bias gradient as per-image column sums (``ucd_plane_sum``).  bf16 GPU tensors with
    C a multiple of 64 and Ct <= 64 only (``UCD_OWN_HEADS=0`` keeps the library); everything else goes to ``F.conv2d``."""

    @staticmethod
    def forward(ctx, x, w, b):
        B, C, h, wd = x.shape
        Ct = w.shape[0]
        M = B * h * wd
        dev = x.device
        rows = x.permute(0, 2, 3, 1).reshape(M, C)                    # a view of the channels-last map
        wp = torch.zeros(64, C, dtype=torch.bfloat16, device=dev)
        wp[:Ct] = w.detach().reshape(Ct, C)
        shift = torch.zeros(64, dtype=torch.float32, device=dev)
        if b is not None:
            shift[:Ct] = b.detach()
        zero, one = torch.zeros(64, dtype=torch.float32, device=dev), torch.ones(64, dtype=torch.float32, device=dev)
        y = torch.empty(M, 64, dtype=torch.bfloat16, device=dev)
        hip.conv1x1(rows, wp, y, out_mode=1, out_norm=(zero, one, shift, None, hip.ACT_IDENTITY, 1.0))
        ctx.save_for_backward(rows, wp)
        ctx.meta = (B, C, h, wd, Ct, w.dtype, None if b is None else b.dtype)
        # dense channels-last [B, Ct, h, w] like the convolution's own output
        return y[:, :Ct].contiguous().view(B, h, wd, Ct).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        rows, wp = ctx.saved_tensors
        B, C, h, wd, Ct, wdt, bdt = ctx.meta
        M = B * h * wd
        dev = g.device
        gp = torch.zeros(M, 64, dtype=torch.bfloat16, device=dev)
        gp[:, :Ct] = g.permute(0, 2, 3, 1).reshape(M, Ct)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dxr = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
            hip.conv1x1(gp, wp.t().contiguous(), dxr)                  # [M, 64] . [C, 64]^T
            dx = dxr.view(B, h, wd, C).permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            dw32 = torch.empty(64, C, dtype=torch.float32, device=dev)
            hip.conv_wgrad(gp, rows, dw32=dw32)
            dw = dw32[:Ct].reshape(Ct, C, 1, 1).to(wdt)
        if bdt is not None and ctx.needs_input_grad[2]:
            per_image = torch.empty(B, 64, dtype=torch.float32, device=dev)
            hip.plane_sum(gp, 64, B, h * wd, 64, 1.0, per_image)
            db = per_image.sum(0)[:Ct].to(bdt)
        return dx, dw, db


def _own_heads_ok(x, w):
    return (x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and x.shape[1] % 64 == 0 and w.shape[0] <= 64
            and x.is_contiguous(memory_format=torch.channels_last) and _switches.get("UCD_OWN_HEADS", "1") != "0"
            and _switches.get("UCD_FUSED_CONV1X1", "1") != "0")


def make_model(opts, classes=None):
    """Reference ``make_model`` (segmentation_module.py:14-53).  ``--norm_act std`` (plain BatchNorm2d)
    is rejected: the reference itself cannot run it (modules/deeplab.py:39 reads ``.activation``).
    Unlike the reference, ``--no_pretrained`` works (there the model is only built inside
    ``if not opts.no_pretrained``)."""
    if opts.norm_act not in _NORMS:
        raise ValueError(f"--norm_act {opts.norm_act!r}: use one of {sorted(_NORMS)}")
    norm = partial(_NORMS[opts.norm_act], activation="leaky_relu", activation_param=.01)
    body = getattr(models, f"net_{opts.backbone}")(norm_act=norm, output_stride=opts.output_stride)
    if not opts.no_pretrained:
        path = f"pretrained/{opts.backbone}_{opts.norm_act}.pth.tar"
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} (pass --no_pretrained to start from random weights)")
        state = torch.load(path, map_location="cpu")["state_dict"]
        state = {k[len("module."):] if k.startswith("module.") else k: v for k, v in state.items()
                 if "classifier.fc." not in k}
        body.load_state_dict(state)
    head_channels = 256
    head = DeeplabV3(body.out_channels, head_channels, 256, norm_act=norm, out_stride=opts.output_stride,
                     pooling_size=opts.pooling)
    if classes is None:
        classes = [opts.num_classes]
    return IncrementalSegmentationModule(body, head, head_channels, classes=classes,
                                         fusion_mode=getattr(opts, "fusion_mode", "mean"))


def att_map(x):
    """``a = sum_c x^2`` normalised per image by its Frobenius norm, detached, times ``x``
    (segmentation_module.py:86-94).  HIP kernel when no gradient is required, differentiable torch
    composition otherwise."""
    if x.is_cuda and not (torch.is_grad_enabled() and x.requires_grad):
        xv, M, Cc, HW, ld = hip.rows_view(x)
        y = hip.empty_like_rows(xv)
        hip.attmap(xv, ld, y, Cc, x.shape[0], HW, Cc)
        return y
    a = (x.float() ** 2).sum(dim=1)
    a = a / a.flatten(1).norm(dim=1)[:, None, None]
    return (a.unsqueeze(1).detach() * x).to(x.dtype)


class Features(dict):
    """The reference's ``{"body", "pre_logits", "sem"}`` dict; the two attention-weighted maps are
    computed on first access, ``raw(name)`` returns the un-weighted map."""

    def __init__(self, x_b, x_pl, sem):
        super().__init__(sem=sem)
        self._raw = {"body": x_b, "pre_logits": x_pl}

    def raw(self, name):
        return self._raw[name]

    def __missing__(self, key):
        if key in self._raw:
            self[key] = att_map(self._raw[key])
            return dict.__getitem__(self, key)
        raise KeyError(key)

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._raw

    def keys(self):
        return ["body", "pre_logits", "sem"]


class IncrementalSegmentationModule(nn.Module):
    def __init__(self, body, head, head_channels, classes, ncm=False, fusion_mode="mean"):
        super().__init__()
        assert isinstance(classes, list), \
            "Classes must be a list where to every index correspond the num of classes for that task"
        self.body, self.head = body, head
        self.cls = nn.ModuleList([nn.Conv2d(head_channels, c, 1) for c in classes])
        self.cls[0].weight.requires_grad = False       # the first head is never trained (:77-78)
        self.cls[0].bias.requires_grad = False
        self.classes, self.head_channels = classes, head_channels
        self.tot_classes = reduce(lambda a, b: a + b, classes)

    def _network(self, x, x_b_old=None, x_pl_old=None, ret_intermediate=False):
        x_b = self.body(x)
        x_pl = self.head(x_b)
        if len(self.cls) == 1:
            c0 = self.cls[0]
            x_o = _HeadProduct.apply(x_pl, c0.weight, c0.bias) if _own_heads_ok(x_pl, c0.weight) else c0(x_pl)
        else:  # all heads in one 1x1 convolution
            w = torch.cat([m.weight for m in self.cls], dim=0)
            b = torch.cat([m.bias for m in self.cls], dim=0)
            x_o = _HeadProduct.apply(x_pl, w, b) if _own_heads_ok(x_pl, w) else F.conv2d(x_pl, w, b)
        return x_o, x_b, x_pl

    def init_new_classifier(self, device):
        """Balanced initialisation (segmentation_module.py:111-123): the newest head starts from the
        background weight row with bias ``b_bkg - log(n_new + 1)``; head 0's background bias is
        overwritten with the same value."""
        cls = self.cls[-1]
        with torch.no_grad():
            new_bias = self.cls[0].bias[0] - torch.log(torch.tensor([self.classes[-1] + 1.0], device=device))[0]
            cls.weight.copy_(self.cls[0].weight[0].expand_as(cls.weight))
            cls.bias.fill_(new_bias.item())
            self.cls[0].bias[0] = new_bias.item()

    def forward(self, x, x_b_old=None, x_pl_old=None, ret_intermediate=False, upsample=True):
        """``upsample=False`` (not in the reference) skips the x16 bilinear up-sampling and returns ``None``
        for the full-resolution logits: the trainer's fused losses read ``features["sem"]`` instead."""
        out_size = x.shape[-2:]
        if x.is_cuda and x.dim() == 4:
            x = x.contiguous(memory_format=torch.channels_last)
        if x.is_cuda and self.training and torch.is_grad_enabled():
            # a training forward begins a step: the statistics accumulators of the conv + ABN nodes (csrc/abn_node.cpp: one zeroed
            # arena per device, a slot per layer and direction) are cleared with ONE fill and handed out again
            from . import abn as _abn
            node = _abn._abn_node()
            if node is not None and hasattr(node, "stat_arena_reset"):
                node.stat_arena_reset(x.device.index if x.device.index is not None else torch.cuda.current_device(), hip.stream())
        sem, x_b, x_pl = self._network(x, x_b_old, x_pl_old, ret_intermediate)
        logits = F.interpolate(sem, size=out_size, mode="bilinear", align_corners=False) if upsample else None
        return logits, Features(x_b, x_pl, sem)

    def fix_bn(self):
        for m in self.modules():
            if isinstance(m, (nn.BatchNorm2d, ABN)):
                m.eval()
                m.weight.requires_grad = False
                m.bias.requires_grad = False
