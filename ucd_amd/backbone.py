"""ResNet body of the segmentation network (ResNet-101 at output stride 16 on the hot path).

Mirror of the reference's ``models/resnet.py:11-136``: ``mod1`` = 7x7/2 conv + ABN + 3x3/2 max
pool, ``mod2..mod5`` = stacks of :class:`~ucd_amd.blocks.ResidualBlock`; the output stride is
reached by trading the stride of the last one (OS16) or two (OS8) stages for dilation
(``resnet.py:48-51,97-101``).  Module names (``mod1.conv1``, ``mod3.block2.convs.bn1`` ...)
are the reference's, so its pretrained files and step checkpoints load unchanged.
"""
from __future__ import annotations

import sys
from collections import OrderedDict
from functools import partial

import torch
import torch.nn as nn

from . import switches as _switches
from .blocks import Conv2d, ResidualBlock, try_index

_STAGE_DILATION = {16: (1, 1, 1, 2), 8: (1, 1, 2, 4)}


class GlobalAvgPool2d(nn.Module):
    """[B, C, H, W] -> [B, C] (reference ``modules/misc.py:4-12``); classifier variant only."""

    def forward(self, inputs):
        return inputs.flatten(2).mean(dim=2)


class ResNet(nn.Module):
    def __init__(self, structure, bottleneck, norm_act=nn.BatchNorm2d, classes=0,
                 output_stride=16, keep_outputs=False):
        super().__init__()
        if len(structure) != 4:
            raise ValueError("Expected a structure with four values")
        if output_stride not in _STAGE_DILATION:
            raise ValueError("Output stride must be 8 or 16")
        self.structure, self.bottleneck, self.keep_outputs = structure, bottleneck, keep_outputs
        self.dilation = dilation = list(_STAGE_DILATION[output_stride])

        stem = [("conv1", Conv2d(3, 64, 7, stride=2, padding=3, bias=False)), ("bn1", norm_act(64))]
        if try_index(dilation, 0) == 1:
            stem.append(("pool1", nn.MaxPool2d(3, stride=2, padding=1)))
        self.mod1 = nn.Sequential(OrderedDict(stem))

        width = (64, 64, 256) if bottleneck else (64, 64)
        cin = 64
        for stage, depth in enumerate(structure):
            d = try_index(dilation, stage)
            blocks = OrderedDict()
            for b in range(depth):
                # a stage downsamples in its first block unless it is dilated (or is the first)
                stride = 2 if (d == 1 and b == 0 and stage > 0) else 1
                blocks[f"block{b + 1}"] = ResidualBlock(cin, width, norm_act=norm_act, stride=stride, dilation=d)
                cin = width[-1]
            self.add_module(f"mod{stage + 2}", nn.Sequential(blocks))
            width = tuple(2 * c for c in width)
        self.out_channels = cin

        if classes != 0:
            self.classifier = nn.Sequential(OrderedDict(
                [("avg_pool", GlobalAvgPool2d()), ("fc", nn.Linear(cin, classes))]))

    def _stem(self, x):
        """mod1 = conv1 -> norm_act -> 3x3/2 max pool (models/resnet.py:58-64).  With the HIP ABN the norm's apply pass and the
        pooling are one kernel and their backward two (ucd_amd.abn.stem_norm_pool, csrc/stem.hip); otherwise the modules."""
        m = self.mod1
        if hasattr(m, "pool1") and getattr(m.bn1, "ucd_fused_abn", False) and x.is_cuda:
            pool = m.pool1
            if (pool.kernel_size, pool.stride, pool.padding, pool.dilation, pool.ceil_mode) == (3, 2, 1, 1, False):
                from .abn import stem_norm_pool
                y = _stem_eval_fused(m.conv1, m.bn1, x)
                if y is not None:
                    return y
                z = _stem_conv(m.conv1, x)
                y = stem_norm_pool(m.bn1, z)
                return y if y is not None else pool(m.bn1(z))
        return m(x)

    def forward(self, x):
        outs = [self._stem(x)]
        for name in ("mod2", "mod3", "mod4", "mod5"):
            outs.append(getattr(self, name)(outs[-1]))
        if hasattr(self, "classifier"):
            outs.append(self.classifier(outs[-1]))
        return outs if self.keep_outputs else outs[-1]


class _StemConvFunction(torch.autograd.Function):
    """conv1 of the stem on the own kernel (csrc/stem.hip::stem_conv7x7_kernel: the fp32 image is converted while it is staged,
    no cast / layout kernels); the weight gradient stays with the library (the image needs no gradient)."""

    @staticmethod
    def forward(ctx, x, w):
        from . import hip
        ctx.save_for_backward(x, w)
        return hip.stem_conv7x7(x, w)

    @staticmethod
    def backward(ctx, dz):
        x, w = ctx.saved_tensors
        dw = None
        if ctx.needs_input_grad[1]:
            xb = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            dw = torch.ops.aten.convolution_backward(dz.contiguous(memory_format=torch.channels_last), xb, w, None, [2, 2], [3, 3],
                                                     [1, 1], False, [0, 0], 1, [False, True, False])[1]
        return None, dw


def _stem_conv(conv, x):
    """``conv(x)`` of the stem: the own 7x7 / 2 kernel for an fp32 image under bf16 autocast (what bench.py / run.py
    --opt_level O1 build; working weight or autocast's per-call cast; ``UCD_OWN_STEM=0`` keeps MIOpen), else the module."""
    w = conv.working_weight() if hasattr(conv, "working_weight") else None
    if w is None and x.is_cuda and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16:
        w = conv.weight.to(torch.bfloat16)              # autocast's per-call cast (the mode without working weights)
    if (w is None or not x.is_cuda or x.dim() != 4 or x.dtype != torch.float32 or x.requires_grad or conv.bias is not None
            or tuple(conv.weight.shape) != (64, 3, 7, 7) or conv.stride != (2, 2) or conv.padding != (3, 3)
            or conv.dilation != (1, 1) or w.dtype != torch.bfloat16 or not w.is_contiguous(memory_format=torch.channels_last)
            or _switches.get("UCD_OWN_STEM", "1") == "0"):
        return conv(x)
    if torch.is_grad_enabled() and w.requires_grad:
        return _StemConvFunction.apply(x, w)
    from . import hip
    return hip.stem_conv7x7(x, w)


def _stem_eval_fused(conv, bn, x):
    """conv1 -> bn1 (frozen statistics) -> pool of the stem as ONE kernel when no gradient is wanted (the teacher, validation):
    csrc/stem.hip::stem_conv_pool_kernel, bit-identical to the two-kernel path without the 203 MB convolution output in memory.
    None when the layers / the input are not what it takes (``UCD_STEM_EVAL_FUSED=0`` switches it off)."""
    from . import abn as _abn
    if (torch.is_grad_enabled() or bn.training or _switches.get("UCD_STEM_EVAL_FUSED", "1") == "0" or _switches.get("UCD_OWN_STEM", "1") == "0"
            or _switches.get("UCD_STEM_FOLD", "1") == "0" or not getattr(bn, "ucd_fused_abn", False)
            or bn.activation not in ("leaky_relu", "identity")):             # (|gamma| + eps is part of the cached scale)
        return None
    w = conv.working_weight() if hasattr(conv, "working_weight") else None
    if w is None and x.is_cuda and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16:
        w = conv.weight.to(torch.bfloat16)
    if (w is None or not x.is_cuda or x.dim() != 4 or x.dtype != torch.float32 or conv.bias is not None
            or tuple(conv.weight.shape) != (64, 3, 7, 7) or conv.stride != (2, 2) or conv.padding != (3, 3) or conv.dilation != (1, 1)
            or w.dtype != torch.bfloat16 or not w.is_contiguous(memory_format=torch.channels_last) or x.shape[2] < 8 or x.shape[3] < 8):
        return None
    from . import hip
    invstd, scale = bn._eval_constants()
    return hip.stem_conv_pool(x, w, bn.running_mean, scale, bn.bias, _abn._act_code(bn.activation), bn.activation_param)


_NETS = {
    "18": ([2, 2, 2, 2], False), "34": ([3, 4, 6, 3], False), "50": ([3, 4, 6, 3], True),
    "101": ([3, 4, 23, 3], True), "152": ([3, 8, 36, 3], True),
}
__all__ = ["ResNet"]
for _n, (_s, _b) in _NETS.items():
    setattr(sys.modules[__name__], f"net_resnet{_n}", partial(ResNet, structure=_s, bottleneck=_b))
    __all__.append(f"net_resnet{_n}")
