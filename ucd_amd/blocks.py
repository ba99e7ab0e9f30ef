"""Bottleneck residual block and DeepLab-V3 ASPP head of the UCD student / teacher.

Host-side mirror of the reference's ``modules/residual.py:7-98`` (``ResidualBlock``) and
``modules/deeplab.py:8-89`` (``DeeplabV3``): same constructor arguments, same sub-module
names (so ``state_dict`` keys are interchangeable with reference checkpoints), same
arithmetic.  What differs is where the HBM-bound glue runs.  When the ``norm_act`` modules
are this package's HIP ABN (``ucd_amd.abn``), the block epilogue
``bn3 -> + residual -> leaky_relu`` (``residual.py:84-97``) and the head's
``cat -> map_bn`` / ``out += pool -> red_bn`` (``deeplab.py:56-69``) are single fused
kernels:

* ``bn3`` normalises, adds the shortcut and applies the block activation in one pass;
* ``map_bn`` runs per 256-channel branch and writes straight into its channel slice of the
  1024-wide ``red_conv`` input, so the concatenation is never materialised;
* the image-level pooling branch enters ``red_bn`` as a per-(image, channel) bias, so the
  ``repeat`` + ``+=`` over the full map never happens.

With any other ``norm_act`` (e.g. a stock BatchNorm shim in the CPU tests) the same modules
fall back to the literal op sequence of the reference.
"""
from __future__ import annotations

import os
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F


def _wgrad_split(M: int) -> int:
    """Number of K-chunks for the weight-gradient GEMM dW = dY^T X of a 1x1 convolution over M = B*H*W rows.
    hipBLASLt's pick for the plain product is a 64x256 macro tile without split-K: 16 workgroups on 256 CUs,
    145 us for 1024x256 at M = 26136.  Splitting M in 8 batched chunks + a sum is 41 us (tools/wgrad_probe.py);
    below ~8k rows the plain product is as fast."""
    if M >= 8192:
        for S in (8, 4, 12, 6, 3, 2):
            if M % S == 0:
                return S
    return 1


from . import switches as _switches


def _env(name, default):
    """An A/B switch (ucd_amd.switches): resolved once per process, a dict lookup afterwards (the layer code asks ~600 times per
    step)."""
    return _switches.get(name, default)


def _own_wgrad() -> bool:
    """Weight gradients on the own kernel (csrc/wgrad.hip: ucd_conv_wgrad) instead of MIOpen's weight-gradient solvers / the
    batched split-M library products; ``UCD_OWN_WGRAD=0`` restores those (the A/B reference of the tests and probes)."""
    return _env("UCD_OWN_WGRAD", "1") != "0"


def _lib_gemm():
    """``ucd_amd.hip`` when its hipBLASLt entry point (tuned once per shape, ~12 us of host time per call instead of
    ~19 through the framework) can be used, else None (then the same products go through torch)."""
    from . import hip
    return hip if hip.gemm_available() else None


_node_cache = [False]


def _gemm_node():
    """ucd_amd._abn_node (C++ autograd nodes) when built and the tuned-GEMM entry point is usable, else None."""
    if _node_cache[0] is False:
        from . import abn
        mod = abn._abn_node()
        _node_cache[0] = mod if (mod is not None and hasattr(mod, "gemm1x1") and _lib_gemm() is not None) else None
    return _node_cache[0]


def _hip_stream():
    from . import hip
    return hip.stream()


def _mm_nt(rows, w):
    """rows[M, K] @ w[N, K]^T (bf16)."""
    lib = _lib_gemm() if rows.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 else None
    if lib is None:
        return rows @ w.t()
    return lib.gemm_bf16(0, rows, w, torch.empty(rows.shape[0], w.shape[0], dtype=rows.dtype, device=rows.device))


class _Gemm1x1(torch.autograd.Function):
    """rows[M, Ci] x w[Co, Ci, 1, 1]^T with the weight gradient computed as a split-K batched GEMM and returned in
    the weight's own 4-D layout (so autograd adopts it without a re-striding copy)."""

    @staticmethod
    def forward(ctx, rows, w4):
        ctx.save_for_backward(rows, w4)
        return _mm_nt(rows, w4.reshape(w4.shape[0], w4.shape[1]))

    @staticmethod
    def backward(ctx, dy):
        rows, w4 = ctx.saved_tensors
        w = w4.reshape(w4.shape[0], w4.shape[1])
        dy = dy.contiguous()
        lib = _lib_gemm() if dy.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 else None
        dx = None
        if ctx.needs_input_grad[0]:
            dx = dy @ w if lib is None else lib.gemm_bf16(1, dy, w, torch.empty_like(rows))
        dw = None
        if ctx.needs_input_grad[1]:
            M, Co = dy.shape
            S = _wgrad_split(M)
            if (_own_wgrad() and dy.dtype == torch.bfloat16 and rows.dtype == torch.bfloat16 and Co % 64 == 0
                    and rows.shape[1] % 64 == 0 and rows.is_contiguous()):
                from . import hip
                dw = hip.conv_wgrad(dy, rows, torch.empty(Co, rows.shape[1], dtype=dy.dtype, device=dy.device))
            elif S > 1:      # K = B*H*W is long: 8 batched chunks + a sum beat every single-kernel candidate (41 vs 96 us)
                dw = torch.bmm(dy.view(S, M // S, Co).transpose(1, 2), rows.view(S, M // S, rows.shape[1])).sum(0)
            elif lib is None:
                dw = dy.t() @ rows
            else:
                dw = lib.gemm_bf16(2, dy, rows, torch.empty(Co, rows.shape[1], dtype=dy.dtype, device=dy.device))
            dw = dw.as_strided(w4.shape, w4.stride())     # [Co, Ci, 1, 1] is one memory order in either format
        return dx, dw


class Conv2d(nn.Conv2d):
    """``nn.Conv2d`` that computes with a bf16 working copy of its weight when one is attached and autocast is on
    (``ucd_amd/master.py``): no per-call cast of the fp32 master weight, no cast node in the backward.  Parameters,
    ``state_dict`` keys and the fp32 behaviour are those of ``nn.Conv2d``."""

    _w16 = None
    _w16_flip = None        # w16.flip(2, 3).transpose(0, 1) kept by ucd_amd.master.Bf16Weights (refreshed with the working copy)

    def working_weight(self):
        w = self._w16
        if w is not None and torch.is_autocast_enabled():
            return w
        return None

    def forward(self, x):
        w = self.working_weight()
        if w is None:
            return super().forward(x)
        return self._conv_forward(x, w, self.bias)


def try_index(scalar_or_list, i):
    """``x[i]`` when indexable, else ``x`` (reference ``models/util.py:1-5``)."""
    try:
        return scalar_or_list[i]
    except TypeError:
        return scalar_or_list


class _StrideOneConvFn(torch.autograd.Function):
    """Stride-1 convolution (3x3 with dilation d and padding d, or 1x1) whose INPUT gradient runs on the forward solver:
    dx = conv2d(dy, w.flip(2, 3).transpose(0, 1), padding=d, dilation=d).  Measured on MI355X with MIOpen's solver
    search (tools/dgrad_probe.py, tools/dgrad1x1_probe.py; bf16 channels-last, B = 24): the backward-data solvers take
    75 / 217 / 367 us where the forward ones take 58 / 138 / 275 us on the same problem (3x3: 256x256 at 33^2 / 512x512
    dilated / the ASPP branches), and 89 / 62 / 110 us against 43 / 32 / 73 us on the narrow 1x1 layers
    (256->64 at 129^2 / 512->128 at 65^2 / 256->128 at 129^2)."""

    @staticmethod
    def forward(ctx, x, w, d, wt=None, own_fwd=False, own_dgrad=False):
        ctx.d = d
        ctx.wt = wt          # w.flip(2, 3).transpose(0, 1), channels-last, when the caller keeps it cached (ucd_amd/master.py)
        ctx.own_dgrad = own_dgrad
        ctx.save_for_backward(x, w)
        if own_fwd:
            return _own3x3(x, w, d)
        return F.conv2d(x, w, None, 1, d * (w.shape[2] // 2), d)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        d = ctx.d
        pad = d * (w.shape[2] // 2)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            wt = ctx.wt
            if wt is None:
                wt = (w.transpose(0, 1) if w.shape[2] == 1 else w.flip(2, 3).transpose(0, 1)).contiguous(
                    memory_format=torch.channels_last)
            if ctx.own_dgrad and dy.dtype == torch.bfloat16:
                dx = _own3x3(dy.contiguous(memory_format=torch.channels_last), wt, d)
            else:
                dx = F.conv2d(dy, wt, None, 1, pad, d)
        if ctx.needs_input_grad[1]:
            dw = _own_wgrad_4d(dy, x, w, d if w.shape[2] == 3 else 0)
            if dw is None:
                dw = torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [pad, pad], [d, d], False, [0, 0], 1,
                                                         [False, True, False])[1]
        return dx, dw, None, None, None, None


def _own_wgrad_4d(dz, x, w4, dilation):
    """Weight gradient of a stride-1 convolution (dilation 0: 1x1) on the own kernel, as a tensor with the weight's sizes and
    channels-last strides; None when the layer is not one it takes (alignment, dtype, UCD_OWN_WGRAD=0)."""
    if not (_own_wgrad() and dz.is_cuda and dz.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and w4.shape[0] % 64 == 0
            and w4.shape[1] % 64 == 0 and x.shape[2] > 1 and x.shape[3] > 1 and x.is_contiguous(memory_format=torch.channels_last)
            and x.shape[0] * x.shape[2] * x.shape[3] < (1 << 22)):
        return None
    if dilation > 0 and not (tuple(w4.shape[2:]) == (3, 3) and w4.is_contiguous(memory_format=torch.channels_last)):
        return None
    from . import hip
    if not dz.is_contiguous(memory_format=torch.channels_last):
        dz = dz.contiguous(memory_format=torch.channels_last)
    N, K = w4.shape[0], w4.shape[1]
    B, _, H, W = x.shape
    rows = lambda t: t.permute(0, 2, 3, 1).reshape(B * H * W, t.shape[1])
    taps = 9 if dilation > 0 else 1
    dw = torch.empty((N, taps * K), dtype=x.dtype, device=x.device)
    hip.conv_wgrad(rows(dz), rows(x), dw, conv3=(H, W, dilation) if dilation > 0 else None)
    k = 3 if dilation > 0 else 1
    return dw.view(N, k, k, K).permute(0, 3, 1, 2)


def _own3x3(x, w, d):
    """3x3 convolution (stride 1, padding = dilation d) of a dense channels-last bf16 map on the implicit-GEMM kernel; ``w``
    [N, K, 3, 3] in channels-last memory order."""
    from . import hip
    B, K, H, W = x.shape
    N = w.shape[0]
    y = torch.empty((B, N, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    rows = lambda t: t.permute(0, 2, 3, 1).reshape(t.shape[0] * t.shape[2] * t.shape[3], t.shape[1])
    hip.conv1x1(rows(x), w.permute(0, 2, 3, 1).reshape(N, 9 * K), rows(y), conv3=(H, W, d))
    return y


def _own3x3_ok(conv, x):
    """Can this stand-alone 3x3 layer (the ASPP branches) run on the own kernel at all (layout / alignment)?"""
    return (x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and conv.bias is None and conv.stride == (1, 1)
            and conv.padding == conv.dilation and conv.dilation[0] == conv.dilation[1] and conv.groups == 1
            and conv.in_channels % 64 == 0 and conv.out_channels % 64 == 0 and x.shape[2] > 1 and x.shape[3] > 1
            and x.is_contiguous(memory_format=torch.channels_last) and _env("UCD_FUSED_CONV1X1", "1") != "0")


def _stride_one_conv(x, w, d, wt=None, own_fwd=False, own_dgrad=False):
    """The C++ autograd node when built (no Python in the backward), else the Python Function: same library calls."""
    from . import abn
    node = abn._abn_node()
    if node is not None and hasattr(node, "conv_stride1"):
        return node.conv_stride1(x, w, d, wt, bool(own_fwd), bool(own_dgrad), _hip_stream(), _own_wgrad())
    return _StrideOneConvFn.apply(x, w, d, wt, bool(own_fwd), bool(own_dgrad))


class Conv3x3(Conv2d):
    """3x3, stride 1, padding = dilation: the bottleneck conv2 layers and the ASPP branches (modules/residual.py:69,
    modules/deeplab.py:27-29).  Same parameters and state_dict keys as nn.Conv2d."""

    def forward(self, x):
        M = x.shape[0] * x.shape[2] * x.shape[3] if x.dim() == 4 else 0
        own = _own3x3_ok(self, x) and self.weight.is_contiguous(memory_format=torch.channels_last)
        if (x.is_cuda and torch.is_grad_enabled() and self.weight.requires_grad and self.bias is None
                and x.dtype == torch.bfloat16 and M >= int(_env("UCD_CONV3_MIN_ROWS", "0")) and _env("UCD_DGRAD_VIA_FWD", "1") != "0"):
            own_fwd = own and _own_conv3x3(M, self.in_channels, self.out_channels)
            own_dgrad = own and _own_conv3x3(M, self.out_channels, self.in_channels)
            w = self.working_weight()
            if w is None:
                w = self.weight.to(x.dtype)
                return _stride_one_conv(x, w, self.dilation[0], None, own_fwd and w.is_contiguous(memory_format=torch.channels_last),
                                        own_dgrad)
            return _stride_one_conv(x, w, self.dilation[0], self._w16_flip, own_fwd, own_dgrad)
        if own and not torch.is_grad_enabled() and _own_conv3x3(M, self.in_channels, self.out_channels):
            w = self.working_weight()                      # the frozen teacher's ASPP branches
            if w is None:
                w = self.weight.to(x.dtype)
            if w.is_contiguous(memory_format=torch.channels_last):
                return _own3x3(x, w, self.dilation[0])
        return super().forward(x)


class Conv1x1(Conv2d):
    """1x1 stride-1 convolution with the reference's parameter shape ([Cout, Cin, 1, 1]).  On the GPU every layer with
    64-aligned channel counts runs as a product on the channels-last row matrix [B*H*W, Cin] x [Cin, Cout] instead of an
    MIOpen implicit-GEMM convolution: inside ``_conv_abn_train`` / the teacher's fused blocks on the own kernel
    (csrc/conv1x1.hip) with the ABN work in its epilogue, called on its own through the tuned library GEMM (measured on
    MI355X, tools/gemm_vs_conv_probe.py / conv1x1_probe.py, bf16, B = 24: the library 1.2x faster than MIOpen at 1024<->256,
    1.5-1.7x at 2048<->512, 2.4x at 1024->2048; the own kernel 1.4-2.1x faster than MIOpen on the narrow layers).  Layers
    off the 64 grid (the classifier heads) stay with MIOpen."""

    def __init__(self, in_channels, out_channels, bias=False):
        super().__init__(in_channels, out_channels, 1, stride=1, padding=0, bias=bias)
        # wide: the tuned library GEMM beats MIOpen's convolution (docstring); as_gemm: the layer runs on the row matrix at all -
        # the narrow 64-aligned layers too since the own kernel exists (tools/conv1x1_probe.py: 256->64 at 129^2 39 us against
        # 55 us hipBLASLt and 61 us MIOpen; 64->256 41 / 54 / 86; 512->128 at 65^2 25 / 29 / 44; 128->512 28 / 36 / 50)
        self.wide = min(in_channels, out_channels) >= 256 and max(in_channels, out_channels) >= 1024
        self.as_gemm = self.wide or (in_channels % 64 == 0 and out_channels % 64 == 0)
        # input gradient d x = d z . w (a GEMM with K = Co, N = Ci) through the own kernel on the cached transposed weight where
        # it beats the library: short K (tools/conv1x1_probe.py: 256 -> 1024 22.8 vs 31.3 us, the narrow layers 25-41 vs 29-54)
        self.own_dgrad = (in_channels % 64 == 0 and out_channels % 64 == 0 and out_channels <= 512
                          and in_channels * out_channels <= (1 << 18))
        # set by ResidualBlock for a layer whose input comes from a conv+ABN node that feeds nothing else: its input-gradient
        # product then also does that ABN's backward reduction (out_mode 3, ``_conv_abn_train``) - worth the own kernel on the
        # transposed weight even where the plain library product is a little faster
        self.link_dgrad = False

    def forward(self, x):
        # bf16 activations only: in the fp32 parity mode (--opt_level O0) every convolution stays on one code
        # path (MIOpen), which is what the 1e-3 logit comparison with the reference was validated on
        if not (self.as_gemm and x.is_cuda and x.dim() == 4 and (x.dtype != torch.float32 or torch.is_autocast_enabled())):
            if (x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and torch.is_grad_enabled() and self.bias is None
                    and self.weight.requires_grad and x.shape[0] * x.shape[2] * x.shape[3] >= 8192
                    and _env("UCD_DGRAD_VIA_FWD", "1") != "0"):
                w = self.working_weight()
                if w is None:
                    return _stride_one_conv(x, self.weight.to(x.dtype), 1)
                return _stride_one_conv(x, w, 1, self._w16_flip)                                  # narrow layer: MIOpen
            return super().forward(x)
        B, C, H, W = x.shape
        rows = x.permute(0, 2, 3, 1)                      # a view of a channels-last tensor
        if not rows.is_contiguous():
            rows = rows.contiguous()
        rows = rows.reshape(B * H * W, C)
        w16 = self.working_weight()
        if self.bias is None and rows.dtype != torch.float32 and torch.is_grad_enabled() and self.weight.requires_grad:
            w4 = w16 if w16 is not None else self.weight.to(rows.dtype)
            node = _gemm_node() if rows.dtype == torch.bfloat16 else None
            if node is not None:          # same library calls, autograd node in C++ (host time per layer 71 -> ~25 us)
                y = node.gemm1x1(rows, w4, _hip_stream(), _own_wgrad())
            else:
                y = _Gemm1x1.apply(rows, w4)
        else:
            w = w16 if (w16 is not None and rows.dtype == w16.dtype) else self.weight
            w = w.reshape(self.out_channels, C)
            if self.bias is None and not torch.is_grad_enabled() and rows.dtype == torch.bfloat16:
                y = _mm_nt(rows, w if w.dtype == rows.dtype else w.to(rows.dtype))      # the frozen teacher
            else:
                y = F.linear(rows, w, self.bias)
        return y.view(B, H, W, self.out_channels).permute(0, 3, 1, 2)


def _conv1_with_skip(conv, x):
    """(conv(x), x') for the first 1x1 convolution of an identity-shortcut block, x' an alias of x to use as the residual:
    the C++ node folds the shortcut's gradient into the input-gradient GEMM (dx = dskip + dy w, beta = 1) instead of
    leaving a separate 3-pass add to autograd.  None when that path does not apply (then the caller does the usual)."""
    if not (isinstance(conv, Conv1x1) and conv.as_gemm and conv.bias is None and x.is_cuda and x.dim() == 4
            and x.dtype == torch.bfloat16 and torch.is_grad_enabled() and conv.weight.requires_grad and x.requires_grad):
        return None
    node = _gemm_node()
    if node is None or not hasattr(node, "gemm1x1_skip"):
        return None
    B, C, H, W = x.shape
    rows = x.permute(0, 2, 3, 1)
    if not rows.is_contiguous():
        return None
    w16 = conv.working_weight()
    y, skip = node.gemm1x1_skip(rows.reshape(B * H * W, C), w16 if w16 is not None else conv.weight.to(x.dtype), _hip_stream(),
                                _own_wgrad())
    return (y.view(B, H, W, conv.out_channels).permute(0, 3, 1, 2), skip.view(B, H, W, C).permute(0, 3, 1, 2))


def _is_fused_abn(m) -> bool:
    return getattr(m, "ucd_fused_abn", False)


def _own_gemm_with_stats(K, N):
    """Forward kernel choice of a conv + training-ABN pair, from tools/conv1x1_probe.py on MI355X (M = 26 136,
    profiles/r03_conv1x1_probe.txt): the fused GEMM with the statistics epilogue costs its plain time + 1-6 us; the tuned library
    GEMM needs a statistics pass over its output on top (11-30 us).  Since the staging loads go through buffer descriptors the own
    kernel is level or ahead on every layer of the network, the two large products included (2048 -> 512: 59.6 against 43.5 + 16;
    1024 -> 2048: 110.8 against 93.1 + 30), so every aligned 1x1 layer takes it; ``UCD_LIB_GEMM_WIDE=1`` restores the library
    for those two (A/B)."""
    if _env("UCD_LIB_GEMM_WIDE", "0") == "1":
        return not (K * N >= (1 << 20) and K >= 1024)
    return True


def _own_stride(conv):
    """The stride s > 1 of a layer the own strided kernels take (csrc/conv1x1.hip ``stride``, csrc/wgrad.hip STR): the first
    block of a stage - conv2 3x3 with padding = dilation and proj_conv 1x1 with padding 0 (modules/residual.py:57-82), 128-aligned
    channels - else 0.  MIOpen needs 209-263 us for each of the four forward products at B = 24 (tools/aten_ops.py), 5-15x their
    traffic / MFMA time.  ``UCD_OWN_STRIDED=0`` keeps the library (A/B)."""
    if not (isinstance(conv, Conv2d) and conv.stride[0] == conv.stride[1] and conv.stride[0] > 1 and conv.groups == 1
            and conv.in_channels % 128 == 0 and conv.out_channels % 128 == 0 and _env("UCD_OWN_STRIDED", "1") != "0"):
        return 0
    if conv.kernel_size == (1, 1) and conv.padding == (0, 0):
        return conv.stride[0]
    if conv.kernel_size == (3, 3) and conv.padding == conv.dilation and conv.dilation[0] == conv.dilation[1]:
        return conv.stride[0]
    return 0


def _own_conv3x3(M, K, N):
    """The implicit-GEMM 3x3 (csrc/conv1x1.hip, taps = 9) instead of MIOpen: measured on MI355X (tools/conv3x3_probe.py,
    B = 24, profiles/r03_conv3x3_probe.txt): 256->256 at 33^2 40 vs 58 us, 512->512 (dilation 2) 132 vs 142, 128->128 at 65^2
    42 vs 48, 64->64 at 129^2 51 vs 63, the ASPP branches 229-252 vs 386-391 - and the following ABN's statistics for +1 us
    instead of a separate pass.  Maps too small to give every CU a tile (3-12 images per GPU, the multi-GPU split of the batch) take
    it as well: the kernel alone is level with MIOpen's there (23 vs 27 us at 3 images) but the step saves MIOpen's weight-gradient
    zero fills / casts and the separate statistics passes - kernel time per step 16.0 -> 14.3 ms at 3 images, 20.0 -> 17.5 at 6
    (profiles/r03_small_batch.txt).  ``UCD_OWN3X3_MIN_TILES=256`` restores the library below 256 tiles (A/B)."""
    tiles = ((M + 127) // 128) * max(1, N // 128)
    if K >= 512 and N >= 512 and _env("UCD_OWN3X3_WIDE", "1") == "0":      # A/B switch: the 512 -> 512 layers on MIOpen
        return False
    return tiles >= int(_env("UCD_OWN3X3_MIN_TILES", "1"))


class _ConvABNFunction(torch.autograd.Function):
    """Python twin of csrc/abn_node.cpp::ConvABNTrainNode (single process; the node adds the SyncBN exchange and the
    shortcut fold): z = x . w^T with the statistics in the GEMM epilogue -> finalize -> y = act(norm(z) [+ residual]).
    The complete implementation and the fallback; bench.py's instrumented pass runs it (every library call visible)."""

    @staticmethod
    def forward(ctx, x, w4, weight, bias, residual, running_mean, running_var, momentum, eps, act, slope, fused, dilation=0,
                wflip=None, own_dgrad=False, wgrad_conv=False, make_link=False, link=None, with_skip=False, blink=None):
        from . import hip
        B, K, H, W = x.shape
        N = w4.shape[0]
        M, HW = B * H * W, H * W
        rows = lambda t: t.permute(0, 2, 3, 1).reshape(t.shape[0] * t.shape[2] * t.shape[3], t.shape[1])
        z = torch.empty((B, N, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        y = torch.empty_like(z)
        buf = torch.empty(6 * N, dtype=torch.float32, device=x.device)
        conv3 = dilation > 0
        w2 = w4.permute(0, 2, 3, 1).reshape(N, 9 * K) if conv3 else w4.reshape(N, K)
        ctx.conv3 = (dilation, wflip, own_dgrad, wgrad_conv)
        # backward link (see ConvABNTrainNode): link = the producer's (z, buf, bias, partial, flag, act, slope)
        ctx.link = link
        ctx.blink = blink if with_skip else None        # block link (kind 3): (z, buf, partial, flag, slope) of the block in front
        ctx.with_skip = with_skip
        ctx.my_link = None
        if conv3 and not fused:
            z = F.conv2d(x, w4, None, 1, dilation, dilation).contiguous(memory_format=torch.channels_last)
            hip.abn_forward(z, N, y, N, residual, N if residual is not None else 0, M, N, None, HW, weight, bias, running_mean,
                            running_var, momentum, eps, True, buf, None, act, slope)
        elif fused:
            part = hip.conv1x1_stats_partial(M, N, x.device)
            hip.conv1x1(rows(x), w2, rows(z), out_mode=2, partial=part, conv3=(H, W, dilation) if conv3 else None)
            hip.conv1x1_stats_finalize(part, M, N, weight, running_mean, running_var, momentum, eps, buf, None, act)
            hip.abn_apply(z, N, y, N, residual, N if residual is not None else 0, M, N, None, HW, buf[3 * N:4 * N], buf[5 * N:],
                          bias, act, slope)
        else:
            hip.gemm_bf16(0, rows(x), w2, rows(z))
            hip.abn_forward(z, N, y, N, residual, N if residual is not None else 0, M, N, None, HW, weight, bias, running_mean,
                            running_var, momentum, eps, True, buf, None, act, slope)
        needs_y = residual is not None and (act & hip.ACT_MASK) != hip.ACT_IDENTITY
        ctx.save_for_backward(x, w4, z, y if needs_y else None, weight, bias, buf)
        ctx.cfg = (act, slope, residual is not None)
        if make_link and bias is not None and (act & hip.ACT_MASK) != hip.ACT_ELU and (residual is None or needs_y):
            partial = torch.empty(hip.conv1x1_row_tiles(M), 2, N, dtype=torch.float32, device=x.device)
            ctx.my_link = (partial, [0, 0, 0])   # {served, address of the consumer's dx, its version} (see ConvABNTrainNode::backward)
            if residual is None:
                y._ucd_link = (z, buf, bias, partial, ctx.my_link[1], act, slope)
            else:                                # the block's last node: the NEXT block's conv1 + shortcut node may serve it
                y._ucd_blink = (z, buf, partial, ctx.my_link[1], slope)
        if with_skip:
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        from . import hip
        x, w4, z, y, weight, bias, buf = ctx.saved_tensors
        act, slope, has_res = ctx.cfg
        B, K, H, W = x.shape
        N = w4.shape[0]
        M, HW = B * H * W, H * W
        rows = lambda t: t.permute(0, 2, 3, 1).reshape(t.shape[0] * t.shape[2] * t.shape[3], t.shape[1])
        if ctx.my_link is not None and ctx.my_link[1][0] == 1 and (dy.data_ptr() != ctx.my_link[1][1]
                                                                   or dy._version != ctx.my_link[1][2]):
            ctx.my_link[1][0] = 0
            raise RuntimeError("ucd conv+abn node: the backward link was served but the gradient that arrived is not the consumer's "
                               "input gradient - the linked map has a second consumer (hook, ret_intermediate tap, retain_graph "
                               "replay).  Run with UCD_BWD_LINK=0.")
        dy_in = dy
        dy, _, _, _, _ = hip.rows_view(dy if dy.dtype == x.dtype else dy.to(x.dtype))
        dz = torch.empty_like(z)
        linked = ctx.my_link is not None and ctx.my_link[1][0] == 1
        dres = torch.empty_like(z) if (has_res and not linked) else None
        sums = torch.empty(2 * N, dtype=torch.float32, device=x.device)
        if linked:
            # the consumer's input-gradient product applied the activation derivative and left the sums as per-tile partials
            ctx.my_link[1][0] = 0
            part = ctx.my_link[0]
            hip._check(hip.load().ucd_abn_reduce_partials(hip.ptr(part), part.shape[0], N, hip.ptr(sums), None, hip.ptr(weight),
                                                          act & hip.NORM_ABS_GAMMA, hip.stream()), "ucd_abn_reduce_partials")
            hip.abn_bwd_apply(z, N, dy, N, None, 0, dz, N, None, 0, M, N, None, HW, buf[3 * N:4 * N], buf[4 * N:5 * N], buf[5 * N:],
                              bias, weight, sums, float(M), 0, hip.ACT_IDENTITY | (act & hip.NORM_ABS_GAMMA), 0.0)
            if has_res:
                dres = dy_in            # block link: dy IS d pre, and so is the shortcut's gradient
        else:
            hip.abn_backward(z, N, dy, N, y, N if y is not None else 0, dz, N, dres, N if has_res else 0, M, N, None, HW,
                             buf[3 * N:4 * N], buf[4 * N:5 * N], buf[5 * N:], bias, weight, sums, float(M), True, True, act, slope)
        dilation, wflip, own_dgrad, wgrad_conv = ctx.conv3
        link = ctx.link

        def link_args(dx):     # out_mode 3 against the producer's statistics; marks the link as served (for THIS dx)
            lz, lbuf, lbias, lpart, lflag, lact, lslope = link
            C = lz.shape[1]
            lflag[0] = 1
            lflag[1] = dx.data_ptr()
            lflag[2] = dx._version
            return dict(out_mode=3, out_norm=(lbuf[3 * C:4 * C], lbuf[5 * C:], lbias, lbuf[4 * C:5 * C], lact & hip.ACT_MASK, lslope),
                        residual=rows(lz), partial=lpart)
        if dilation > 0:
            dx = dw = None
            if ctx.needs_input_grad[0]:
                if wflip is None:
                    wflip = w4.flip(2, 3).transpose(0, 1).contiguous(memory_format=torch.channels_last)
                if own_dgrad:
                    dx = torch.empty_like(x)
                    hip.conv1x1(rows(dz), wflip.permute(0, 2, 3, 1).reshape(K, 9 * N), rows(dx), conv3=(H, W, dilation),
                                **(link_args(dx) if link is not None else {}))
                else:
                    dx = F.conv2d(dz, wflip, None, 1, dilation, dilation)
            if ctx.needs_input_grad[1]:
                dw = _own_wgrad_4d(dz, x, w4, dilation) if wgrad_conv == 2 else None
                if dw is None:
                    dw = torch.ops.aten.convolution_backward(dz, x, w4, None, [1, 1], [dilation, dilation], [dilation, dilation],
                                                             False, [0, 0], 1, [False, True, False])[1]
            return dx, dw, sums[N:], sums[:N], dres, None, None, None, None, None, None, None, None, None, None, None, None, None, None, None
        w2 = w4.reshape(N, K)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            fold = (dskip is not None and dskip.dtype == x.dtype
                    and dskip.is_contiguous(memory_format=torch.channels_last) and own_dgrad and wflip is not None)
            dx = dskip if fold else torch.empty_like(x)
            if own_dgrad and wflip is not None:
                extra = {}
                if ctx.blink is not None and (fold or dskip is None):
                    bz, bbuf, bpart, bflag, bslope = ctx.blink            # block link: out_mode 4 against the block in front
                    bflag[0], bflag[1], bflag[2] = 1, dx.data_ptr(), dx._version
                    extra = dict(out_mode=4, out_norm=(bbuf[3 * K:4 * K], None, None, bbuf[4 * K:5 * K], hip.ACT_LEAKY_RELU, bslope),
                                 residual=rows(x), side2=rows(bz), partial=bpart)
                elif link is not None and dskip is None:
                    extra = link_args(dx)
                hip.conv1x1(rows(dz), wflip.reshape(K, N), rows(dx), accumulate=fold, **extra)
            else:
                hip.gemm_bf16(1, rows(dz), w2, rows(dx))
            if dskip is not None and not fold:
                dx = dx + dskip
        if ctx.needs_input_grad[1] and wgrad_conv == 2:
            dw = _own_wgrad_4d(dz, x, w4, 0)
            if dw is None:
                wgrad_conv = 1
        if dw is not None:
            pass
        elif ctx.needs_input_grad[1] and wgrad_conv == 1:
            dw = torch.ops.aten.convolution_backward(dz, x, w4, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1,
                                                     [False, True, False])[1]
        elif ctx.needs_input_grad[1]:
            S = _wgrad_split(M)
            dzr, xr = rows(dz), rows(x)
            dw = (torch.bmm(dzr.view(S, M // S, N).transpose(1, 2), xr.view(S, M // S, K)).sum(0) if S > 1 else dzr.t() @ xr)
            dw = dw.as_strided(w4.shape, w4.stride())
        return dx, dw, sums[N:], sums[:N], dres, None, None, None, None, None, None, None, None, None, None, None, None, None, None, None


def _conv_abn_train(conv, bn, x, residual=None, activation=None, activation_param=None, with_skip=False, make_link=False):
    """``bn(conv(x) [, residual])`` of a wide 1x1 convolution and a training-mode HIP ABN as ONE autograd node
    (csrc/abn_node.cpp::ConvABNTrainNode): the ABN's batch statistics come out of the GEMM's epilogue.  Returns None when the
    pair is not eligible (the caller then runs the modules one after the other), else ``y`` or ``(y, shortcut alias of x)``.
    ``make_link``: the caller promises that ``y`` feeds exactly one further ``_conv_abn_train`` call and nothing else; ``y`` then
    carries the link (``y._ucd_link``) through which that consumer's input-gradient product does this ABN's backward reduction
    in its epilogue (csrc/abn_node.cpp; under SyncBN the producer all-reduces the combined sums; ``UCD_BWD_LINK=0`` switches it
    off)."""
    if _env("UCD_FUSED_CONV1X1", "1") == "0":
        return None
    link = getattr(x, "_ucd_link", None) if _env("UCD_BWD_LINK", "1") != "0" else None
    # block link (csrc/abn_node.cpp: block_link_epilogue): x is the output of a residual block whose last node offers its
    # backward reduction to the first convolution of THIS identity-shortcut block (UCD_BLOCK_LINK=0 switches only this kind off)
    blink = (getattr(x, "_ucd_blink", None) if (with_skip and _env("UCD_BWD_LINK", "1") != "0"
                                                and _env("UCD_BLOCK_LINK", "1") != "0") else None)
    make_link = make_link and _env("UCD_BWD_LINK", "1") != "0"
    if residual is not None and _env("UCD_BLOCK_LINK", "1") == "0":
        make_link = False
    is3 = isinstance(conv, Conv3x3)
    stride = _own_stride(conv) if not (with_skip or residual is not None) else 0
    if stride:
        is3 = conv.kernel_size == (3, 3)
    if not ((is3 or stride or (isinstance(conv, Conv1x1) and conv.as_gemm)) and conv.bias is None and conv.weight.requires_grad
            and _is_fused_abn(bn) and bn.training and bn.weight is not None and torch.is_grad_enabled() and x.is_cuda
            and x.dim() == 4 and x.dtype == torch.bfloat16 and conv.in_channels % 64 == 0 and conv.out_channels % 64 == 0):
        return None
    dilation, wflip, fused, own_dgrad, wgrad_conv = 0, None, None, False, 0      # wgrad_conv: 0 batched products, 1 MIOpen, 2 own
    if stride:
        # the strided layers of a stage's first block: forward (+ statistics) and weight gradient on the own kernels, the input
        # gradient stays with the library's backward-data solver; no link consumed (the producer falls back to its own reduction)
        if is3 and not conv.weight.is_contiguous(memory_format=torch.channels_last):
            return None
        dilation, fused, link, blink = (conv.dilation[0] if is3 else 0), True, None, None
        wgrad_conv = 1
    elif is3:
        if not (conv.stride == (1, 1) and conv.padding == conv.dilation and conv.dilation[0] == conv.dilation[1]
                and conv.groups == 1 and not with_skip and conv.weight.is_contiguous(memory_format=torch.channels_last)):
            return None
        M = x.shape[0] * x.shape[2] * x.shape[3]
        dilation = conv.dilation[0]
        fused = _own_conv3x3(M, conv.in_channels, conv.out_channels)
        own_dgrad = _own_conv3x3(M, conv.out_channels, conv.in_channels)
        if not fused and not own_dgrad:
            return None                         # nothing of ours to gain: the module path (MIOpen + ABN node) stays
        wflip = conv._w16_flip if conv.working_weight() is not None else None
    else:
        fused = _own_gemm_with_stats(conv.in_channels, conv.out_channels)
        own_dgrad = conv.own_dgrad or (link is not None and conv.link_dgrad and not with_skip) or (blink is not None and conv.link_dgrad)
        # the transposed weight: cached with the bf16 working copies, else made per call below (same kernels either way)
        wflip = conv._w16_flip if (own_dgrad and conv.working_weight() is not None) else None
        wgrad_conv = 0 if conv.wide else 1
    if _own_wgrad():
        wgrad_conv = 2
    if link is not None and (not own_dgrad or with_skip):
        link = None                              # the consumer's input gradient does not run on the own kernel: no link
    if blink is not None and (not own_dgrad or is3):
        blink = None
    from . import abn as _abn
    from . import hip
    node = _gemm_node()
    if node is None or not hasattr(node, "conv_abn_train"):
        # no C++ node (not built, or switched off by bench.py's instrumented pass): the Python twin, single process only
        if stride:
            return None                          # the twin has no strided form: the module path
        dense = lambda t: t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last) and t.shape[2] > 1 and t.shape[3] > 1
        if (_lib_gemm() is None or _abn._group_size(bn._group()) > 1 or not dense(x)
                or (residual is not None and not (dense(residual) and residual.dtype == x.dtype))):
            return None
        w16 = conv.working_weight()
        if w16 is None:
            w16 = conv.weight.to(x.dtype)
        if own_dgrad and not is3 and wflip is None:
            wflip = w16.reshape(conv.out_channels, conv.in_channels).t().contiguous().view(conv.in_channels, conv.out_channels, 1, 1)
        act = _abn._act_code(bn.activation if activation is None else activation) | (hip.NORM_ABS_GAMMA if bn._abs_gamma else 0)
        slope = bn.activation_param if activation_param is None else activation_param
        bn.__dict__.pop("_eval_cache", None)
        return _ConvABNFunction.apply(x, w16, bn.weight, bn.bias, residual, bn.running_mean, bn.running_var, bn.momentum, bn.eps,
                                      act, slope, fused, dilation, wflip, own_dgrad, wgrad_conv, bool(make_link), link,
                                      bool(with_skip), blink)
    if not node.dense_channels_last(x):
        return None
    if residual is not None and not (residual.dtype == x.dtype and node.dense_channels_last(residual)):
        return None
    group = bn._group()
    world = _abn._group_size(group)
    sync = group is not False and (world > 1 or (_abn._FORCE_SYNC and torch.distributed.is_initialized()))
    comm = _abn.direct_comm(group) if sync else None
    if sync and comm is None:
        return None
    w16 = conv.working_weight()
    if w16 is None:
        w16 = conv.weight.to(x.dtype)
    if own_dgrad and not is3 and wflip is None:
        wflip = w16.reshape(conv.out_channels, conv.in_channels).t().contiguous().view(conv.in_channels, conv.out_channels, 1, 1)
    act = _abn._act_code(bn.activation if activation is None else activation) | (hip.NORM_ABS_GAMMA if bn._abs_gamma else 0)
    slope = bn.activation_param if activation_param is None else activation_param
    bn.__dict__.pop("_eval_cache", None)
    out = node.conv_abn_train(x, w16, bn.weight, bn.bias, residual, bn.running_mean, bn.running_var, bn.momentum, bn.eps, act,
                              slope, comm.handle if comm is not None else 0, world, _hip_stream(), bn._direct_grad_ptr(),
                              bool(with_skip), bool(fused), dilation, wflip, bool(own_dgrad), int(wgrad_conv),
                              bool(make_link),
                              *((blink[0], blink[1], None, blink[2], blink[3], 0, float(blink[4]), 3) if blink is not None else
                                (link[0], link[1], link[2], link[3], link[4], int(link[5]), float(link[6]), 1) if link is not None
                                else (None, None, None, None, None, 0, 0.0, 0)), int(stride) if stride else 1,
                              _env("UCD_STAT_ATOMIC", "1") != "0")
    k = 2 if with_skip else 1
    if len(out) > k:                             # the node made a link: (z, buf, partial, flag) follow the regular outputs
        if residual is None:
            out[0]._ucd_link = (out[k], out[k + 1], bn.bias, out[k + 2], out[k + 3], act, slope)
        else:
            out[0]._ucd_blink = (out[k], out[k + 1], out[k + 2], out[k + 3], slope)
    return (out[0], out[1]) if with_skip else out[0]


def _apply_act(x, activation, param):
    if activation == "leaky_relu":
        return F.leaky_relu(x, negative_slope=param, inplace=True)
    if activation == "elu":
        return F.elu(x, alpha=param, inplace=True)
    if activation == "identity":
        return x
    raise RuntimeError(f"unknown activation {activation!r}")


class ResidualBlock(nn.Module):
    """``channels`` of length 3 -> bottleneck 1x1 / 3x3(stride, dilation) / 1x1, length 2 -> basic
    3x3 / 3x3.  The last norm of the block is created like the others and then switched to
    ``activation = "identity"`` (reference ``residual.py:52,65``); the projection shortcut exists
    when the stride or the channel count changes (``residual.py:48,79-82``).
    """

    def __init__(self, in_channels, channels, stride=1, dilation=1, groups=1,
                 norm_act=nn.BatchNorm2d, dropout=None):
        super().__init__()
        if len(channels) not in (2, 3):
            raise ValueError("channels must contain either two or three values")
        if len(channels) == 2 and groups != 1:
            raise ValueError("groups > 1 are only valid if len(channels) == 3")

        def conv(cin, cout, k, s=1, g=1):
            if k == 1 and s == 1 and g == 1:
                return Conv1x1(cin, cout)
            pad = dilation if k == 3 else 0
            if k == 3 and s == 1 and g == 1:
                return Conv3x3(cin, cout, 3, stride=1, padding=dilation, dilation=dilation, bias=False)
            return Conv2d(cin, cout, k, stride=s, padding=pad, dilation=dilation if k == 3 else 1,
                             groups=g, bias=False)

        if len(channels) == 3:
            spec = [(1, 1, 1), (3, stride, groups), (1, 1, 1)]
        else:
            spec = [(3, stride, 1), (3, 1, 1)]
        layers, cin = [], in_channels
        for i, ((k, s, g), cout) in enumerate(zip(spec, channels), 1):
            layers.append((f"conv{i}", conv(cin, cout, k, s, g)))
            layers.append((f"bn{i}", norm_act(cout)))
            if dropout is not None and i == len(spec) - 1:
                layers.append(("dropout", dropout()))
            cin = cout
        layers[-1][1].activation = "identity"
        self.convs = nn.Sequential(OrderedDict(layers))
        self._last_bn = f"bn{len(spec)}"
        c3 = getattr(self.convs, "conv3", None)
        if (isinstance(c3, Conv1x1) and isinstance(getattr(self.convs, "conv2", None), Conv3x3) and c3.in_channels % 64 == 0
                and c3.out_channels % 64 == 0 and c3.out_channels <= 1024):
            c3.link_dgrad = True          # conv2 + bn2 -> conv3: bn2's backward reduction rides on conv3's input gradient

        c1 = getattr(self.convs, "conv1", None)
        if (stride == 1 and in_channels == channels[-1] and isinstance(c1, Conv1x1) and len(channels) == 3
                and c1.in_channels % 64 == 0 and c1.out_channels % 64 == 0):
            # identity shortcut: conv1's input-gradient product (with the shortcut's gradient folded in) forms the gradient
            # w.r.t. the previous block's output - the block link has it do that block's bn3 backward reduction too, which
            # needs the own kernel on the cached transposed weight
            c1.link_dgrad = True

        if stride != 1 or in_channels != channels[-1]:
            self.proj_conv = (Conv1x1(in_channels, channels[-1]) if stride == 1 else
                              Conv2d(in_channels, channels[-1], 1, stride=stride, padding=0, bias=False))
            self.proj_bn = norm_act(channels[-1])
            self.proj_bn.activation = "identity"

    # -- frozen-statistics forward (the teacher): each 1x1 convolution and the ABN that follows it are ONE kernel ----------
    def _eval_fusable(self, x):
        c = self.convs
        if not (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and len(c) >= 6 and hasattr(c, "conv3")):
            return False
        if not (isinstance(c.conv1, Conv1x1) and isinstance(c.conv3, Conv1x1) and c.conv1.bias is None and c.conv3.bias is None):
            return False
        norms = [c.bn1, c.bn2, c.bn3] + ([self.proj_bn] if hasattr(self, "proj_conv") else [])
        if not all(_is_fused_abn(m) and m.activation in ("leaky_relu", "identity") and m.weight is not None for m in norms):
            return False
        chans = (c.conv1.in_channels, c.conv1.out_channels, c.conv3.in_channels, c.conv3.out_channels)
        st = x.stride()
        return (all(v % 64 == 0 for v in chans) and "dropout" not in c._modules
                and st[1] == 1 and st[3] == x.shape[1] and _env("UCD_FUSED_CONV1X1", "1") != "0")

    @staticmethod
    def _eval_norm(m, act=None, slope=None):
        """(mean, scale, shift, invstd, act code, slope) of an ABN layer under its running statistics."""
        from . import hip
        consts = m._eval_constants()
        a = m.activation if act is None else act
        return (m.running_mean, consts[1], m.bias, None, hip.ACT_CODES[a], m.activation_param if slope is None else slope)

    def _forward_eval_fused(self, x):
        """conv1 + bn1 -> one GEMM with the affine + activation epilogue; conv3 + bn3 + shortcut + block activation -> one
        GEMM with the affine + residual + activation epilogue (csrc/conv1x1.hip); conv2 stays MIOpen, bn2 its in-place
        apply.  Same arithmetic as the layer-by-layer path with the conv outputs kept in fp32 up to the activation."""
        from . import hip
        c = self.convs
        B, C, H, W = x.shape
        cl = torch.channels_last

        def rows(t):
            b, ch, h, w = t.shape
            return t.permute(0, 2, 3, 1).reshape(b * h * w, ch)

        def w2d(conv, like):
            w = conv.working_weight()
            if w is None:
                w = conv.weight.to(like.dtype)
            return w.reshape(conv.out_channels, conv.in_channels)

        xr = rows(x)
        h1 = torch.empty((B, c.conv1.out_channels, H, W), dtype=x.dtype, device=x.device, memory_format=cl)
        hip.conv1x1(xr, w2d(c.conv1, x), rows(h1), out_mode=1, out_norm=self._eval_norm(c.bn1))
        cv = c.conv2
        if (isinstance(cv, Conv3x3) and cv.stride == (1, 1) and cv.padding == cv.dilation and cv.bias is None
                and cv.in_channels % 64 == 0 and cv.out_channels % 64 == 0
                and _own_conv3x3(B * H * W, cv.in_channels, cv.out_channels)):
            w3 = cv.working_weight()
            if w3 is None:
                w3 = cv.weight.to(x.dtype)
            if not w3.is_contiguous(memory_format=cl):
                w3 = w3.contiguous(memory_format=cl)
            h2 = torch.empty((B, cv.out_channels, H, W), dtype=x.dtype, device=x.device, memory_format=cl)
            hip.conv1x1(rows(h1), w3.permute(0, 2, 3, 1).reshape(cv.out_channels, 9 * cv.in_channels), rows(h2), out_mode=1,
                        out_norm=self._eval_norm(c.bn2), conv3=(H, W, cv.dilation[0]))       # conv2 + bn2: one kernel
        elif _own_stride(cv) and cv.kernel_size == (3, 3):          # the stage's first block: conv2 with a stride, + bn2
            st = _own_stride(cv)
            w3 = cv.working_weight()
            if w3 is None:
                w3 = cv.weight.to(x.dtype)
            if not w3.is_contiguous(memory_format=cl):
                w3 = w3.contiguous(memory_format=cl)
            h2 = torch.empty((B, cv.out_channels, (H - 1) // st + 1, (W - 1) // st + 1), dtype=x.dtype, device=x.device,
                             memory_format=cl)
            hip.conv1x1(rows(h1), w3.permute(0, 2, 3, 1).reshape(cv.out_channels, 9 * cv.in_channels), rows(h2), out_mode=1,
                        out_norm=self._eval_norm(c.bn2), conv3=(H, W, cv.dilation[0], st))
        else:
            h2 = c.bn2(c.conv2(h1))                                # in place under no_grad (InPlaceABN contract)
        if hasattr(self, "proj_conv"):
            pst = _own_stride(self.proj_conv)
            if isinstance(self.proj_conv, Conv1x1) and self.proj_conv.in_channels % 64 == 0:
                res = torch.empty((B, self.proj_conv.out_channels, H, W), dtype=x.dtype, device=x.device, memory_format=cl)
                hip.conv1x1(xr, w2d(self.proj_conv, x), rows(res), out_mode=1, out_norm=self._eval_norm(self.proj_bn))
            elif pst and self.proj_conv.kernel_size == (1, 1):     # strided shortcut projection + proj_bn: one row-gather product
                res = torch.empty((B, self.proj_conv.out_channels, (H - 1) // pst + 1, (W - 1) // pst + 1), dtype=x.dtype,
                                  device=x.device, memory_format=cl)
                hip.conv1x1(xr, w2d(self.proj_conv, x), rows(res), out_mode=1, out_norm=self._eval_norm(self.proj_bn),
                            strided=(H, W, pst))
            else:
                res = self.proj_bn(self.proj_conv(x))
        else:
            res = x
        if not h2.is_contiguous(memory_format=cl):
            h2 = h2.contiguous(memory_format=cl)
        if not res.is_contiguous(memory_format=cl):
            res = res.contiguous(memory_format=cl)
        out = torch.empty((h2.shape[0], c.conv3.out_channels, h2.shape[2], h2.shape[3]), dtype=x.dtype, device=x.device,
                          memory_format=cl)
        hip.conv1x1(rows(h2), w2d(c.conv3, x), rows(out), out_mode=1,
                    out_norm=self._eval_norm(c.bn3, c.bn1.activation, c.bn1.activation_param), residual=rows(res))
        return out

    def forward(self, x):
        if not self.training and not torch.is_grad_enabled() and self._eval_fusable(x):
            return self._forward_eval_fused(x)
        act, slope = self.convs.bn1.activation, self.convs.bn1.activation_param
        last = getattr(self.convs, self._last_bn)
        fused_train = (self.training and _is_fused_abn(last) and last.activation == "identity" and self._last_bn == "bn3"
                       and "dropout" not in self.convs._modules and x.is_cuda and x.dtype == torch.bfloat16)

        def project(t):
            r = _conv_abn_train(self.proj_conv, self.proj_bn, t) if self.training else None
            return r if r is not None else self.proj_bn(self.proj_conv(t))

        has_proj = hasattr(self, "proj_conv")
        # projection blocks in training: the block input feeds conv1 AND the projection; like the identity blocks, conv1's node
        # returns an alias of x for the second consumer, so x has ONE consumer and the projection's input gradient arrives as
        # the accumulate operand of conv1's input-gradient product instead of a separate 3-pass add (UCD_PROJ_ALIAS=0: A/B)
        defer = has_proj and fused_train and x.requires_grad and _env("UCD_PROJ_ALIAS", "1") != "0"
        residual = x if not has_proj else (None if defer else project(x))
        if fused_train:
            # wide bottleneck in training: every 1x1 convolution and its ABN are one node (statistics in the GEMM epilogue)
            c = self.convs
            skip = (residual is x or defer) and x.requires_grad
            first = _conv_abn_train(c.conv1, c.bn1, x, with_skip=skip, make_link=True)      # h1 feeds conv2 only
            if first is None and defer:
                residual = project(x)
            if first is not None:
                h1, res = first if skip else (first, residual)
                if defer:
                    res = project(res)                       # the alias of x: the shortcut's only path to the block input
                h2 = _conv_abn_train(c.conv2, c.bn2, h1, make_link=True)      # 3x3 as implicit GEMM + statistics; h2 feeds conv3 only
                if h2 is None:
                    h2 = c.bn2(c.conv2(h1))
                out = _conv_abn_train(c.conv3, c.bn3, h2, residual=res, activation=act, activation_param=slope, make_link=True)
                if out is None:
                    out = c.bn3(c.conv3(h2), residual=res, activation=act, activation_param=slope)
                return out
        if _is_fused_abn(last) and last.activation == "identity" and act in ("leaky_relu", "identity"):
            # fused epilogue: act(bn(conv_out) + residual) in one HBM pass
            h, first = x, None
            if residual is x and self.training:
                first = _conv1_with_skip(self.convs.conv1, x)
                if first is not None:
                    h, residual = first
            bn2_done = False
            for name, mod in self.convs.named_children():
                if mod is last:
                    h = mod(h, residual=residual, activation=act, activation_param=slope)
                elif first is not None and name == "conv1":
                    continue                                   # already applied together with the shortcut
                elif name == "conv2" and self.training and hasattr(self.convs, "bn2") and self.convs.bn2 is not last:
                    y = _conv_abn_train(mod, self.convs.bn2, h)   # narrow blocks: the 3x3 + its ABN as one node
                    bn2_done = y is not None
                    h = y if bn2_done else mod(h)
                elif name == "bn2" and bn2_done:
                    continue
                else:
                    h = mod(h)
            return h
        return _apply_act(self.convs(x) + residual, act, slope)


class DeeplabV3(nn.Module):
    """ASPP head: four parallel 2048->256 convs (1x1 and 3x3 with dilation 6/12/18 at output
    stride 16, 12/24/32 at 8), ABN over the 1024 concatenated channels, 1x1 reduction, plus the
    image-level pooling branch added before the last ABN (reference ``deeplab.py:54-70``).
    """

    def __init__(self, in_channels, out_channels, hidden_channels=256, out_stride=16,
                 norm_act=nn.BatchNorm2d, pooling_size=None):
        super().__init__()
        self.pooling_size = pooling_size
        if out_stride == 16:
            dilations = (6, 12, 18)
        elif out_stride == 8:
            dilations = (12, 24, 32)
        else:
            raise ValueError("out_stride must be 8 or 16")
        self.hidden_channels = hidden_channels

        branches = [Conv1x1(in_channels, hidden_channels)]
        branches += [Conv3x3(in_channels, hidden_channels, 3, bias=False, dilation=d, padding=d)
                     for d in dilations]
        self.map_convs = nn.ModuleList(branches)
        self.map_bn = norm_act(hidden_channels * len(branches))

        self.global_pooling_conv = Conv2d(in_channels, hidden_channels, 1, bias=False)
        self.global_pooling_bn = norm_act(hidden_channels)

        self.red_conv = Conv1x1(hidden_channels * len(branches), out_channels)
        self.pool_red_conv = Conv2d(hidden_channels, out_channels, 1, bias=False)
        self.red_bn = norm_act(out_channels)

        self.reset_parameters(self.map_bn.activation, self.map_bn.activation_param)

    def reset_parameters(self, activation, slope):
        """Xavier-normal conv weights with the activation's gain, unit/zero affine norms
        (reference ``deeplab.py:41-52``)."""
        gain = nn.init.calculate_gain(activation, slope)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_normal_(m.weight.data, gain)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d) or _is_fused_abn(m):
                if getattr(m, "weight", None) is not None:
                    nn.init.constant_(m.weight, 1)
                if getattr(m, "bias", None) is not None:
                    nn.init.constant_(m.bias, 0)

    # -- image-level pooling -----------------------------------------------------------------
    def _global_pooling(self, x):
        """(pooled map, replicate-padding still to apply or None).  Training: the plain global average [B, C, 1, 1].
        Evaluation: sliding average of (at most) pooling_size, stride 1, replicate-padded back to the input size
        (reference deeplab.py:77-88).  The reference pads first and runs the branch's two 1x1 convolutions and its
        norm on the full-size map; they are pointwise, so they commute with the replication and run here on the
        small pooled map (2x2 at 513^2 with the default --pooling 32) - same values, 270x fewer pixels."""
        if self.training or self.pooling_size is None:
            if _is_fused_abn(self.red_bn):
                from . import abn as _abn
                return _abn.global_avg_pool(x), None
            return x.flatten(2).mean(dim=-1)[:, :, None, None], None
        ph = min(try_index(self.pooling_size, 0), x.shape[2])
        pw = min(try_index(self.pooling_size, 1), x.shape[3])
        pad = ((pw - 1) // 2, (pw - 1) // 2 + (1 - pw % 2),
               (ph - 1) // 2, (ph - 1) // 2 + (1 - ph % 2))
        if (x.is_cuda and _is_fused_abn(self.red_bn) and not torch.is_grad_enabled()
                and x.shape[1] % (8 if x.dtype == torch.bfloat16 else 4) == 0 and x.dtype in (torch.bfloat16, torch.float32)):
            from . import hip
            return hip.window_mean(x, ph, pw), pad                # separable running sums: one read of the map
        return F.avg_pool2d(x, (ph, pw), stride=1), pad

    def forward(self, x):
        fused = _is_fused_abn(self.map_bn) and _is_fused_abn(self.red_bn)
        if fused:
            # each branch is normalised on its own channel slice and lands in the shared
            # 1024-channel buffer that red_conv reads: no cat
            out = self.map_bn.forward_branches([m(x) for m in self.map_convs])
        else:
            out = self.map_bn(torch.cat([m(x) for m in self.map_convs], dim=1))
        out = self.red_conv(out)

        pool, pad = self._global_pooling(x)
        pool = self.pool_red_conv(self.global_pooling_bn(self.global_pooling_conv(pool)))
        if pad is not None and pool.shape[-2:] != (1, 1):
            pool = F.pad(pool, pad=pad, mode="replicate")
        if fused and pool.shape[-2:] == (1, 1):
            # per-(image, channel) bias folded into red_bn's statistics and apply passes
            return self.red_bn(out, plane_bias=pool)
        if pool.shape[-2:] == (1, 1):
            pool = pool.expand(-1, -1, x.size(2), x.size(3))
        out = out + pool
        return self.red_bn(out)
