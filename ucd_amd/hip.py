"""ctypes binding of ``libucd_hip.so`` (the C ABI declared in ``include/ucd_hip.h``).

PyTorch is plumbing here: it owns device memory and streams; every compute call below hands raw
device pointers and the current ``hipStream_t`` to the library.  There is no CPU fallback - if the
shared library is missing or a call fails this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libucd_hip.so")

F32, BF16 = 0, 1
ACT_IDENTITY, ACT_LEAKY_RELU, ACT_ELU = 0, 1, 2
ACT_CODES = {"identity": ACT_IDENTITY, "leaky_relu": ACT_LEAKY_RELU, "elu": ACT_ELU}
ACT_MASK = 0xFF
NORM_ABS_GAMMA = 0x100   # flag bit of `act`: gamma~ = |weight| + eps (InPlaceABN / InPlaceABNSync), see include/ucd_hip.h
PIX_TILE = 128          # kPixTile of csrc/pixcon.h
PIXCON_LD = 256         # feature rows of the contrast matrix are padded to 256 columns
PIXCON_F32, PIXCON_F16, PIXCON_F16_SPLIT = 0, 1, 2
PIXCON_PRECISION = {"f32": PIXCON_F32, "fp32": PIXCON_F32, "f16": PIXCON_F16, "fp16": PIXCON_F16,
                    "f16_split": PIXCON_F16_SPLIT}


class PixconMeta(C.Structure):
    """Mirror of ``ucd_pixcon_meta`` (include/ucd_hip.h)."""
    _fields_ = [("A", C.c_int32), ("Co", C.c_int32), ("min_new", C.c_int32), ("n_new", C.c_int32),
                ("Apad", C.c_int32), ("Cpad", C.c_int32), ("n_valid", C.c_int32), ("sorted", C.c_int32),
                ("label_start_a", C.c_int32 * 257), ("label_start_o", C.c_int32 * 257),
                ("label_count_a", C.c_int32 * 256), ("label_count_c", C.c_int32 * 256),
                ("reserved", C.c_int32 * 2)]


META_BYTES = C.sizeof(PixconMeta)


class Conv1x1Desc(C.Structure):
    """Mirror of ``ucd_conv1x1_desc`` (include/ucd_hip.h)."""
    _fields_ = [("a", C.c_void_p), ("lda", C.c_int), ("w", C.c_void_p), ("ldw", C.c_int), ("y", C.c_void_p), ("ldy", C.c_int),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("in_mean", C.c_void_p), ("in_scale", C.c_void_p), ("in_shift", C.c_void_p), ("in_act", C.c_int),
                ("in_slope", C.c_float), ("out_mode", C.c_int),
                ("out_mean", C.c_void_p), ("out_scale", C.c_void_p), ("out_shift", C.c_void_p), ("out_invstd", C.c_void_p),
                ("residual", C.c_void_p), ("ldr", C.c_int), ("out_act", C.c_int), ("out_slope", C.c_float),
                ("partial", C.c_void_p), ("accumulate", C.c_int),
                ("taps", C.c_int), ("H", C.c_int), ("W", C.c_int), ("dilation", C.c_int),
                ("side2", C.c_void_p), ("ld2", C.c_int), ("stride", C.c_int),
                ("stat_acc", C.c_void_p), ("stat_shift", C.c_void_p), ("stat_acc2", C.c_void_p), ("stat_rep", C.c_int)]


_p, _i, _f, _z = C.c_void_p, C.c_int, C.c_float, C.c_size_t
# name -> (restype, argtypes); must list every symbol of include/ucd_hip.h (checked by the CPU tests)
SIGNATURES = {
    "ucd_version": (_i, []),
    "ucd_last_error": (C.c_char_p, []),
    "ucd_abn_workspace_bytes": (_z, [_i, _i]),
    "ucd_abn_stats": (_i, [_p, _i, _i, _i, _i, _p, _i, _p, _p, _p, _z, _p]),
    "ucd_abn_stats_finalize": (_i, [_p, _i, _i, _i, _i, _p, _i, _p, _p, _p, _p, _p, _f, _f, _p, _p, _p, _i, _p, _z, _p]),
    "ucd_abn_finalize": (_i, [_p, _p, _f, _i, _p, _p, _p, _f, _f, _p, _p, _p, _i, _p]),
    "ucd_abn_eval_params": (_i, [_p, _p, _f, _i, _p, _p, _i, _p]),
    "ucd_abn_apply": (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _i, _p, _i, _p, _p, _p, _i, _f, _p]),
    "ucd_conv1x1_stat_replicas": (_i, [_i]),
    "ucd_fill_zero": (_i, [_p, _z, _p]),
    "ucd_abn_apply_stats": (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _p, _i, _p, _f, _p, _p, _p, _p, _f, _f, _p, _p, _p, _i, _f, _p]),
    "ucd_abn_bwd_apply_raw": (_i, [_p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _i, _p, _f, _i, _f, _p]),
    "ucd_abn_bwd_reduce": (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _i, _p, _i, _p, _p, _p, _p, _p, _i, _f, _p, _p, _z, _p]),
    "ucd_label_path": (_i, [_p, _p, _i, _i, _p, _p, _p, _p]),
    "ucd_image_path": (_i, [_p, _p, _i, _i, _i, _i, _f, _f, _f, _f, _f, _f, _p, _p, _p, _p]),
    "ucd_gemm_load": (_i, [C.c_char_p]),
    "ucd_gemm_workspace_bytes": (_z, []),
    "ucd_gemm_bf16": (_i, [_i, _i, _i, _i, _p, _i, _p, _i, _p, _i, _p, _z, _i, _p]),
    "ucd_gemm_bf16_acc": (_i, [_i, _i, _i, _i, _p, _i, _p, _i, _p, _i, _p, _z, _p]),
    "ucd_gemm_has_plan": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "ucd_gemm_last_tuned_us": (_f, []),
    "ucd_gemm_last_candidates": (_i, []),
    "ucd_comm_load": (_i, [C.c_char_p]),
    "ucd_comm_unique_id": (_i, [_p, _z]),
    "ucd_comm_init": (_i, [_p, _z, _i, _i, C.POINTER(C.c_void_p)]),
    "ucd_comm_destroy": (_i, [_p]),
    "ucd_comm_all_gather": (_i, [_p, _p, _p, _z, _p]),
    "ucd_comm_all_reduce_sum": (_i, [_p, _p, _z, _p]),
    "ucd_comm_init_local": (_i, [_i, _i, C.POINTER(C.c_void_p)]),
    "ucd_comm_ipc_handle_bytes": (_z, []),
    "ucd_comm_ipc_create": (_i, [_p, _i, _i, _p]),
    "ucd_comm_ipc_connect": (_i, [_p, _p]),
    "ucd_comm_ipc_drop": (_i, [_p]),
    "ucd_comm_ipc_timeouts": (C.c_uint, [_p]),
    "ucd_abn_sync_forward_comm": (_i, [_p, _i, _p, _i, _p, _i, _p, _i, _i, _i, _i, _p, _i, _p, _p, _p, _p, _f, _f, _p, _i, _f,
                                       _p, _z, _p]),
    "ucd_abn_sync_backward_comm": (_i, [_p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _i, _i, _i, _p, _i, _p, _p, _p, _p, _p,
                                        _p, _p, _i, _f, _p, _z, _p]),
    "ucd_abn_sync_stats": (_i, [_p, _i, _i, _i, _i, _p, _i, _p, _p, _p, _p, _z, _p]),
    "ucd_abn_sync_forward": (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _i, _p, _i, _p, _i, _p, _p, _p, _p, _f, _f, _p, _i, _f, _p]),
    "ucd_abn_sync_bwd_reduce": (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _i, _p, _i, _p, _p, _p, _p, _p, _i, _f, _p, _p, _p, _z, _p]),
    "ucd_abn_bwd_apply": (_i, [_p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _i, _i, _i, _p, _i, _p, _p, _p, _p, _p, _p,
                               _f, _i, _i, _f, _p]),
    "ucd_abn_forward": (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _i, _p, _i, _p, _p, _p, _p, _f, _f, _i, _p, _p, _i, _f, _p, _z, _p]),
    "ucd_abn_backward": (_i, [_p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _i, _i, _i, _p, _i, _p, _p, _p, _p, _p, _p, _f, _i, _i,
                              _i, _f, _p, _z, _p]),
    "ucd_plane_sum": (_i, [_p, _i, _i, _i, _i, _i, _f, _p, _p]),
    "ucd_window_mean_workspace_bytes": (_z, [_i, _i, _i, _i, _i]),
    "ucd_window_mean": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p, _z, _p]),
    "ucd_attmap_workspace_bytes": (_z, [_i, _i]),
    "ucd_attmap": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _p, _z, _p]),
    "ucd_conv1x1_row_tiles": (_i, [_i]),
    "ucd_conv1x1_stats_partial_bytes": (_z, [_i, _i]),
    "ucd_conv1x1": (_i, [C.POINTER(Conv1x1Desc), _p]),
    "ucd_conv1x1_stats_finalize": (_i, [_p, _i, _i, _p, _p, _p, _f, _f, _p, _p, _i, _p]),
    "ucd_abn_reduce_partials": (_i, [_p, _i, _i, _p, _p, _p, _i, _p]),
    "ucd_conv1x1_wgrad_workspace_bytes": (_z, [_i, _i, _i]),
    "ucd_conv1x1_wgrad": (_i, [_p, _i, _p, _i, _i, _i, _i, _p, _p, _p, _i, _f, _p, _p, _z, _p]),
    "ucd_abn_sync_finalize": (_i, [_p, _i, _i, _i, _p, _p, _p, _f, _f, _p, _i, _p]),
    "ucd_stem_pooled_size": (_i, [_i]),
    "ucd_stem_pool_workspace_bytes": (_z, [_i]),
    "ucd_stem_apply_pool": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _i, _f, _p, _p, _p]),
    "ucd_stem_pool_backward": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _f, _i, _f, _p, _p, _z, _i, _p]),
    "ucd_conv_wgrad_workspace_bytes": (_z, [_i, _i, _i, _i]),
    "ucd_conv_wgrad": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _i, _p, _z, _p]),
    "ucd_stem_conv7x7": (_i, [_p, C.c_longlong, C.c_longlong, C.c_longlong, C.c_longlong, _i, _i, _i, _p, _p, _p]),
    "ucd_stem_conv_pool": (_i, [_p, C.c_longlong, C.c_longlong, C.c_longlong, C.c_longlong, _i, _i, _i, _p, _p, _p, _p, _i, _f, _p, _p]),
    "ucd_conv_wgrad_strided": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _i, _p, _z, _p]),
    "ucd_conv_wgrad_ex": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _i, _p, _z, _i, _p]),
    "ucd_conv_wgrad_defer": (_i, [_i]),
    "ucd_conv_wgrad_mode": (_i, []),
    "ucd_conv_wgrad_flush": (_i, [_p]),
    "ucd_conv_wgrad_drop": (_i, [_p]),
    "ucd_transpose_bf16": (_i, [_p, _i, _i, _p, _p]),
    "ucd_flip_weights_batched": (_i, [_p, _p, _p, _i, _p, _p]),
    "ucd_flip_weights_batched64": (_i, [_p, _p, _p, _i, _p, _p]),
    "ucd_sgd_chunk": (_i, []),
    "ucd_sgd_step": (_i, [_p, _p, _i, _p, _p]),
    "ucd_sgd_hyper_store": (_i, [_p, _p, _p]),
    "ucd_sgd_step_dev": (_i, [_p, _p, _i, _p, _p]),
    "ucd_pixcon_prep_workspace_bytes": (_z, [_i, _i]),
    "ucd_pixcon_prep": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _z, _p]),
    "ucd_pixcon_gather": (_i, [_p, _i, _p, _i, _i, _i, _i, _p, _p, _p, _i, _p, _p, _i, _p, _i, _p, _p, _p, _p]),
    "ucd_pixcon_loss_workspace_bytes": (_z, [_i, _i, _i]),
    "ucd_pixcon_loss": (_i, [_p, _i, _i, _p, _p, _i, _i, _p, _p, _i, _p, _i, _f, _i, _i, _p, _p, _i, _p, _p, _z, _p]),
    "ucd_pixcon_loss_given_p": (_i, [_p, _i, _i, _p, _p, _i, _p, _i, _f, _i, _p, _p, _i, _p, _p, _z, _p]),
    "ucd_pixcon_scatter_grad": (_i, [_p, _p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "ucd_seg_confusion": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p]),
    "ucd_seg_losses_workspace_bytes": (_z, [_i, _i, _i]),
    "ucd_seg_losses": (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _p, _p, _i, _p, _z, _p]),
}

_lib = None


def load():
    """Load the library once; raise (never fall back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build the HIP kernels first (python -c 'import __graft_entry__ as g; "
                f"g.build()' or make -C ucd_amd/csrc).  ucd_amd has no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def _check(rc, name):
    if rc != 0:
        msg = load().ucd_last_error().decode(errors="replace")
        raise RuntimeError(f"{name} failed (code {rc}): {msg}")


# Optional per-call timing (bench.py's instrumented pass): HIP events recorded on the stream each
# library call is launched on, with the call's algorithmic work (bytes or flops).
_timing = None


def enable_call_timing():
    global _timing
    _timing = {}
    return _timing


def disable_call_timing():
    global _timing
    _timing = None


class _Timed:
    """with _timed(name, work): <library call>"""

    def __init__(self, name, work, flops=None, staged=None):
        # work: the call type's roofline unit (bytes for the HBM-bound types, flop for the MFMA-bound ones); flops: for the 1x1
        # products ALSO the flop count, so that bench.py can classify each call by min(bytes / 8 TB/s, flop / 2.5 PFLOP/s);
        # staged (round 6): bytes the GEMM's workgroups pull through their CUs' LDS-DMA path, M N K 2 (1 / BM + 1 / BN)
        self.name, self.work, self.flops, self.staged = name, work, flops, staged

    def __enter__(self):
        self.start = torch.cuda.Event(enable_timing=True)
        self.end = torch.cuda.Event(enable_timing=True)
        self.start.record(torch.cuda.current_stream())

    def __exit__(self, *exc):
        self.end.record(torch.cuda.current_stream())
        _timing.setdefault(self.name, []).append((self.start, self.end, self.work) if self.flops is None
                                                 else (self.start, self.end, self.work, self.flops, self.staged or 0))
        return False


class _NoTimer:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_TIMER = _NoTimer()


def _timed(name, work, flops=None, staged=None):
    return _NO_TIMER if _timing is None else _Timed(name, work, flops, staged)


def ptr(t):
    """Device address as a plain int (ctypes converts ints for c_void_p parameters; None is NULL)."""
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """hipStream_t of torch's current stream as an int.  torch.cuda.current_stream() builds a Python Stream
    object (~10 us); the raw getter is a plain C call."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def dtype_code(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"ucd_amd kernels take float32 or bfloat16 activations, got {t.dtype}")


_workspaces = {}


def workspace(nbytes, device, tag="abn"):
    """Grow-only scratch buffer per (device, tag, stream): calls on one stream are ordered, so they share it; work on
    another stream (the teacher running beside the student) gets its own."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), tag, stream())
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


# ---------------------------------------------------------------------------------------------
# activations as row matrices
# ---------------------------------------------------------------------------------------------
def rows_view(x):
    """[B, C, H, W] tensor -> (tensor, M, C, HW, ld): the channels-last row matrix behind it.  The
    tensor is returned unchanged when its strides already describe rows of C contiguous channels with
    a constant pitch (a channels_last tensor or a channel slice of one); otherwise it is copied."""
    B, Cc, H, W = x.shape
    st = x.stride()
    if st[1] == 1 and st[3] == Cc and st[2] == W * Cc and st[0] == H * W * Cc and W > 1 and H > 1 \
            and not (x.data_ptr() & 15) and not ((Cc * x.element_size()) & 15):
        return x, B * H * W, Cc, H * W, Cc                 # dense channels-last: the common case
    ok = _rows_ld(x)
    if ok is None:
        x = x.contiguous(memory_format=torch.channels_last)
        ok = _rows_ld(x)
        if ok is None:      # 1x1 maps etc. whose channels_last strides are ambiguous
            x = x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
            ok = Cc
    return x, B * H * W, Cc, H * W, ok


def _rows_ld(x):
    B, Cc, H, W = x.shape
    sb, sc, sh, sw = x.stride()
    if Cc > 1 and sc != 1:
        return None
    if W > 1:
        ld = sw
    elif H > 1:
        ld = sh
    elif B > 1:
        ld = sb
    else:
        ld = Cc
    if ld < Cc:
        return None
    if W > 1 and H > 1 and sh != W * ld:
        return None
    if B > 1 and (H > 1 or W > 1) and sb != H * W * ld:
        return None
    if B > 1 and H == 1 and W == 1 and sb != ld:
        return None
    es = x.element_size()
    if (x.data_ptr() % 16) or (ld * es) % 16:
        return None
    return ld


def empty_like_rows(x, channels=None, dtype=None):
    """New channels_last [B, C', H, W] tensor."""
    B, Cc, H, W = x.shape
    return torch.empty((B, channels or Cc, H, W), dtype=dtype or x.dtype, device=x.device,
                       memory_format=torch.channels_last)


# ---------------------------------------------------------------------------------------------
# thin wrappers (no autograd)
# ---------------------------------------------------------------------------------------------
def abn_stats(x, ld, M, Cc, plane_bias, HW, sums, kshift):
    lib = load()
    nbytes = lib.ucd_abn_workspace_bytes(M, Cc)
    ws = workspace(nbytes, x.device)
    with _timed("ucd_abn_stats", M * Cc * x.element_size()):
        _check(lib.ucd_abn_stats(ptr(x), ld, dtype_code(x), M, Cc, ptr(plane_bias), HW, ptr(sums), ptr(kshift),
                                 ptr(ws), nbytes, stream()), "ucd_abn_stats")


def abn_stats_finalize(x, ld, M, Cc, plane_bias, HW, sums, kshift, weight, running_mean, running_var, momentum, eps,
                       mean, invstd, scale, flags=0):
    lib = load()
    nbytes = lib.ucd_abn_workspace_bytes(M, Cc)
    ws = workspace(nbytes, x.device)
    with _timed("ucd_abn_stats", M * Cc * x.element_size()):
        _check(lib.ucd_abn_stats_finalize(ptr(x), ld, dtype_code(x), M, Cc, ptr(plane_bias), HW, ptr(sums), ptr(kshift),
                                          ptr(weight), ptr(running_mean), ptr(running_var), float(momentum), float(eps),
                                          ptr(mean), ptr(invstd), ptr(scale), int(flags) & NORM_ABS_GAMMA, ptr(ws), nbytes,
                                          stream()), "ucd_abn_stats_finalize")


def abn_forward(x, ld_x, y, ld_y, residual, ld_r, M, Cc, plane_bias, HW, weight, bias, running_mean, running_var,
                momentum, eps, training, buf, eval_consts, act, slope):
    lib = load()
    nbytes = lib.ucd_abn_workspace_bytes(M, Cc)
    ws = workspace(nbytes, x.device)
    es = x.element_size()
    if _timing is not None:      # instrumented bench pass: keep the per-kernel attribution
        if training:
            abn_stats_finalize(x, ld_x, M, Cc, plane_bias, HW, buf[:2 * Cc], buf[2 * Cc:3 * Cc], weight, running_mean,
                               running_var, momentum, eps, buf[3 * Cc:4 * Cc], buf[4 * Cc:5 * Cc], buf[5 * Cc:], act)
            abn_apply(x, ld_x, y, ld_y, residual, ld_r, M, Cc, plane_bias, HW, buf[3 * Cc:4 * Cc], buf[5 * Cc:], bias, act, slope)
        else:
            sc = eval_consts[1] if eval_consts is not None else buf[5 * Cc:]
            if eval_consts is None:
                abn_eval_params(weight, running_var, eps, Cc, buf[4 * Cc:5 * Cc], buf[5 * Cc:], act)
            abn_apply(x, ld_x, y, ld_y, residual, ld_r, M, Cc, plane_bias, HW, running_mean, sc, bias, act, slope)
        return
    rc = lib.ucd_abn_forward(x.data_ptr(), ld_x, y.data_ptr(), ld_y, ptr(residual), ld_r, BF16 if es == 2 else F32, M, Cc,
                             ptr(plane_bias), HW, ptr(weight), ptr(bias), running_mean.data_ptr(), running_var.data_ptr(),
                             momentum, eps, 1 if training else 0, ptr(buf), ptr(eval_consts), act, slope, ws.data_ptr(),
                             nbytes, stream())
    if rc:
        _check(rc, "ucd_abn_forward")


def abn_backward(x, ld_x, dy, ld_dy, y, ld_y, dx, ld_dx, dz, ld_dz, M, Cc, plane_bias, HW, mean, invstd, scale, bias, weight,
                 sums, count, training, need_sums, act, slope):
    lib = load()
    if _timing is not None:
        if training or need_sums:
            abn_bwd_reduce(x, ld_x, dy, ld_dy, y, ld_y, M, Cc, plane_bias, HW, mean, invstd, scale, bias, act, slope, sums,
                           weight)
        abn_bwd_apply(x, ld_x, dy, ld_dy, y, ld_y, dx, ld_dx, dz, ld_dz, M, Cc, plane_bias, HW, mean, invstd, scale, bias,
                      weight, sums, count, 0 if training else 1, act, slope)
        return
    nbytes = lib.ucd_abn_workspace_bytes(M, Cc)
    ws = workspace(nbytes, x.device)
    rc = lib.ucd_abn_backward(x.data_ptr(), ld_x, dy.data_ptr(), ld_dy, ptr(y), ld_y, dx.data_ptr(), ld_dx, ptr(dz), ld_dz,
                              BF16 if x.element_size() == 2 else F32, M, Cc, ptr(plane_bias), HW, mean.data_ptr(),
                              invstd.data_ptr(), scale.data_ptr(), ptr(bias), ptr(weight), sums.data_ptr(), count,
                              1 if training else 0, 1 if need_sums else 0, act, slope, ws.data_ptr(), nbytes, stream())
    if rc:
        _check(rc, "ucd_abn_backward")


_gemm_ready = None


def _loaded_library_path(fragment):
    """Path of a shared object this process already has mapped (PyTorch-ROCm ships its own hipBLASLt / RCCL; a second
    instance of either must not be mixed in)."""
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if fragment in line:
                    return line.split()[-1]
    except OSError:
        pass
    return ""


def gemm_available():
    global _gemm_ready
    if _gemm_ready is None:
        try:
            _check(load().ucd_gemm_load(_loaded_library_path("libhipblaslt").encode()), "ucd_gemm_load")
            _gemm_ready = True
        except (RuntimeError, ImportError, OSError):
            _gemm_ready = False
    return _gemm_ready


def gemm_bf16(mode, a, b, out, tune=True):
    """Row-major bf16 GEMM into ``out`` (see ucd_gemm_bf16): mode 0 a[M,K] b[N,K]^T, 1 a[M,K] b[K,N], 2 a[K,M]^T b[K,N]."""
    lib = load()
    M, N = out.shape
    K = a.shape[0] if mode == 2 else a.shape[1]
    nbytes = lib.ucd_gemm_workspace_bytes()
    ws = workspace(nbytes, out.device, "gemm")
    _check(lib.ucd_gemm_bf16(mode, M, N, K, ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(out), out.stride(0), ptr(ws), nbytes,
                             1 if tune else 0, stream()), "ucd_gemm_bf16")
    return out


def abn_sync_forward_comm(comm, world, x, ld_x, y, ld_y, residual, ld_r, M, Cc, plane_bias, HW, weight, bias, running_mean,
                          running_var, momentum, eps, buf, act, slope):
    """Whole SyncBN forward of a layer (statistics, all-gather on the current stream, combination, apply)."""
    lib = load()
    nbytes = lib.ucd_abn_workspace_bytes(M, Cc)
    ws = workspace(nbytes, x.device)
    _check(lib.ucd_abn_sync_forward_comm(comm, world, ptr(x), ld_x, ptr(y), ld_y, ptr(residual), ld_r, dtype_code(x), M, Cc,
                                         ptr(plane_bias), HW, ptr(weight), ptr(bias), ptr(running_mean), ptr(running_var),
                                         float(momentum), float(eps), ptr(buf), act, float(slope), ptr(ws), nbytes, stream()),
           "ucd_abn_sync_forward_comm")


def abn_sync_backward_comm(comm, world, x, ld_x, dy, ld_dy, y, ld_y, dx, ld_dx, dz, ld_dz, M, Cc, plane_bias, HW, mean, invstd,
                           scale, bias, weight, sums4, act, slope):
    lib = load()
    nbytes = lib.ucd_abn_workspace_bytes(M, Cc)
    ws = workspace(nbytes, x.device)
    _check(lib.ucd_abn_sync_backward_comm(comm, world, ptr(x), ld_x, ptr(dy), ld_dy, ptr(y), ld_y, ptr(dx), ld_dx, ptr(dz),
                                          ld_dz, dtype_code(x), M, Cc, ptr(plane_bias), HW, ptr(mean), ptr(invstd), ptr(scale),
                                          ptr(bias), ptr(weight), ptr(sums4), sums4.data_ptr() + 8 * Cc, act, float(slope),
                                          ptr(ws), nbytes, stream()),
           "ucd_abn_sync_backward_comm")


def abn_sync_stats(x, ld, M, Cc, plane_bias, HW, sums, kshift, pack):
    lib = load()
    nbytes = lib.ucd_abn_workspace_bytes(M, Cc)
    ws = workspace(nbytes, x.device)
    with _timed("ucd_abn_stats", M * Cc * x.element_size()):
        _check(lib.ucd_abn_sync_stats(ptr(x), ld, dtype_code(x), M, Cc, ptr(plane_bias), HW, ptr(sums), ptr(kshift),
                                      ptr(pack), ptr(ws), nbytes, stream()), "ucd_abn_sync_stats")


def abn_sync_forward(x, ld_x, y, ld_y, residual, ld_r, M, Cc, plane_bias, HW, gathered, world, weight, bias, running_mean,
                     running_var, momentum, eps, buf, act, slope):
    with _timed("ucd_abn_apply", M * Cc * x.element_size() * (2 + (residual is not None))):
        _check(load().ucd_abn_sync_forward(ptr(x), ld_x, ptr(y), ld_y, ptr(residual), ld_r, dtype_code(x), M, Cc,
                                           ptr(plane_bias), HW, ptr(gathered), world, ptr(weight), ptr(bias),
                                           ptr(running_mean), ptr(running_var), float(momentum), float(eps), ptr(buf),
                                           act, float(slope), stream()), "ucd_abn_sync_forward")


def abn_sync_bwd_reduce(x, ld_x, dy, ld_dy, y, ld_y, M, Cc, plane_bias, HW, mean, invstd, scale, shift, act, slope, sums,
                        local_sums, weight=None):
    lib = load()
    nbytes = lib.ucd_abn_workspace_bytes(M, Cc)
    ws = workspace(nbytes, x.device)
    with _timed("ucd_abn_bwd_reduce", M * Cc * x.element_size() * (2 + (y is not None))):
        _check(lib.ucd_abn_sync_bwd_reduce(ptr(x), ld_x, ptr(dy), ld_dy, ptr(y), ld_y, dtype_code(x), M, Cc,
                                           ptr(plane_bias), HW, ptr(mean), ptr(invstd), ptr(scale), ptr(shift), ptr(weight),
                                           act, float(slope), ptr(sums), ptr(local_sums), ptr(ws), nbytes, stream()),
               "ucd_abn_sync_bwd_reduce")


def abn_finalize(sums, kshift, count, Cc, weight, running_mean, running_var, momentum, eps, mean, invstd, scale, flags=0):
    _check(load().ucd_abn_finalize(ptr(sums), ptr(kshift), float(count), Cc, ptr(weight), ptr(running_mean),
                                   ptr(running_var), float(momentum), float(eps), ptr(mean), ptr(invstd), ptr(scale),
                                   int(flags) & NORM_ABS_GAMMA, stream()), "ucd_abn_finalize")


def abn_eval_params(weight, running_var, eps, Cc, invstd, scale, flags=0):
    _check(load().ucd_abn_eval_params(ptr(weight), ptr(running_var), float(eps), Cc, ptr(invstd), ptr(scale),
                                      int(flags) & NORM_ABS_GAMMA, stream()), "ucd_abn_eval_params")


def abn_apply(x, ld_x, y, ld_y, residual, ld_r, M, Cc, plane_bias, HW, mean, scale, shift, act, slope):
    with _timed("ucd_abn_apply", M * Cc * x.element_size() * (2 + (residual is not None))):
        _check(load().ucd_abn_apply(ptr(x), ld_x, ptr(y), ld_y, ptr(residual), ld_r, dtype_code(x), M, Cc,
                                    ptr(plane_bias), HW, ptr(mean), ptr(scale), ptr(shift), act, float(slope),
                                    stream()), "ucd_abn_apply")


def abn_bwd_reduce(x, ld_x, dy, ld_dy, y, ld_y, M, Cc, plane_bias, HW, mean, invstd, scale, shift, act, slope, sums,
                   weight=None):
    lib = load()
    nbytes = lib.ucd_abn_workspace_bytes(M, Cc)
    ws = workspace(nbytes, x.device)
    with _timed("ucd_abn_bwd_reduce", M * Cc * x.element_size() * (2 + (y is not None))):
        _check(lib.ucd_abn_bwd_reduce(ptr(x), ld_x, ptr(dy), ld_dy, ptr(y), ld_y, dtype_code(x), M, Cc,
                                      ptr(plane_bias), HW, ptr(mean), ptr(invstd), ptr(scale), ptr(shift), ptr(weight),
                                      act, float(slope), ptr(sums), ptr(ws), nbytes, stream()), "ucd_abn_bwd_reduce")


def abn_bwd_apply(x, ld_x, dy, ld_dy, y, ld_y, dx, ld_dx, dz, ld_dz, M, Cc, plane_bias, HW, mean, invstd, scale,
                  shift, weight, sums, count, frozen, act, slope):
    with _timed("ucd_abn_bwd_apply", M * Cc * x.element_size() * (3 + (y is not None) + (dz is not None))):
        _check(load().ucd_abn_bwd_apply(ptr(x), ld_x, ptr(dy), ld_dy, ptr(y), ld_y, ptr(dx), ld_dx, ptr(dz), ld_dz,
                                        dtype_code(x), M, Cc, ptr(plane_bias), HW, ptr(mean), ptr(invstd),
                                        ptr(scale), ptr(shift), ptr(weight), ptr(sums), float(count), int(frozen),
                                        act, float(slope), stream()), "ucd_abn_bwd_apply")


def plane_sum(x, ld, B, HW, Cc, alpha, out):
    with _timed("ucd_plane_sum", B * HW * Cc * x.element_size()):
        _check(load().ucd_plane_sum(ptr(x), ld, dtype_code(x), B, HW, Cc, float(alpha), ptr(out), stream()),
               "ucd_plane_sum")


def attmap(x, ld_x, y, ld_y, B, HW, Cc):
    lib = load()
    nbytes = lib.ucd_attmap_workspace_bytes(B, HW)
    ws = workspace(nbytes, x.device, "attmap")
    _check(lib.ucd_attmap(ptr(x), ld_x, ptr(y), ld_y, dtype_code(x), B, HW, Cc, ptr(ws), nbytes, stream()),
           "ucd_attmap")


# ---------------------------------------------------------------------------------------------
# 1x1 convolutions as fused GEMMs (csrc/conv1x1.hip)
# ---------------------------------------------------------------------------------------------
def conv1x1(a, w, y, in_norm=None, out_mode=0, out_norm=None, residual=None, partial=None, accumulate=False, conv3=None,
            side2=None, strided=None, stat_acc=None, stat_shift=None, stat_acc2=None, stat_rep=1):
    """y[M, N] = out(in(a)[M, K] . w[N, K]^T).  ``in_norm`` = (mean, scale, shift, act, slope) of the producer's ABN or None;
    ``out_norm`` = (mean, scale, shift, invstd, act, slope) for out_mode 1 / 3.  All 2-D bf16 row matrices.
    ``conv3 = (H, W, dilation)``: 3x3 convolution (stride 1, padding = dilation) of the [B, H, W, K] map behind ``a`` with the
    channels-last weight ``w`` given as its [N, 9 K] row matrix; ``conv3 = (H, W, dilation, stride)`` / ``strided = (H, W, stride)``
    (1x1): the strided layers, ``a`` the [B*H*W, K] rows of the input map and ``y`` the [B*OH*OW, N] rows of the output."""
    lib = load()
    d = Conv1x1Desc()
    M, K = y.shape[0], a.shape[1]           # strided: a holds the (larger) input map, y has the product's rows
    N = w.shape[0]
    d.a, d.lda, d.w, d.ldw, d.y, d.ldy = a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), y.data_ptr(), y.stride(0)
    d.M, d.N, d.K = M, N, K
    if conv3 is not None:
        d.taps, d.H, d.W, d.dilation = 9, int(conv3[0]), int(conv3[1]), int(conv3[2])
        if len(conv3) > 3:
            d.stride = int(conv3[3])
    elif strided is not None:
        d.H, d.W, d.stride = int(strided[0]), int(strided[1]), int(strided[2])
    if in_norm is not None:
        mean, scale, shift, act, slope = in_norm
        d.in_mean, d.in_scale, d.in_shift, d.in_act, d.in_slope = ptr(mean), ptr(scale), ptr(shift), act & ACT_MASK, slope
    d.out_mode = out_mode
    if out_norm is not None:
        mean, scale, shift, invstd, act, slope = out_norm
        d.out_mean, d.out_scale, d.out_shift, d.out_invstd = ptr(mean), ptr(scale), ptr(shift), ptr(invstd)
        d.out_act, d.out_slope = act & ACT_MASK, slope
    if residual is not None:
        d.residual, d.ldr = residual.data_ptr(), residual.stride(0)
    if side2 is not None:           # out_mode 4: the conv output z of the block whose output is ``residual``
        d.side2, d.ld2 = side2.data_ptr(), side2.stride(0)
    d.partial = ptr(partial)
    d.accumulate = 1 if accumulate else 0
    # atomic statistics (round 5): column sums into a zeroed [2 N] accumulator instead of per-tile partial rows
    d.stat_acc, d.stat_shift, d.stat_acc2, d.stat_rep = ptr(stat_acc), ptr(stat_shift), ptr(stat_acc2), int(stat_rep)
    # roofline work of the instrumented bench pass: algorithmic bytes for the 1x1 products (HBM-bound at the network's
    # shapes), flop for the 3x3 implicit GEMM (9 K deep: MFMA-bound)
    work = 2 * M * 9 * K * N if conv3 is not None else 2 * (M * K + M * N * (1 + (residual is not None) + bool(accumulate)
                                                                                 + (side2 is not None)))
    # bytes staged through LDS by the tiled kernel: every (BM x BN) tile pulls (BM + BN) K values per tap; BN as the library picks it
    bn = 128 if (N % 128 == 0 and ((M + 127) // 128) * (N // 128) > 128) else 64
    staged = 2 * M * N * K * (9 if conv3 is not None else 1) * (1.0 / 128 + 1.0 / bn)
    with _timed("ucd_conv3x3" if conv3 is not None else "ucd_conv1x1", work, None if conv3 is not None else 2 * M * K * N, staged):
        _check(lib.ucd_conv1x1(C.byref(d), stream()), "ucd_conv1x1")
    return y


def abn_apply_stats(x, y, residual, M, Cc, acc, kshift, count, weight, bias, running_mean, running_var, momentum, eps, buf, act,
                    slope, reps=1):
    """y = act(norm(x) [+ residual]) with the statistics finalised in the kernel's prologue from the raw sums ``acc`` about
    ``kshift`` (ucd_abn_apply_stats); mean / invstd / scale land in ``buf[3C:6C]`` like ucd_conv1x1_stats_finalize's."""
    with _timed("ucd_abn_apply", M * Cc * x.element_size() * (2 + (residual is not None))):
        _check(load().ucd_abn_apply_stats(ptr(x), Cc, ptr(y), Cc, ptr(residual), Cc if residual is not None else 0, M, Cc, ptr(acc),
                                          int(reps), ptr(kshift), float(count), ptr(weight), ptr(bias), ptr(running_mean), ptr(running_var),
                                          float(momentum), float(eps), ptr(buf[3 * Cc:]), ptr(buf[4 * Cc:]), ptr(buf[5 * Cc:]), act,
                                          float(slope), stream()), "ucd_abn_apply_stats")
    return y


def abn_bwd_apply_raw(x, dy, y, dx, dz, M, Cc, mean, invstd, scale, shift, weight, sums, grad_sums, grad_out, count, act, slope,
                      reps=1):
    """dx (and dz) from RAW link sums, the parameter gradients written by the same launch (ucd_abn_bwd_apply_raw)."""
    with _timed("ucd_abn_bwd_apply", M * Cc * x.element_size() * (3 + (y is not None) + (dz is not None))):
        _check(load().ucd_abn_bwd_apply_raw(ptr(x), Cc, ptr(dy), Cc, ptr(y), Cc if y is not None else 0, ptr(dx), Cc, ptr(dz),
                                            Cc if dz is not None else 0, M, Cc, ptr(mean), ptr(invstd), ptr(scale), ptr(shift),
                                            ptr(weight), ptr(sums), ptr(grad_sums), int(reps), ptr(grad_out), float(count), act, float(slope),
                                            stream()), "ucd_abn_bwd_apply_raw")
    return dx


def conv1x1_wgrad(dy, a, dw, in_norm=None):
    """dw[N, K] = dy[M, N]^T . in(a)[M, K] (bf16)."""
    lib = load()
    M, N = dy.shape
    K = a.shape[1]
    nbytes = lib.ucd_conv1x1_wgrad_workspace_bytes(M, N, K)
    ws = workspace(nbytes, dy.device, "wgrad")
    mean = scale = shift = None
    act, slope = ACT_IDENTITY, 1.0
    if in_norm is not None:
        mean, scale, shift, act, slope = in_norm
    with _timed("ucd_conv1x1_wgrad", 2 * (M * N + M * K)):
        _check(lib.ucd_conv1x1_wgrad(ptr(dy), dy.stride(0), ptr(a), a.stride(0), M, N, K, ptr(mean), ptr(scale), ptr(shift),
                                     act & ACT_MASK, float(slope), ptr(dw), ptr(ws), nbytes, stream()), "ucd_conv1x1_wgrad")
    return dw


def abn_sync_finalize(gathered, world, M, Cc, weight, running_mean, running_var, momentum, eps, buf, flags=0):
    _check(load().ucd_abn_sync_finalize(ptr(gathered), world, M, Cc, ptr(weight), ptr(running_mean), ptr(running_var),
                                        float(momentum), float(eps), ptr(buf), int(flags) & NORM_ABS_GAMMA, stream()),
           "ucd_abn_sync_finalize")


def stem_conv7x7(x, w):
    """The stem's convolution (ucd_stem_conv7x7): ``x`` fp32 [B, 3, H, W] in any memory format, ``w`` the bf16 [64, 3, 7, 7]
    weight in channels-last memory order -> z [B, 64, OH, OW] dense channels-last bf16."""
    lib = load()
    B, Cc, H, W = x.shape
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    z = torch.empty((B, 64, OH, OW), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    sb, sc, sh, sw = x.stride()
    with _timed("ucd_stem_conv7x7", x.numel() * 4 + z.numel() * 2):
        _check(lib.ucd_stem_conv7x7(ptr(x), sb, sc, sh, sw, B, H, W, ptr(w), ptr(z), stream()), "ucd_stem_conv7x7")
    return z


def stem_conv_pool(x, w, mean, scale, beta, act, slope):
    """conv1 + frozen norm + activation + 3x3 / 2 max pool of the stem in one kernel (ucd_stem_conv_pool): ``x`` fp32 [B, 3, H, W],
    ``w`` bf16 [64, 3, 7, 7] channels-last; returns the pooled bf16 map [B, 64, PH, PW] (channels-last)."""
    lib = load()
    B, _, H, W = x.shape
    OH, OW = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    PH, PW = lib.ucd_stem_pooled_size(OH), lib.ucd_stem_pooled_size(OW)
    out = torch.empty((B, 64, PH, PW), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    sb, sc, sh, sw = x.stride()
    with _timed("ucd_stem_conv7x7", x.numel() * 4 + out.numel() * 2):
        _check(lib.ucd_stem_conv_pool(ptr(x), sb, sc, sh, sw, B, H, W, ptr(w), ptr(mean), ptr(scale), ptr(beta), act, float(slope),
                                      ptr(out), stream()), "ucd_stem_conv_pool")
    return out


def stem_apply_pool(z, mean, scale, beta, act, slope, want_idx):
    """max_pool2d(abn_apply(z), 3, 2, 1) in one pass (ucd_stem_apply_pool); ``z`` [B, C, H, W] dense channels-last bf16.
    Returns (pooled map, uint8 index map or None)."""
    lib = load()
    B, Cc, H, W = z.shape
    PH, PW = lib.ucd_stem_pooled_size(H), lib.ucd_stem_pooled_size(W)
    out = torch.empty((B, Cc, PH, PW), dtype=z.dtype, device=z.device, memory_format=torch.channels_last)
    idx = torch.empty((B, PH, PW, Cc), dtype=torch.uint8, device=z.device) if want_idx else None
    with _timed("ucd_stem_apply_pool", (z.numel() + out.numel()) * 2 + (idx.numel() if idx is not None else 0)):
        _check(lib.ucd_stem_apply_pool(ptr(z), B, H, W, Cc, ptr(mean), ptr(scale), ptr(beta), act, float(slope), ptr(out), ptr(idx),
                                       stream()), "ucd_stem_apply_pool")
    return out, idx


def stem_pool_backward(z, dpool, idx, mean, invstd, scale, beta, weight, sums, count, act, slope, dz, phase):
    lib = load()
    B, Cc, H, W = z.shape
    nbytes = lib.ucd_stem_pool_workspace_bytes(Cc)
    ws = workspace(nbytes, z.device)
    work = (z.numel() * (2 if phase & 1 else 0) + (2 * z.numel() * 2 if phase & 2 else 0) + dpool.numel() * 3 * (1 + (phase == 3)))
    with _timed("ucd_stem_pool_backward", work):
        _check(lib.ucd_stem_pool_backward(ptr(z), ptr(dpool), ptr(idx), B, H, W, Cc, ptr(mean), ptr(invstd), ptr(scale), ptr(beta),
                                          ptr(weight), ptr(sums), float(count), act, float(slope), ptr(dz), ptr(ws), nbytes,
                                          int(phase), stream()), "ucd_stem_pool_backward")


def wgrad_defer(mode):
    """Mode of the weight-gradient calls that allow it (``ucd_conv_wgrad_defer``): bit 0 (or True) defers their slab sums into the
    next launch, bit 1 moves them to the library's side stream until ``wgrad_flush``; returns the previous mode."""
    return int(load().ucd_conv_wgrad_defer(int(mode) & 3))


def _wgrad_side_release():
    # the C++ nodes hold the operands of their side-stream calls until the join (csrc/abn_node.cpp: g_side_hold)
    from . import abn as _abn
    node = _abn._abn_node()
    if node is not None and hasattr(node, "wgrad_side_release"):
        node.wgrad_side_release()


def wgrad_flush():
    """Launch the pending slab sum of the current stream, if any, and join the side stream of the weight gradients back into it
    (``ucd_conv_wgrad_flush``)."""
    try:
        _check(load().ucd_conv_wgrad_flush(stream()), "ucd_conv_wgrad_flush")
    finally:
        _wgrad_side_release()


def wgrad_drop():
    """Forget the pending slab sum of the current stream (an aborted backward); side-stream work already launched is joined."""
    try:
        load().ucd_conv_wgrad_drop(stream())
    finally:
        _wgrad_side_release()


_wgrad_turn = {}


def conv_wgrad(dz, x, dw=None, conv3=None, dw32=None, accumulate32=False, strided=None, defer=False, side=False):
    """Weight gradient of a stride-1 convolution (ucd_conv_wgrad): ``dz`` [M, N] and ``x`` [M, K] bf16 row matrices ->
    ``dw`` [N, taps * K] bf16 (channels-last weight order [N][kh][kw][K]) and / or ``dw32`` fp32 (+= when ``accumulate32``);
    ``conv3 = (H, W, dilation)`` selects the 3x3 form over the [B, H, W, K] map behind ``x``; ``conv3 = (H, W, dilation, stride)`` /
    ``strided = (H, W, stride)`` the strided layers (``x`` the input map's rows, ``dz`` the smaller output map's)."""
    lib = load()
    M, N = dz.shape
    K = x.shape[1]
    taps = 9 if conv3 is not None else 1
    H, W, d = (int(conv3[0]), int(conv3[1]), int(conv3[2])) if conv3 is not None else (0, 0, 1)
    stride = 1
    if conv3 is not None and len(conv3) > 3:
        stride = int(conv3[3])
    elif strided is not None:              # 1x1 with a stride: (H, W, stride) of the input map behind x
        H, W, stride = int(strided[0]), int(strided[1]), int(strided[2])
    nbytes = lib.ucd_conv_wgrad_workspace_bytes(M, N, K, taps)
    # ``defer``: the caller does not read the gradient before the next weight-gradient call or wgrad_flush() - the library may then
    # carry this call's slab sum in the next launch (two slab buffers in turn: a call's slabs outlive the next call)
    # ``side`` (with ``defer``): the call may also leave the current stream for the library's side stream until wgrad_flush() - the
    # caller keeps dz, x and the gradient alive and untouched until then (slab buffers of their own: never shared across streams)
    side = bool(side and defer)
    key = (dz.device.index, stream(), side)
    _wgrad_turn[key] = _wgrad_turn.get(key, 0) ^ 1
    ws = workspace(nbytes, dz.device, "wgrad%s%d" % ("side" if side else "", _wgrad_turn[key]))
    # MFMA-bound for the 3x3 layers, HBM / L2-bound for the 1x1 layers: flop for one, algorithmic bytes for the other
    work = 2 * M * 9 * K * N if conv3 is not None else 2 * (M * N + M * K)
    with _timed("ucd_conv3x3_wgrad" if conv3 is not None else "ucd_conv1x1_wgrad", work):
        _check(lib.ucd_conv_wgrad_ex(ptr(dz), dz.stride(0), ptr(x), x.stride(0), M, N, K, taps, H, W, d, stride, ptr(dw),
                                     ptr(dw32), 1 if accumulate32 else 0, ptr(ws), nbytes, (1 if defer else 0) | (2 if side else 0), stream()),
               "ucd_conv_wgrad")
    return dw if dw is not None else dw32


def transpose_bf16(src, dst):
    _check(load().ucd_transpose_bf16(ptr(src), src.shape[0], src.shape[1], ptr(dst), stream()), "ucd_transpose_bf16")
    return dst


def window_mean(x, ph, pw):
    """F.avg_pool2d(x, (ph, pw), stride=1) of a channels-last map (ucd_window_mean)."""
    lib = load()
    xv, M, Cc, HW, ld = rows_view(x)
    B, _, H, W = xv.shape
    out = torch.empty((B, Cc, H - ph + 1, W - pw + 1), dtype=xv.dtype, device=xv.device, memory_format=torch.channels_last)
    if out.shape[2] == 1 or out.shape[3] == 1 or B == 1:      # ambiguous channels_last strides: make the rows explicit
        out = torch.empty((B, H - ph + 1, W - pw + 1, Cc), dtype=xv.dtype, device=xv.device).permute(0, 3, 1, 2)
    nbytes = lib.ucd_window_mean_workspace_bytes(B, H, W, Cc, ph)
    ws = workspace(nbytes, xv.device, "winmean")
    with _timed("ucd_window_mean", B * H * W * Cc * xv.element_size()):
        _check(lib.ucd_window_mean(ptr(xv), ld, dtype_code(xv), B, H, W, Cc, ph, pw, ptr(out), Cc, ptr(ws), nbytes, stream()),
               "ucd_window_mean")
    return out


def conv1x1_stats_finalize(partial, M, Cc, weight, running_mean, running_var, momentum, eps, buf, pack=None, flags=0):
    _check(load().ucd_conv1x1_stats_finalize(ptr(partial), M, Cc, ptr(weight), ptr(running_mean), ptr(running_var),
                                             float(momentum), float(eps), ptr(buf), ptr(pack), int(flags) & NORM_ABS_GAMMA,
                                             stream()), "ucd_conv1x1_stats_finalize")


def conv1x1_row_tiles(M):
    return load().ucd_conv1x1_row_tiles(M)


def conv1x1_stats_partial(M, Cc, device):
    """Float buffer for the out_mode-2 partials of an [M, Cc] product (a triple per row tile and channel)."""
    return torch.empty(load().ucd_conv1x1_stats_partial_bytes(M, Cc) // 4, dtype=torch.float32, device=device)
