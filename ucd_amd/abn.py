"""ABN / InPlaceABN / InPlaceABNSync on the HIP kernels of ``libucd_hip.so``.

Drop-in for the three classes the reference imports from the third-party ``inplace_abn`` wheel
(``segmentation_module.py:5-6,15-20``): same constructor, same attributes (``weight``, ``bias``,
``running_mean``, ``running_var``, mutable ``activation`` / ``activation_param``) and the same
``state_dict`` keys, so the reference's pretrained files and step checkpoints load unchanged.

MI355X design notes
  * activations are channels-last ([B*H*W, C] rows; bf16 or fp32), statistics and parameters fp32;
  * the "in-place" of inplace_abn is a memory trick for 16 GB cards: with 288 GB of HBM the training
    forward keeps the pre-norm tensor and recomputes xhat from it in the backward (exact in bf16,
    no division by gamma), at the same HBM traffic (2 reads per pass).  Under ``torch.no_grad()``
    (the frozen teacher) InPlaceABN / InPlaceABNSync do overwrite their input, as their contract says;
  * the block epilogue ``+ residual -> leaky_relu`` and the ASPP ``cat`` / ``+= pool`` ride in the
    same passes (``forward(..., residual=)``, ``forward_branches``, ``forward(..., plane_bias=)``);
  * InPlaceABNSync exchanges one packed ``(mean_r, M2_r)`` vector per layer in the forward (all-gather, combined with
    Chan's formula on the device) and ``[sum_dz, sum_dz_xhat]`` in the backward (all-reduce) - over a library-owned RCCL
    communicator on the compute stream inside the layer's library call (``ucd_amd/comm.py``), or ``torch.distributed``
    when that is unavailable; without an initialised process group it is InPlaceABN;
  * the student's training-mode layers run through a C++ autograd node (``csrc/abn_node.cpp``) when it is built; the
    Python ``_ABNFunction`` below is the complete implementation and the fallback.
Semantics: biased batch variance, unbiased running variance, eps 1e-5, momentum 0.1, then ``leaky_relu`` / ``elu`` /
identity.  ``ABN`` is ``F.batch_norm`` (raw gamma), as in the wheel.  ``InPlaceABN`` / ``InPlaceABNSync`` normalise with
``gamma~ = |gamma| + eps`` like the wheel's in-place kernels (inplace-abn 1.0.x, the published forward ``(x - mean) *
rsqrt(var + eps) * (abs(weight) + eps) + bias`` and backward ``d weight = sign(weight) * sum dz*xhat``; the wheel's
source is not in the reference tree, so this restates its published algorithm - SURVEY.md section 8-a5): a reference
checkpoint with negative gammas therefore gives the same activations here as there.
"""
from __future__ import annotations

import os

from . import switches as _switches

import torch
import torch.distributed as dist
import torch.nn as nn
from torch.optim.optimizer import register_optimizer_step_post_hook

from . import hip
from .comm import direct_comm


def _act_code(name):
    try:
        return hip.ACT_CODES[name]
    except KeyError:
        raise RuntimeError(f"activation {name!r} is not supported by the HIP ABN (leaky_relu, elu, identity)") from None


# Bumped after every optimiser step (fused optimisers update parameters without touching their version counters):
# part of the key of the cached evaluation-mode constants of layers whose affine parameters are trainable.
_param_epoch = [0]


def _after_optimizer_step(optimizer, args, kwargs):
    _param_epoch[0] += 1


register_optimizer_step_post_hook(_after_optimizer_step)


# take the multi-rank code path at world 1 (profiling aid; bench.py --force_dist sets UCD_FORCE_COLLECTIVES for the whole step)
_FORCE_SYNC = os.environ.get("UCD_ABN_FORCE_SYNC") == "1" or os.environ.get("UCD_FORCE_COLLECTIVES") in ("1", "abn")


def _group_size(group):
    if group is False or not (dist.is_available() and dist.is_initialized()):
        return 1
    return dist.get_world_size(group if group is not None else None)


_node_mod = False


def _abn_node():
    """The C++ autograd node (ucd_amd/csrc/abn_node.cpp) or None when it has not been built; the Python Function below
    is the complete implementation, the node a faster host path for the training-mode layers."""
    global _node_mod
    if _node_mod is False:
        if _switches.get("UCD_ABN_NODE", "1") == "0":
            _node_mod = None
        else:
            try:
                hip.load()                                   # libucd_hip.so first: the node links against it
                from . import _abn_node as mod
                _node_mod = mod
            except ImportError:
                _node_mod = None
    return _node_mod


def _all_gather_stats(pack, world, group):
    """[world, 2C] table of every rank's (mean_r | M2_r) - the one forward collective of a SyncBN layer: on the
    library-owned RCCL communicator (compute stream, no dispatcher) when there is one, else through torch.distributed."""
    flat = torch.empty(world * pack.numel(), dtype=torch.float32, device=pack.device)
    comm = direct_comm(group) if pack.is_cuda else None
    if comm is not None:
        hip._check(hip.load().ucd_comm_all_gather(comm.handle, hip.ptr(pack), hip.ptr(flat), pack.numel(), hip.stream()),
                   "ucd_comm_all_gather")
    else:
        dist.all_gather_into_tensor(flat, pack, group=group if group is not None else None)
    return flat


def _all_reduce_sums(sums, group):
    """In-place sum over the ranks of a backward's [sum dz | sum dz*xhat] vector (same routing as the gather)."""
    comm = direct_comm(group) if sums.is_cuda else None
    if comm is not None:
        hip._check(hip.load().ucd_comm_all_reduce_sum(comm.handle, hip.ptr(sums), sums.numel(), hip.stream()),
                   "ucd_comm_all_reduce_sum")
    else:
        dist.all_reduce(sums, group=group if group is not None else None)


class _ABNFunction(torch.autograd.Function):
    """y = act(BN(x [+ plane_bias]) [+ residual]) with batch (training) or running (eval) statistics."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, plane_bias, running_mean, running_var, training, momentum, eps,
                act, slope, group, out, inplace, eval_cache=None):
        x_in = x
        x, M, Cc, HW, ld_x = hip.rows_view(x)
        dev = x.device
        ld_r = 0
        if residual is not None:
            residual, _, _, _, ld_r = hip.rows_view(residual)
        if plane_bias is not None:
            plane_bias = plane_bias.reshape(x.shape[0], Cc).float().contiguous()
        world = _group_size(group) if training else 1
        count = float(M * world)
        if out is not None:
            y = out
            y, _, _, _, ld_y = hip.rows_view(y)
            if y.data_ptr() != out.data_ptr():
                raise RuntimeError("ABN: `out` must be a channels-last tensor or channel slice")
        elif inplace:
            y, ld_y = x, ld_x
        else:
            y = hip.empty_like_rows(x)
            ld_y = Cc
        # [sums(2C) | kshift | mean | invstd | scale]
        use_cache = not training and eval_cache is not None
        sync = training and group is not False and (world > 1 or (_FORCE_SYNC and dist.is_initialized()))
        comm = direct_comm(group) if sync else None
        buf = None if use_cache else torch.empty(((8 + 2 * world) if sync else 6) * Cc, dtype=torch.float32, device=dev)
        if comm is not None:
            # the whole layer, statistics exchange included, in one library call on this stream
            hip.abn_sync_forward_comm(comm.handle, world, x, ld_x, y, ld_y, residual, ld_r, M, Cc, plane_bias, HW, weight,
                                      bias, running_mean, running_var, momentum, eps, buf, act, slope)
        elif not sync:
            # one library call: statistics + finalize + apply (training) or running-statistics apply (eval)
            hip.abn_forward(x, ld_x, y, ld_y, residual, ld_r, M, Cc, plane_bias, HW, weight, bias, running_mean,
                            running_var, momentum, eps, training, buf, eval_cache if use_cache else None, act, slope)
        else:
            # statistics of this rank -> all_gather -> combination + finalize + apply: two library calls, one collective
            pack = buf[6 * Cc:8 * Cc]
            hip.abn_sync_stats(x, ld_x, M, Cc, plane_bias, HW, buf[:2 * Cc], buf[2 * Cc:3 * Cc], pack)
            gathered = _all_gather_stats(pack, world, group)
            hip.abn_sync_forward(x, ld_x, y, ld_y, residual, ld_r, M, Cc, plane_bias, HW, gathered, world, weight, bias,
                                 running_mean, running_var, momentum, eps, buf, act, slope)
        if use_cache:   # the backward (if any) reads invstd / scale from the same [.. | invstd | scale] layout
            buf = torch.cat((torch.empty(4 * Cc, dtype=torch.float32, device=dev), eval_cache.reshape(-1))) \
                if ctx.needs_input_grad[0] else eval_cache
        mean = running_mean if not training else None
        needs_y = residual is not None and (act & hip.ACT_MASK) != hip.ACT_IDENTITY
        ctx.save_for_backward(x, y if needs_y else None, plane_bias, weight, bias, buf, mean)
        ctx.cfg = (M, Cc, HW, ld_x, ld_y, training, act, slope, group, count, sync, residual is not None,
                   plane_bias is not None, x.shape[0])
        ctx.comm = comm
        if y is x_in:
            ctx.mark_dirty(x_in)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, plane_bias, weight, shift, buf, mean_eval = ctx.saved_tensors
        M, Cc, HW, ld_x, ld_y, training, act, slope, group, count, sync, has_res, has_pb, B = ctx.cfg
        dy, _, _, _, ld_dy = hip.rows_view(dy if dy.dtype == x.dtype else dy.to(x.dtype))
        mean = buf[3 * Cc:4 * Cc] if training else mean_eval
        invstd, scale = buf[4 * Cc:5 * Cc], buf[5 * Cc:]
        sums = torch.empty((4 if sync else 2) * Cc, dtype=torch.float32, device=x.device)
        need_param_grad = weight is not None and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        dx = hip.empty_like_rows(x)
        dz = hip.empty_like_rows(x) if has_res else None
        ld_yy = ld_y if y is not None else 0
        if sync and ctx.comm is not None:
            hip.abn_sync_backward_comm(ctx.comm.handle, ctx.comm.world, x, ld_x, dy, ld_dy, y, ld_yy, dx, Cc, dz,
                                       Cc if has_res else 0, M, Cc, plane_bias, HW, mean, invstd, scale, shift, weight, sums,
                                       act, slope)
            dbias = dweight = None
            if need_param_grad:
                dbias, dweight = sums[2 * Cc:3 * Cc], sums[3 * Cc:]
        elif not sync:
            hip.abn_backward(x, ld_x, dy, ld_dy, y, ld_yy, dx, Cc, dz, Cc if has_res else 0, M, Cc, plane_bias, HW, mean,
                             invstd, scale, shift, weight, sums, count, training, need_param_grad, act, slope)
            dbias = dweight = None
            if need_param_grad:
                dbias, dweight = sums[:Cc], sums[Cc:]
        else:
            local, sums = sums[2 * Cc:], sums[:2 * Cc]           # this rank's sums = its d bias / d weight
            hip.abn_sync_bwd_reduce(x, ld_x, dy, ld_dy, y, ld_yy, M, Cc, plane_bias, HW, mean, invstd, scale, shift, act,
                                    slope, sums, local, weight)
            dbias = dweight = None
            if need_param_grad:
                dbias, dweight = local[:Cc], local[Cc:]
            _all_reduce_sums(sums, group)
            hip.abn_bwd_apply(x, ld_x, dy, ld_dy, y, ld_yy, dx, Cc, dz, Cc if has_res else 0, M, Cc, plane_bias, HW, mean,
                              invstd, scale, shift, weight, sums, count, 0, act, slope)
        dpb = None
        if has_pb:
            # plane_bias enters like x: its gradient is dx summed over each image plane
            dpb = torch.empty(B, Cc, dtype=torch.float32, device=x.device)
            hip.plane_sum(dx, Cc, B, HW, Cc, 1.0, dpb)
            dpb = dpb.view(B, Cc, 1, 1)
        return (dx, dweight, dbias, dz, dpb) + (None,) * 11


class _ABNBranchesFunction(torch.autograd.Function):
    """ABN over the channel concatenation of several maps, written branch by branch into one
    [B, sum C_i, H, W] buffer (the concatenation of modules/deeplab.py:56 is never materialised)."""

    @staticmethod
    def forward(ctx, weight, bias, running_mean, running_var, training, momentum, eps, act, slope, group, *xs):
        n = len(xs)
        views = [hip.rows_view(x) for x in xs]
        x0, M, _, HW, _ = views[0]
        chans = [v[2] for v in views]
        Ct = sum(chans)
        dev = x0.device
        out = hip.empty_like_rows(x0, channels=Ct)
        # per-branch [sums(2c) | kshift(c)] blocks first, then full-width mean | invstd | scale
        buf = torch.empty(6 * Ct, dtype=torch.float32, device=dev)
        mean, invstd, scale = buf[3 * Ct:4 * Ct], buf[4 * Ct:5 * Ct], buf[5 * Ct:]
        world = _group_size(group) if training else 1
        if world > 1:
            return _ABNBranchesFunction._forward_sync(ctx, weight, bias, running_mean, running_var, momentum, eps, act,
                                                      slope, group, views, out, buf, world)
        count = float(M * world)
        offs, o = [], 0
        for c in chans:
            offs.append(o)
            o += c
        for (x, _, c, _, ld), o in zip(views, offs):
            sl = slice(o, o + c)
            sums, kshift = buf[3 * o:3 * o + 2 * c], buf[3 * o + 2 * c:3 * (o + c)]
            w, b = (weight[sl], bias[sl]) if weight is not None else (None, None)
            if training:
                hip.abn_stats(x, ld, M, c, None, HW, sums, kshift)
                hip.abn_finalize(sums, kshift, count, c, w, running_mean[sl], running_var[sl], momentum, eps, mean[sl],
                                 invstd[sl], scale[sl], act)
                mu = mean[sl]
            else:
                mu = running_mean[sl]
                hip.abn_eval_params(w, running_var[sl], eps, c, invstd[sl], scale[sl], act)
            hip.abn_apply(x, ld, out[:, sl], Ct, None, 0, M, c, None, HW, mu, scale[sl], b, act, slope)
        if not training:
            mean.copy_(running_mean)
        ctx.save_for_backward(weight, bias, buf, *[v[0] for v in views])
        ctx.cfg = (M, HW, chans, offs, [v[4] for v in views], training, act, slope, group, count, world)
        return out

    @staticmethod
    def _forward_sync(ctx, weight, bias, running_mean, running_var, momentum, eps, act, slope, group, views, out, buf, world):
        """Training forward across ranks: every branch's (mean_r | M2_r) goes into ONE gather; the combination then runs
        per branch on its slice of the gathered table."""
        x0, M, _, HW, _ = views[0]
        chans = [v[2] for v in views]
        Ct = sum(chans)
        offs, o = [], 0
        for c in chans:
            offs.append(o)
            o += c
        # pack layout [mean(Ct) | M2(Ct)] so that a channel slice of the table is again [world][mean | M2] with stride 2Ct
        pack = torch.empty(2 * Ct, dtype=torch.float32, device=x0.device)
        tmp = torch.empty(2 * max(chans), dtype=torch.float32, device=x0.device)
        for (x, _, c, _, ld), o in zip(views, offs):
            hip.abn_sync_stats(x, ld, M, c, None, HW, buf[3 * o:3 * o + 2 * c], buf[3 * o + 2 * c:3 * (o + c)], tmp)
            pack[o:o + c].copy_(tmp[:c])
            pack[Ct + o:Ct + o + c].copy_(tmp[c:2 * c])
        gathered = _all_gather_stats(pack, world, group).view(world, 2, Ct)
        bbuf = torch.empty(6 * max(chans), dtype=torch.float32, device=x0.device)
        mean, invstd, scale = buf[3 * Ct:4 * Ct], buf[4 * Ct:5 * Ct], buf[5 * Ct:]
        for (x, _, c, _, ld), o in zip(views, offs):
            sl = slice(o, o + c)
            g = gathered[:, :, sl].contiguous()                  # [world][2][c] = [world][mean_r | M2_r]
            w, b = (weight[sl], bias[sl]) if weight is not None else (None, None)
            hip.abn_sync_forward(x, ld, out[:, sl], Ct, None, 0, M, c, None, HW, g, world, w, b, running_mean[sl],
                                 running_var[sl], momentum, eps, bbuf, act, slope)
            mean[sl].copy_(bbuf[3 * c:4 * c]); invstd[sl].copy_(bbuf[4 * c:5 * c]); scale[sl].copy_(bbuf[5 * c:6 * c])
        ctx.save_for_backward(weight, bias, buf, *[v[0] for v in views])
        ctx.cfg = (M, HW, chans, offs, [v[4] for v in views], True, act, slope, group, float(M * world), world)
        return out

    @staticmethod
    def backward(ctx, dy):
        weight, bias, buf, *xs = ctx.saved_tensors
        M, HW, chans, offs, lds, training, act, slope, group, count, world = ctx.cfg
        Ct = sum(chans)
        dy, _, _, _, ld_dy = hip.rows_view(dy if dy.dtype == xs[0].dtype else dy.to(xs[0].dtype))
        mean, invstd, scale = buf[3 * Ct:4 * Ct], buf[4 * Ct:5 * Ct], buf[5 * Ct:]
        shift = bias if bias is not None else torch.zeros(Ct, dtype=torch.float32, device=dy.device)
        sums = torch.empty(2 * Ct, dtype=torch.float32, device=dy.device)
        if training:
            for x, c, o, ld in zip(xs, chans, offs, lds):
                sl = slice(o, o + c)
                hip.abn_bwd_reduce(x, ld, dy[:, sl], ld_dy, None, 0, M, c, None, HW, mean[sl], invstd[sl], scale[sl],
                                   shift[sl], act, slope, sums[2 * o:2 * (o + c)], weight[sl] if weight is not None else None)
        dweight = dbias = None
        if training and weight is not None and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
            dbias = torch.cat([sums[2 * o:2 * o + c] for c, o in zip(chans, offs)])
            dweight = torch.cat([sums[2 * o + c:2 * (o + c)] for c, o in zip(chans, offs)])
        if training and world > 1:
            _all_reduce_sums(sums, group)
        dxs = []
        for x, c, o, ld in zip(xs, chans, offs, lds):
            sl = slice(o, o + c)
            dx = hip.empty_like_rows(x)
            hip.abn_bwd_apply(x, ld, dy[:, sl], ld_dy, None, 0, dx, c, None, 0, M, c, None, HW, mean[sl], invstd[sl],
                              scale[sl], shift[sl], weight[sl] if weight is not None else None,
                              sums[2 * o:2 * (o + c)], count, 0 if training else 1, act, slope)
            dxs.append(dx)
        return (dweight, dbias) + (None,) * 8 + tuple(dxs)


class _PlaneMean(torch.autograd.Function):
    """Global average pooling [B, C, H, W] -> [B, C, 1, 1] (modules/deeplab.py:72-76)."""

    @staticmethod
    def forward(ctx, x):
        x, M, Cc, HW, ld = hip.rows_view(x)
        B = x.shape[0]
        out = torch.empty(B, Cc, dtype=torch.float32, device=x.device)
        hip.plane_sum(x, ld, B, HW, Cc, 1.0 / HW, out)
        ctx.shape, ctx.dtype = x.shape, x.dtype
        return out.to(x.dtype).view(B, Cc, 1, 1)

    @staticmethod
    def backward(ctx, g):
        B, Cc, H, W = ctx.shape
        return (g.to(ctx.dtype) / (H * W)).expand(B, Cc, H, W)


def global_avg_pool(x):
    return _PlaneMean.apply(x)


class _StemABNPoolFunction(torch.autograd.Function):
    """``max_pool2d(abn(z), 3, stride 2, padding 1)`` of the stem (models/resnet.py:58-64: mod1.bn1 + mod1.pool1) as one pass
    forward and two passes backward (csrc/stem.hip): the normalised 257 x 257 map - the largest activation of the network - is
    never written, and the pooling's backward, the norm's reduction and the norm's apply stop being three trips over it.
    Training: batch statistics (shifted sums + finalize; under SyncBN the per-rank (mean, M2) pairs are gathered and combined),
    then the fused apply + pool; evaluation: the running-statistics constants.  The backward's two sums are all-reduced between
    its phases under SyncBN, exactly like ucd_abn_sync_backward_comm."""

    @staticmethod
    def forward(ctx, z, weight, bias, running_mean, running_var, training, momentum, eps, act, slope, group, eval_consts):
        B, Cc, H, W = z.shape
        M, HW = B * H * W, H * W
        zr, _, _, _, ld = hip.rows_view(z)
        dev = z.device
        world = _group_size(group) if training else 1
        sync = training and group is not False and (world > 1 or (_FORCE_SYNC and dist.is_initialized()))
        if training:
            buf = torch.empty(((8 + 2 * world) if sync else 6) * Cc, dtype=torch.float32, device=dev)
            if sync:
                pack = buf[6 * Cc:8 * Cc]
                hip.abn_sync_stats(zr, ld, M, Cc, None, HW, buf[:2 * Cc], buf[2 * Cc:3 * Cc], pack)
                comm = direct_comm(group)
                if comm is not None:
                    gathered = buf[8 * Cc:]
                    hip._check(hip.load().ucd_comm_all_gather(comm.handle, hip.ptr(pack), hip.ptr(gathered), 2 * Cc, hip.stream()),
                               "ucd_comm_all_gather")
                else:
                    gathered = _all_gather_stats(pack, world, group)
                hip.abn_sync_finalize(gathered, world, M, Cc, weight, running_mean, running_var, momentum, eps, buf, act)
            else:
                hip.abn_stats_finalize(zr, ld, M, Cc, None, HW, buf[:2 * Cc], buf[2 * Cc:3 * Cc], weight, running_mean, running_var,
                                       momentum, eps, buf[3 * Cc:4 * Cc], buf[4 * Cc:5 * Cc], buf[5 * Cc:6 * Cc], act)
            mean, invstd, scale = buf[3 * Cc:4 * Cc], buf[4 * Cc:5 * Cc], buf[5 * Cc:6 * Cc]
        else:
            mean, invstd, scale = running_mean, eval_consts[0], eval_consts[1]
        need_bwd = any(ctx.needs_input_grad[:3])
        out, idx = hip.stem_apply_pool(z, mean, scale, bias, act, slope, need_bwd)
        if need_bwd:
            ctx.save_for_backward(z, idx, weight, bias, mean, invstd, scale)
            ctx.cfg = (training, act, slope, group, world, sync, float(M * world))
        return out

    @staticmethod
    def backward(ctx, dp):
        z, idx, weight, bias, mean, invstd, scale = ctx.saved_tensors
        training, act, slope, group, world, sync, count = ctx.cfg
        if not training:
            raise RuntimeError("ucd_amd.abn: the fused stem has no frozen-statistics backward (fix_bn keeps the separate layers)")
        Cc = z.shape[1]
        if dp.dtype != z.dtype:
            dp = dp.to(z.dtype)
        if not dp.is_contiguous(memory_format=torch.channels_last):
            dp = dp.contiguous(memory_format=torch.channels_last)
        sums = torch.empty(2 * Cc, dtype=torch.float32, device=z.device)
        dz = torch.empty_like(z) if ctx.needs_input_grad[0] else None
        if not sync:
            hip.stem_pool_backward(z, dp, idx, mean, invstd, scale, bias, weight, sums, count, act, slope, dz, 3 if dz is not None else 1)
            dbias, dweight = sums[:Cc], sums[Cc:]
        else:
            hip.stem_pool_backward(z, dp, idx, mean, invstd, scale, bias, weight, sums, count, act, slope, None, 1)
            local = sums.clone()                                   # this rank's sums = its d bias / d weight
            dbias, dweight = local[:Cc], local[Cc:]
            comm = direct_comm(group)
            if comm is not None:
                hip._check(hip.load().ucd_comm_all_reduce_sum(comm.handle, hip.ptr(sums), 2 * Cc, hip.stream()), "ucd_comm_all_reduce_sum")
            else:
                _all_reduce_sums(sums, group)
            if dz is not None:
                hip.stem_pool_backward(z, dp, idx, mean, invstd, scale, bias, weight, sums, count, act, slope, dz, 2)
        return (dz, dweight if weight is not None else None, dbias if bias is not None else None) + (None,) * 9


def stem_norm_pool(bn, z):
    """``pool(bn(z))`` of the stem through the fused kernels, or None when this layer / input is not one they take (the caller
    then runs the two modules): HIP ABN with leaky_relu / identity, dense channels-last bf16 map, channel count a multiple of
    8 dividing 2048; frozen statistics only without gradients (``UCD_STEM_FOLD=0`` switches the fold off)."""
    import os
    if _switches.get("UCD_STEM_FOLD", "1") == "0" or not getattr(bn, "ucd_fused_abn", False):
        return None
    Cc = z.shape[1] if z.dim() == 4 else 0
    if not (z.is_cuda and z.dim() == 4 and z.dtype == torch.bfloat16 and z.is_contiguous(memory_format=torch.channels_last)
            and z.shape[2] > 2 and z.shape[3] > 2 and Cc % 8 == 0 and 2048 % max(Cc, 1) == 0
            and bn.activation in ("leaky_relu", "identity")):
        return None
    needs_grad = torch.is_grad_enabled() and (z.requires_grad or (bn.weight is not None and bn.weight.requires_grad))
    if needs_grad and not bn.training:
        return None
    act = _act_code(bn.activation) | (hip.NORM_ABS_GAMMA if bn._abs_gamma else 0)
    if bn.training:
        bn.__dict__.pop("_eval_cache", None)
    return _StemABNPoolFunction.apply(z, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.training, bn.momentum, bn.eps, act,
                                      bn.activation_param, bn._group(), None if bn.training else bn._eval_constants())


class ABN(nn.Module):
    """BatchNorm + activation on the HIP kernels; base class of the in-place variants."""

    ucd_fused_abn = True
    _inplace_contract = False
    _sync = False
    _abs_gamma = False          # F.batch_norm's raw gamma (inplace_abn.ABN); the in-place variants use |gamma| + eps

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, activation="leaky_relu",
                 activation_param=0.01, group=None):
        super().__init__()
        self.num_features, self.eps, self.momentum, self.affine = num_features, eps, momentum, affine
        self.activation, self.activation_param = activation, activation_param
        self.group = group
        if affine:
            self.weight = nn.Parameter(torch.ones(num_features))
            self.bias = nn.Parameter(torch.zeros(num_features))
        else:
            self.register_parameter("weight", None)
            self.register_parameter("bias", None)
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))

    def reset_parameters(self):
        nn.init.constant_(self.running_mean, 0)
        nn.init.constant_(self.running_var, 1)
        if self.affine:
            nn.init.constant_(self.weight, 1)
            nn.init.constant_(self.bias, 0)

    def _group(self):
        return self.group if self._sync else False

    def forward(self, x, residual=None, activation=None, activation_param=None, plane_bias=None, out=None):
        if not x.is_cuda:
            raise RuntimeError("ucd_amd.abn runs on the GPU only (there is no CPU fallback)")
        act = _act_code(self.activation if activation is None else activation) | (hip.NORM_ABS_GAMMA if self._abs_gamma else 0)
        slope = self.activation_param if activation_param is None else activation_param
        if self.training:
            self.__dict__.pop("_eval_cache", None)     # the running statistics are about to change
        inplace = self._inplace_contract and not torch.is_grad_enabled() and out is None
        if not torch.is_grad_enabled() and not self.training and residual is None and plane_bias is None and out is None:
            return self._forward_eval_nograd(x, act, slope, inplace)
        if self.training and plane_bias is None and out is None and self.weight is not None and torch.is_grad_enabled():
            node = _abn_node()
            if node is not None and node.dense_channels_last(x) and (residual is None or node.dense_channels_last(residual)):
                group = self._group()
                world = _group_size(group)
                sync = group is not False and (world > 1 or (_FORCE_SYNC and dist.is_initialized()))
                comm = direct_comm(group) if sync else None
                if comm is not None or not sync:
                    return node.abn_train(x, self.weight, self.bias, residual, self.running_mean, self.running_var,
                                          self.momentum, self.eps, act, slope, comm.handle if comm is not None else 0,
                                          world, hip.stream(), self._direct_grad_ptr())
        return _ABNFunction.apply(x, self.weight, self.bias, residual, plane_bias, self.running_mean,
                                  self.running_var, self.training, self.momentum, self.eps, act, slope,
                                  self._group(), out, inplace, None if self.training else self._eval_constants())

    _direct_grads = None        # set by ucd_amd.ddp: (flat [d bias | d weight] view, its address) the kernels write into

    def _direct_grad_ptr(self):
        """Address of this layer's [d bias | d weight] gradient storage when the reducer owns it (the backward kernels
        then write the parameter gradients in place), else 0.  Valid only while both .grad views are the ones handed out."""
        d = self._direct_grads
        if d is None or not (self.weight.requires_grad and self.bias.requires_grad):
            return 0
        flat, addr = d
        C = self.num_features
        gb, gw = self.bias.grad, self.weight.grad
        if gb is None or gw is None or gb.data_ptr() != addr or gw.data_ptr() != addr + 4 * C:
            return 0
        return addr

    def _forward_eval_nograd(self, x, act, slope, inplace):
        """Frozen-statistics forward outside autograd (the teacher): one library call, no Function object."""
        x, M, Cc, HW, ld_x = hip.rows_view(x)
        y = x if inplace else hip.empty_like_rows(x)
        hip.abn_forward(x, ld_x, y, ld_x if inplace else Cc, None, 0, M, Cc, None, HW, self.weight, self.bias,
                        self.running_mean, self.running_var, self.momentum, self.eps, False, None,
                        self._eval_constants(), act, slope)
        return y

    def _eval_constants(self):
        """invstd / scale of the running statistics, cached until a parameter or buffer changes (the frozen
        teacher never recomputes them)."""
        # what can change the constants: in-place writes seen by the version counters (load_state_dict, copy_), a
        # training-mode forward of this layer (it drops the cache: the kernels update running_var through raw pointers),
        # and an optimiser step on a trainable gamma (fused optimisers do not bump versions: _param_epoch does)
        w = self.weight
        key = (self.running_var._version, self.running_var.data_ptr(), self.eps,
               None if w is None else (w._version, w.data_ptr(), _param_epoch[0] if w.requires_grad else -1))
        cache = self.__dict__.get("_eval_cache")
        if cache is None or cache[0] != key:
            Cc = self.num_features
            c = torch.empty(2, Cc, dtype=torch.float32, device=self.running_var.device)
            hip.abn_eval_params(self.weight, self.running_var, self.eps, Cc, c[0], c[1],
                                hip.NORM_ABS_GAMMA if self._abs_gamma else 0)
            cache = (key, c)
            self.__dict__["_eval_cache"] = cache
        return cache[1]

    def forward_branches(self, xs):
        """``self(torch.cat(xs, 1))`` without the concatenation."""
        return _ABNBranchesFunction.apply(self.weight, self.bias, self.running_mean, self.running_var, self.training,
                                          self.momentum, self.eps,
                                          _act_code(self.activation) | (hip.NORM_ABS_GAMMA if self._abs_gamma else 0),
                                          self.activation_param, self._group(), *xs)

    def extra_repr(self):
        s = "{num_features}, eps={eps}, momentum={momentum}, affine={affine}, activation={activation}"
        if self.activation in ("leaky_relu", "elu"):
            s += "[{activation_param}]"
        return s.format(**self.__dict__)


class InPlaceABN(ABN):
    """``gamma~ = |gamma| + eps`` like inplace_abn's in-place kernels; under ``no_grad`` the input tensor is overwritten
    (inplace_abn's contract)."""
    _inplace_contract = True
    _abs_gamma = True


class InPlaceABNSync(InPlaceABN):
    """InPlaceABN with statistics reduced over the process group (RCCL over xGMI)."""
    _sync = True
