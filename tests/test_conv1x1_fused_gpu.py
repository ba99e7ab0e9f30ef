"""GPU: the fused 1x1-convolution GEMM (csrc/conv1x1.hip) through the C ABI - every mode against an fp32
product of the same bf16 operands on the device (hipBLASLt / MIOpen are the A/B reference here, never a fallback), and the
module paths built on it against the layer-by-layer path and the reference goldens."""
import os

import pytest
import torch
import torch.nn.functional as F

from ucd_amd import switches, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _mk(M, K, N, seed):
    g = torch.Generator(DEV).manual_seed(seed)
    a = (torch.randn(M, K, device=DEV, generator=g) * 1.3 + 0.2).bfloat16()
    w = (torch.randn(N, K, device=DEV, generator=g) * (2.0 / K) ** 0.5).bfloat16()
    r = torch.randn(M, N, device=DEV, generator=g).bfloat16()
    vk = [torch.randn(K, device=DEV, generator=g) * 0.3, torch.rand(K, device=DEV, generator=g) + 0.5,
          torch.randn(K, device=DEV, generator=g) * 0.2]
    vn = [torch.randn(N, device=DEV, generator=g) * 0.3, torch.rand(N, device=DEV, generator=g) + 0.5,
          torch.randn(N, device=DEV, generator=g) * 0.2, torch.rand(N, device=DEV, generator=g) + 0.5]
    return a, w, r, vk, vn


def _rel(x, ref):
    return ((x.float() - ref).norm() / ref.norm()).item()


SHAPES = [(300, 64, 64), (1000, 128, 256), (2178, 256, 128), (4356, 512, 1024), (777, 64, 256), (129, 1024, 192), (128, 64, 320)]


@pytest.mark.parametrize("M,K,N", SHAPES)
def test_conv1x1_modes_against_fp32_product(M, K, N):
    """bf16 rounding of the stored result is the only error: 2^-9 relative per element, ~1.7e-3 in L2."""
    from ucd_amd import hip
    a, w, r, (im, isc, ish), (om, osc, osh, oinv) = _mk(M, K, N, 3 + M)
    af, wf = a.float(), w.float()
    ref = af @ wf.t()
    y = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    hip.conv1x1(a, w, y)
    assert _rel(y, ref) < 3e-3
    # exact product check with small integers (catches any fragment / swizzle / tile-mapping slip: every output is exact)
    ai = torch.randint(-3, 4, (M, K), device=DEV).bfloat16()
    wi = torch.randint(-2, 3, (N, K), device=DEV).bfloat16()
    yi = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    hip.conv1x1(ai, wi, yi)
    exact = ai.float() @ wi.float().t()
    assert torch.equal(yi.float(), exact.bfloat16().float())
    # accumulate (beta = 1)
    y2 = r.clone()
    hip.conv1x1(a, w, y2, accumulate=True)
    assert _rel(y2, ref + r.float()) < 3e-3
    # input transform + affine + residual + activation
    ap = F.leaky_relu((af - im) * isc + ish, 0.01).bfloat16().float()
    hip.conv1x1(a, w, y, in_norm=(im, isc, ish, hip.ACT_LEAKY_RELU, 0.01), out_mode=1,
                out_norm=(om, osc, osh, None, hip.ACT_LEAKY_RELU, 0.01), residual=r)
    ref3 = F.leaky_relu((ap @ wf.t() - om) * osc + osh + r.float(), 0.01)
    assert _rel(y, ref3) < 3e-3
    # identity activations, no residual
    hip.conv1x1(a, w, y, out_mode=1, out_norm=(om, osc, osh, None, hip.ACT_IDENTITY, 0.0))
    assert _rel(y, (ref - om) * osc + osh) < 3e-3
    # statistics epilogue + finalize (|gamma| + eps scale, running statistics)
    tiles = hip.load().ucd_conv1x1_row_tiles(M)
    part = hip.conv1x1_stats_partial(M, N, DEV).fill_(float("nan"))
    hip.conv1x1(a, w, y, out_mode=2, partial=part)
    buf = torch.zeros(6 * N, device=DEV)
    rm, rv = torch.zeros(N, device=DEV), torch.ones(N, device=DEV)
    gamma = torch.rand(N, device=DEV) - 0.3
    hip._check(hip.load().ucd_conv1x1_stats_finalize(hip.ptr(part), M, N, hip.ptr(gamma), hip.ptr(rm), hip.ptr(rv), 0.1, 1e-5,
                                                     hip.ptr(buf), None, hip.NORM_ABS_GAMMA, hip.stream()), "finalize")
    yf = y.float()                                   # statistics are those of the STORED tensor
    mean, var = yf.mean(0), yf.var(0, unbiased=False)
    torch.testing.assert_close(buf[3 * N:4 * N], mean, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(buf[4 * N:5 * N], 1 / torch.sqrt(var + 1e-5), rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(buf[5 * N:], (gamma.abs() + 1e-5) / torch.sqrt(var + 1e-5), rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(rm, 0.1 * mean, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(rv, 0.9 + 0.1 * var * M / (M - 1), rtol=1e-4, atol=1e-6)
    # SyncBN packing: (mean_r, M2_r)
    pack = torch.zeros(2 * N, device=DEV)
    hip._check(hip.load().ucd_conv1x1_stats_finalize(hip.ptr(part), M, N, None, None, None, 0.1, 1e-5, hip.ptr(buf), hip.ptr(pack),
                                                     0, hip.stream()), "finalize")
    torch.testing.assert_close(pack[:N], mean, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(pack[N:], var * M, rtol=1e-3, atol=1e-3)
    # activation backward + sums
    part2 = torch.full((tiles, 2, N), float("nan"), device=DEV)
    hip.conv1x1(a, w, y, out_mode=3, out_norm=(om, osc, osh, oinv, hip.ACT_LEAKY_RELU, 0.01), residual=r, partial=part2)
    z = (r.float() - om) * osc + osh
    assert _rel(y, ref * torch.where(z > 0, 1.0, 0.01)) < 3e-3
    sums = torch.zeros(2 * N, device=DEV)
    sgn = torch.where(gamma < 0, -1.0, 1.0)
    hip._check(hip.load().ucd_abn_reduce_partials(hip.ptr(part2), tiles, N, hip.ptr(sums), None, hip.ptr(gamma),
                                                  hip.NORM_ABS_GAMMA, hip.stream()), "reduce")
    dz = y.float()
    torch.testing.assert_close(sums[:N], dz.sum(0), rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(sums[N:], sgn * (dz * (r.float() - om) * oinv).sum(0), rtol=1e-3, atol=1e-2)
    # transposed weight and (shapes it supports) the weight gradient
    wt = torch.empty(K, N, device=DEV, dtype=torch.bfloat16)
    hip.transpose_bf16(w, wt)
    assert torch.equal(wt, w.t().contiguous())
    if N % 128 == 0 and K % 128 == 0:
        dw = torch.empty(N, K, device=DEV, dtype=torch.bfloat16)
        hip.conv1x1_wgrad(r, a, dw)
        assert _rel(dw, r.float().t() @ af) < 3e-3
        hip.conv1x1_wgrad(r, a, dw, in_norm=(im, isc, ish, hip.ACT_LEAKY_RELU, 0.01))
        assert _rel(dw, r.float().t() @ ap) < 3e-3


def test_conv1x1_argument_errors_and_slice_output():
    from ucd_amd import hip
    a, w, r, _, (om, osc, osh, _) = _mk(256, 64, 64, 1)
    y = torch.empty(256, 64, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError):            # K not a multiple of 64
        hip.conv1x1(a[:, :32].contiguous(), w[:, :32].contiguous(), y)
    with pytest.raises(RuntimeError):            # elu is not a fused activation
        hip.conv1x1(a, w, y, out_mode=1, out_norm=(om, osc, osh, None, hip.ACT_ELU, 1.0))
    # output into a channel slice of a wider buffer (the ASPP concatenation), input a channel slice too
    wide = torch.zeros(256, 256, device=DEV, dtype=torch.bfloat16)
    awide = torch.zeros(256, 192, device=DEV, dtype=torch.bfloat16)
    awide[:, 64:128] = a
    hip.conv1x1(awide[:, 64:128], w, wide[:, 128:192])
    ref = a.float() @ w.float().t()
    assert _rel(wide[:, 128:192], ref) < 3e-3
    assert not wide[:, :128].any() and not wide[:, 192:].any()


@pytest.mark.parametrize("stride,dil,cin", [(1, 1, 256), (2, 1, 128), (1, 2, 256)])
def test_residual_block_eval_fused_equals_layer_by_layer(stride, dil, cin):
    """The frozen-statistics forward of a bottleneck (the teacher): conv1 + bn1 and conv3 + bn3 + shortcut + activation as
    one GEMM each, against the same module with the fusion switched off (MIOpen / hipBLASLt + separate ABN kernels) and
    against the fp32 run of the module."""
    from functools import partial
    from ucd_amd.abn import InPlaceABNSync
    from ucd_amd.blocks import ResidualBlock
    norm = partial(InPlaceABNSync, activation="leaky_relu", activation_param=0.01)
    blk = ResidualBlock(cin, (64, 64, 256), norm_act=norm, stride=stride, dilation=dil)
    blk.load_state_dict(synth.fill_state_dict(blk.state_dict(), 5))
    blk = blk.to(DEV).to(memory_format=torch.channels_last).eval()
    x = synth.t_normal(9, (3, cin, 17, 19), stream=1).to(DEV).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        ref32 = blk(x.clone())
        with torch.autocast("cuda", dtype=torch.bfloat16):
            xb = x.bfloat16()
            assert blk._eval_fusable(xb)
            fused = blk(xb.clone())
            switches.set("UCD_FUSED_CONV1X1", "0")
            try:
                assert not blk._eval_fusable(xb)
                plain = blk(xb.clone())
            finally:
                switches.unset("UCD_FUSED_CONV1X1")
    assert fused.dtype == torch.bfloat16 and fused.shape == ref32.shape
    e_f, e_p = _rel(fused, ref32), _rel(plain, ref32)
    assert e_f < 1e-2 and e_f < 1.5 * e_p + 1e-3, (e_f, e_p)       # no worse than the unfused bf16 path


@pytest.mark.parametrize("cin,chans,with_ddp", [(1024, (256, 256, 1024), False), (512, (256, 256, 1024), False),
                                                (1024, (256, 256, 1024), True)])
def test_residual_block_training_fused_node_equals_module_path(cin, chans, with_ddp):
    """Training forward + backward of a wide bottleneck with every 1x1 convolution + ABN as one node (statistics in the GEMM
    epilogue, C++ autograd) against the same block run module by module (library GEMM + separate statistics kernels):
    same outputs, input / parameter gradients and running statistics up to bf16 rounding of the intermediate maps."""
    from functools import partial
    from ucd_amd import abn
    from ucd_amd.blocks import ResidualBlock
    from ucd_amd.ddp import DistributedDataParallel
    assert abn._abn_node() is not None and hasattr(abn._abn_node(), "conv_abn_train")
    norm = partial(abn.InPlaceABNSync, activation="leaky_relu", activation_param=0.01)
    x0 = synth.t_normal(9, (4, cin, 15, 13), stream=1).to(DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    dy = synth.t_normal(10, (4, chans[2], 15, 13), stream=1).to(DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    outs = []
    for fused in (True, False):
        blk = ResidualBlock(cin, chans, norm_act=norm, stride=1, dilation=1)
        blk.load_state_dict(synth.fill_state_dict(blk.state_dict(), 5))
        blk = blk.to(DEV).to(memory_format=torch.channels_last).train()
        mod = DistributedDataParallel(blk, bf16_weights=True) if with_ddp else blk
        if not fused:
            switches.set("UCD_FUSED_CONV1X1", "0")
        try:
            x = x0.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = mod(x * 1.0)
            y.backward(dy)
            if with_ddp:
                mod.finish_grad_sync()
        finally:
            switches.unset("UCD_FUSED_CONV1X1")
        grads = {n: p.grad.float().clone() for n, p in blk.named_parameters()}
        outs.append((y.detach().float(), x.grad.float(), grads, blk.convs.bn3.running_var.clone(), blk.convs.bn1.running_mean.clone()))
    (yf, gxf, gf, rvf, rmf), (yp, gxp, gp, rvp, rmp) = outs
    assert _rel(yf, yp) < 1e-2
    assert _rel(gxf, gxp) < 3e-2
    torch.testing.assert_close(rvf, rvp, rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(rmf, rmp, rtol=1e-3, atol=1e-4)
    for n in gf:
        assert _rel(gf[n], gp[n]) < 5e-2, n


@pytest.mark.parametrize("tile", [32, 64])
def test_batched_flip_kernels_on_ragged_shapes(tile):
    """ucd_flip_weights_batched (32 x 32 tiles, 2-byte accesses) and ucd_flip_weights_batched64 (64 x 64 tiles, 16-byte accesses where
    the addresses allow) on a table of layers whose channel counts are not multiples of the tile, of 8, and whose offsets into the
    flat buffers are odd: dst_e == src_e.flip(2, 3).transpose(0, 1) in channels-last memory order, nothing outside the entries written."""
    from ucd_amd import hip
    lib = hip.load()
    shapes = [(64, 64, 3), (256, 128, 3), (21, 256, 1), (40, 72, 3), (72, 40, 1), (130, 66, 3), (8, 8, 1), (1, 300, 1)]
    g = torch.Generator(DEV).manual_seed(11)
    entries, blocks, srcs, off_s, off_d = [], [], [], 3, 5                     # odd starts: no 16-byte alignment for free
    for e, (co, ci, k) in enumerate(shapes):
        n = co * ci * k * k
        entries.append([off_s, off_d, co, ci, k * k])
        for sp in range(k * k):
            for a in range((co + tile - 1) // tile):
                for b in range((ci + tile - 1) // tile):
                    blocks.append([e, sp, a, b])
        srcs.append((off_s, off_d, n))
        off_s += n + (e % 3)                                                      # ragged gaps between the entries
        off_d += n + ((e + 1) % 4)
    src = torch.randint(-30000, 30000, (off_s + 8,), device=DEV, generator=g, dtype=torch.int16)
    dst = torch.full((off_d + 8,), 12345, device=DEV, dtype=torch.int16)
    ent = torch.tensor(entries, dtype=torch.int64, device=DEV)
    blk = torch.tensor(blocks, dtype=torch.int32, device=DEV)
    fn = lib.ucd_flip_weights_batched if tile == 32 else lib.ucd_flip_weights_batched64
    hip._check(fn(src.data_ptr(), dst.data_ptr(), blk.data_ptr(), blk.shape[0], ent.data_ptr(), hip.stream()), "flip")
    torch.cuda.synchronize()
    covered = torch.zeros_like(dst, dtype=torch.bool)
    for (so, do, n), (co, ci, k) in zip(srcs, shapes):
        w = src[so:so + n].view(co, k, k, ci).permute(0, 3, 1, 2)                 # [co, ci, kh, kw] behind channels-last memory
        ref = w.flip(2, 3).transpose(0, 1).permute(0, 2, 3, 1).reshape(-1)       # [ci][kh][kw][co] memory order
        assert torch.equal(dst[do:do + n], ref), (co, ci, k)
        covered[do:do + n] = True
    assert (dst[~covered] == 12345).all()


def test_cached_flipped_weights_equal_flip_transpose():
    """ucd_flip_weights_batched (one launch for every stride-1 layer, refreshed with the bf16 working copies) against
    w.flip(2, 3).transpose(0, 1) per layer, before and after an optimiser step; and the step with the cached copies gives the
    same input gradients as the per-call flip."""
    from functools import partial
    from ucd_amd import abn
    from ucd_amd.blocks import Conv1x1, Conv3x3, ResidualBlock
    from ucd_amd.master import Bf16Weights
    norm = partial(abn.InPlaceABN, activation="leaky_relu", activation_param=0.01)
    net = torch.nn.Sequential(ResidualBlock(64, (64, 64, 256), norm_act=norm, stride=1, dilation=1),
                              ResidualBlock(256, (128, 128, 256), norm_act=norm, stride=1, dilation=2))
    net.load_state_dict(synth.fill_state_dict(net.state_dict(), 3))
    net = net.to(DEV).to(memory_format=torch.channels_last).train()
    bw = Bf16Weights(net)
    mods = [m for m in net.modules() if isinstance(m, (Conv3x3, Conv1x1)) and m._w16_flip is not None]
    assert len(mods) >= 5
    opt = torch.optim.SGD(net.parameters(), lr=0.1)

    def check():
        bw.refresh_if_stale()
        for m in mods:
            ref = m._w16.detach().flip(2, 3).transpose(0, 1)
            assert m._w16_flip.shape == ref.shape and torch.equal(m._w16_flip, ref)
            assert m._w16_flip.is_contiguous(memory_format=torch.channels_last)
    check()
    x = synth.t_normal(4, (24, 64, 21, 19), stream=1).to(DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    grads = []
    for cached in (True, False):
        xi = x.clone().requires_grad_(True)
        if not cached:
            saved = [(m, m._w16_flip) for m in mods]
            for m in mods:
                m._w16_flip = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            net(xi).float().square().mean().backward()
        grads.append(xi.grad.float().clone())
        if not cached:
            for m, f in saved:
                m._w16_flip = f
        for p in net.parameters():
            p.grad = None
        for m in net.modules():
            if getattr(m, "_w16", None) is not None:
                m._w16.grad = None
    assert _rel(grads[0], grads[1]) < 2e-2          # same solvers, same values: only MIOpen's own run-to-run noise
    for p in net.parameters():
        p.grad = torch.randn_like(p) * 0.01
    opt.step()
    check()


@pytest.mark.usefixtures("deterministic_stats")
def test_conv_abn_python_twin_equals_cpp_node():
    """blocks._ConvABNFunction (the complete Python implementation, what bench.py's instrumented pass runs) and
    csrc/abn_node.cpp::ConvABNTrainNode issue the same library calls: bit-identical outputs, gradients and statistics."""
    from functools import partial
    from ucd_amd import abn, blocks
    norm = partial(abn.InPlaceABN, activation="leaky_relu", activation_param=0.01)
    node = blocks._gemm_node()
    assert node is not None and hasattr(node, "conv_abn_train")
    x0 = synth.t_normal(9, (4, 1024, 15, 13), stream=1).to(DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    dy = synth.t_normal(10, (4, 1024, 15, 13), stream=1).to(DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    outs = []
    torch.backends.cudnn.deterministic = True
    try:
        for use_node in (True, False):
            blocks._node_cache[0] = node if use_node else None
            try:
                blk = blocks.ResidualBlock(1024, (256, 256, 1024), norm_act=norm, stride=1, dilation=1)
                blk.load_state_dict(synth.fill_state_dict(blk.state_dict(), 5))
                blk = blk.to(DEV).to(memory_format=torch.channels_last).train()
                x = x0.clone().requires_grad_(True)
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    y = blk(x * 1.0)
                y.backward(dy)
                outs.append([y.detach(), blk.convs.bn1.weight.grad.clone(), blk.convs.bn3.bias.grad.clone(),
                             blk.convs.bn3.running_var.clone(), blk.convs.conv3.weight.grad.clone(), x.grad.clone()])
            finally:
                blocks._node_cache[0] = node
    finally:
        torch.backends.cudnn.deterministic = False
    for i, (a, b) in enumerate(zip(*outs)):
        if i < 5:
            assert torch.equal(a, b), i
        else:   # the node folds the shortcut's gradient into the GEMM (one rounding), the twin leaves the add to autograd (two)
            assert _rel(a, b.float()) < 1e-2


@pytest.mark.parametrize("B,K,N,H,W,d", [(2, 64, 64, 9, 11, 1), (3, 128, 192, 17, 13, 2), (2, 256, 128, 33, 33, 6), (1, 64, 64, 5, 5, 12),
                                         (2, 64, 256, 16, 8, 18)])
def test_conv3x3_implicit_gemm_mode(B, K, N, H, W, d):
    """taps = 9: the same kernel as a 3x3 convolution (stride 1, padding = dilation; modules/residual.py:69, the ASPP branches
    of modules/deeplab.py:27-29) against F.conv2d in fp32 on the same bf16 operands: exact on small integers (tap shifts,
    zero padding, dilation larger than the map, tile edges), bf16-rounding close on random data, statistics / affine
    epilogues, and the input gradient as the same call on the flipped + transposed weight."""
    from ucd_amd import hip
    cl = torch.channels_last
    rows = lambda t: t.permute(0, 2, 3, 1).reshape(t.shape[0] * t.shape[2] * t.shape[3], t.shape[1])
    wrow = lambda w: w.permute(0, 2, 3, 1).reshape(w.shape[0], 9 * w.shape[1])
    g = torch.Generator(DEV).manual_seed(K + H)
    xi = torch.randint(-3, 4, (B, K, H, W), device=DEV, generator=g).bfloat16().contiguous(memory_format=cl)
    wi = torch.randint(-2, 3, (N, K, 3, 3), device=DEV, generator=g).bfloat16().contiguous(memory_format=cl)
    y = torch.empty(B, N, H, W, device=DEV, dtype=torch.bfloat16).contiguous(memory_format=cl)
    hip.conv1x1(rows(xi), wrow(wi), rows(y), conv3=(H, W, d))
    exact = F.conv2d(xi.float(), wi.float(), None, 1, d, d)
    assert torch.equal(y.float(), exact.bfloat16().float())
    x = (torch.randn(B, K, H, W, device=DEV, generator=g) * 1.2 + 0.1).bfloat16().contiguous(memory_format=cl)
    w = (torch.randn(N, K, 3, 3, device=DEV, generator=g) * (2.0 / (9 * K)) ** 0.5).bfloat16().contiguous(memory_format=cl)
    ref = F.conv2d(x.float(), w.float(), None, 1, d, d)
    M = B * H * W
    part = hip.conv1x1_stats_partial(M, N, DEV).fill_(float("nan"))
    hip.conv1x1(rows(x), wrow(w), rows(y), conv3=(H, W, d), out_mode=2, partial=part)
    assert _rel(y, ref) < 3e-3
    buf = torch.zeros(6 * N, device=DEV)
    hip.conv1x1_stats_finalize(part, M, N, None, None, None, 0.1, 1e-5, buf)
    yf = rows(y).float()
    torch.testing.assert_close(buf[3 * N:4 * N], yf.mean(0), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(buf[4 * N:5 * N], 1 / torch.sqrt(yf.var(0, unbiased=False) + 1e-5), rtol=1e-4, atol=1e-6)
    om, osc, osh = torch.randn(N, device=DEV, generator=g) * 0.3, torch.rand(N, device=DEV, generator=g) + 0.5, torch.randn(N, device=DEV, generator=g) * 0.2
    hip.conv1x1(rows(x), wrow(w), rows(y), conv3=(H, W, d), out_mode=1, out_norm=(om, osc, osh, None, hip.ACT_LEAKY_RELU, 0.01))
    assert _rel(y, F.leaky_relu((ref - om.view(1, -1, 1, 1)) * osc.view(1, -1, 1, 1) + osh.view(1, -1, 1, 1), 0.01)) < 3e-3
    # input gradient
    dy = torch.randn(B, N, H, W, device=DEV, generator=g).bfloat16().contiguous(memory_format=cl)
    wt = w.flip(2, 3).transpose(0, 1).contiguous(memory_format=cl)
    dx = torch.empty_like(x)
    hip.conv1x1(rows(dy), wrow(wt), rows(dx), conv3=(H, W, d))
    refdx = torch.nn.grad.conv2d_input(x.shape, w.float(), dy.float(), 1, d, d)
    assert _rel(dx, refdx) < 3e-3


@pytest.mark.parametrize("cin,chans,dil,hw", [(1024, (256, 256, 1024), 1, 33), (256, (64, 64, 256), 1, 65), (512, (128, 128, 512), 2, 49)])
def test_residual_block_training_with_own_3x3(cin, chans, dil, hw):
    """Training forward + backward with the 3x3 convolution + ABN as one node on the implicit-GEMM kernel (forward with the
    statistics epilogue, input gradient on the cached flipped weight) against the module path (MIOpen + separate ABN): wide and
    narrow bottlenecks, B large enough for the own kernel to be chosen.  Both are bf16 implementations of the same block, so
    the yardstick is the fp32 run of the block: the fused path must not be further from it than the module path is (every
    stored map differs by bf16 rounding, which flips the leaky-ReLU branch of the elements nearest zero in each of the
    three normalisations - a few percent on the input gradient for either path)."""
    from functools import partial
    from ucd_amd import abn, blocks
    from ucd_amd.ddp import DistributedDataParallel
    norm = partial(abn.InPlaceABNSync, activation="leaky_relu", activation_param=0.01)
    B, H, W = 24, hw, hw
    assert blocks._own_conv3x3(B * H * W, chans[0], chans[1])
    x0 = synth.t_normal(9, (B, cin, H, W), stream=1).to(DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    dy = synth.t_normal(10, (B, chans[2], H, W), stream=1).to(DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    outs = []
    for mode in ("fused", "module", "fp32"):
        blk = blocks.ResidualBlock(cin, chans, norm_act=norm, stride=1, dilation=dil)
        blk.load_state_dict(synth.fill_state_dict(blk.state_dict(), 5))
        blk = blk.to(DEV).to(memory_format=torch.channels_last).train()
        mod = DistributedDataParallel(blk, bf16_weights=True) if mode != "fp32" else blk
        if mode == "module":
            switches.set("UCD_FUSED_CONV1X1", "0")
        try:
            if mode == "fp32":
                x = x0.float().clone().requires_grad_(True)
                y = mod(x * 1.0)
                y.backward(dy.float())
            else:
                x = x0.clone().requires_grad_(True)
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    y = mod(x * 1.0)
                y.backward(dy)
                mod.finish_grad_sync()
        finally:
            switches.unset("UCD_FUSED_CONV1X1")
        grads = {n: p.grad.float().clone() for n, p in blk.named_parameters()}
        outs.append((y.detach().float(), x.grad.float(), grads, blk.convs.bn2.running_var.clone()))
    (yf, gxf, gf, rvf), (yp, gxp, gp, rvp), (y32, gx32, g32, rv32) = outs
    assert _rel(yf, y32) < 1.5 * _rel(yp, y32) + 2e-3 and _rel(yf, y32) < 2e-2
    assert _rel(gxf, gx32) < 1.5 * _rel(gxp, gx32) + 5e-3, (_rel(gxf, gx32), _rel(gxp, gx32))
    torch.testing.assert_close(rvf, rv32, rtol=5e-3, atol=1e-5)
    for n in gf:
        assert _rel(gf[n], g32[n]) < 1.5 * _rel(gp[n], g32[n]) + 1e-2, (n, _rel(gf[n], g32[n]), _rel(gp[n], g32[n]))


@pytest.mark.usefixtures("deterministic_stats")
@pytest.mark.parametrize("cin,chans,hw", [(1024, (256, 256, 1024), 33), (256, (64, 64, 256), 65)])
def test_backward_link_moves_the_abn_reduction_into_the_input_gradient_product(cin, chans, hw):
    """conv1 + bn1 -> conv2 + bn2 -> conv3 + bn3 of a bottleneck: with the backward link the input-gradient products of conv2
    (3x3) and conv3 (1x1) run in out_mode 3 - activation derivative of the producer's ABN and its two sums in the epilogue -
    and bn1 / bn2 skip their reduction pass.  Same gradients as without the link (UCD_BWD_LINK=0) up to bf16 rounding of one
    intermediate map; the number of ucd_abn_bwd_reduce calls drops from 3 to 1 (counted on the Python twin, which issues the
    same library calls as the C++ node)."""
    from functools import partial
    from ucd_amd import abn, blocks, hip
    from ucd_amd.ddp import DistributedDataParallel
    norm = partial(abn.InPlaceABNSync, activation="leaky_relu", activation_param=0.01)
    B = 24
    x0 = synth.t_normal(9, (B, cin, hw, hw), stream=1).to(DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    dy = synth.t_normal(10, (B, chans[2], hw, hw), stream=1).to(DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    node = blocks._gemm_node()
    res, counts = {}, {}
    real_reduce = hip.abn_bwd_reduce
    for use_node in (True, False):
        for link in ("1", "0"):
            calls = [0]

            def counting(*a, **k):
                calls[0] += 1
                return real_reduce(*a, **k)
            switches.set("UCD_BWD_LINK", link)
            blocks._node_cache[0] = node if use_node else None
            hip.abn_bwd_reduce = counting
            saved_timing = hip._timing
            if not use_node:
                hip._timing = {}                       # the twin's per-kernel path (what bench.py's instrumented pass runs)
            try:
                blk = blocks.ResidualBlock(cin, chans, norm_act=norm, stride=1, dilation=1)
                blk.load_state_dict(synth.fill_state_dict(blk.state_dict(), 5))
                blk = blk.to(DEV).to(memory_format=torch.channels_last).train()
                mod = DistributedDataParallel(blk, bf16_weights=True)
                x = x0.clone().requires_grad_(True)
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    y = mod(x * 1.0)
                y.backward(dy)
                mod.finish_grad_sync()
                res[(use_node, link)] = [y.detach().float(), x.grad.float()] + [p.grad.float().clone() for p in blk.parameters()]
                counts[(use_node, link)] = calls[0]
            finally:
                switches.unset("UCD_BWD_LINK")
                blocks._node_cache[0] = node
                hip.abn_bwd_reduce = real_reduce
                hip._timing = saved_timing
    assert counts[(False, "0")] == 3 and counts[(False, "1")] == 1, counts
    for use_node in (True, False):
        a, b = res[(use_node, "1")], res[(use_node, "0")]
        assert torch.equal(a[0], b[0])                                   # the forward is untouched
        for i in range(1, len(a)):
            assert _rel(a[i], b[i]) < 3e-2, (use_node, i, _rel(a[i], b[i]))
    for i, (a, b) in enumerate(zip(res[(True, "1")], res[(False, "1")])):
        assert _rel(a, b) < 1e-2, i                                       # node and twin agree with the link on


@pytest.mark.usefixtures("deterministic_stats")
@pytest.mark.parametrize("use_node", [True, False])
def test_backward_link_refuses_a_second_consumer(use_node):
    """The link's promise - the linked map feeds exactly ONE consumer - is checked at backward time: when the producer's output
    is read by anything besides the consumer whose input-gradient product served the link, the gradient that arrives is a sum
    made by the autograd engine, not the consumer's dx, and the producer raises instead of applying a half-activated gradient
    (hooks, attention taps, retain_graph replays).  C++ node and Python twin."""
    from functools import partial
    from ucd_amd import abn, blocks
    from ucd_amd.ddp import DistributedDataParallel
    norm = partial(abn.InPlaceABNSync, activation="leaky_relu", activation_param=0.01)
    node = blocks._gemm_node()
    if use_node and node is None:
        pytest.skip("C++ node not built")
    blk = blocks.ResidualBlock(1024, (256, 256, 1024), norm_act=norm, stride=1, dilation=1)
    blk.load_state_dict(synth.fill_state_dict(blk.state_dict(), 5))
    blk = blk.to(DEV).to(memory_format=torch.channels_last).train()
    mod = DistributedDataParallel(blk, bf16_weights=True)
    x = synth.t_normal(9, (24, 1024, 33, 33), stream=1).to(DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True)
    c = blk.convs
    blocks._node_cache[0] = node if use_node else None
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            h1 = blocks._conv_abn_train(c.conv1, c.bn1, x * 1.0, make_link=True)
            assert getattr(h1, "_ucd_link", None) is not None
            early = h1.to(torch.bfloat16) * 1.0                   # a reader of h1 created BEFORE the linked consumer (see below)
            h2 = blocks._conv_abn_train(c.conv2, c.bn2, h1, make_link=True)
            good = h2.float().sum()
            bad = h2.float().sum() + h1.float().mean()            # h1 gains a second consumer
            # a second consumer created BEFORE the linked convolution runs AFTER it in the backward - the engine's input
            # buffer then holds the consumer's dx already and may add the late gradient INTO it in place: same address (the address
            # check alone would pass), bumped version counter - which the producer now checks as well
            bad_early = h2.float().sum() + early.float().sum()
        with pytest.raises(RuntimeError, match="second consumer"):
            bad.backward(retain_graph=True)
        torch.cuda.synchronize()
        x.grad = None
        mod.finish_grad_sync(); mod.zero_grad()                  # (the aborted backward had delivered conv2's gradient already)
        with pytest.raises(RuntimeError, match="second consumer"):
            bad_early.backward(retain_graph=True)
        torch.cuda.synchronize()
        mod.finish_grad_sync(); mod.zero_grad()
        # the flag was cleared with the error: the same graph without the extra reader is fine afterwards
        x.grad = None
        good.backward()
        torch.cuda.synchronize()
        assert torch.isfinite(x.grad.float()).all()
    finally:
        blocks._node_cache[0] = node
        mod.finish_grad_sync()


@pytest.mark.usefixtures("deterministic_stats")
def test_backward_link_parity_through_the_model_with_intermediate_features_read():
    """UCD_BWD_LINK=0/1 through the whole student with --loss_de on (ret_intermediate: features["body"] and ["pre_logits"] are
    read by a second loss, train.py:118-121): the link only ever spans maps that stay inside a bottleneck, so the extra readers
    of the block OUTPUTS do not touch it - same losses, same gradients up to bf16 rounding of one intermediate map."""
    from ucd_amd import argparser, tasks
    from ucd_amd.run import build_models, load_step_checkpoint, make_optimizer
    from ucd_amd.train import Trainer
    img = synth.images(511, 2, 257)
    labels = synth.seg_labels(511, 2, 257, 257, range(16, 21))
    res = {}
    torch.backends.cudnn.deterministic = True
    try:
        for link in ("1", "0"):
            switches.set("UCD_BWD_LINK", link)
            opts = argparser.modify_command_options(argparser.get_argparser().parse_args(
                ["--method", "UCD", "--task", "15-5", "--step", "1", "--lr", "0.001", "--no_pretrained", "--norm_act", "iabn_sync",
                 "--opt_level", "O1", "--loss_de", "1"]))
            classes = tasks.get_per_task_classes("voc", "15-5", 1)
            dev = torch.device(DEV)
            model, model_old = build_models(opts, dev, classes)
            state = synth.fill_state_dict({k: v.cpu() for k, v in model_old.state_dict().items()}, 42, calibrated=True)
            load_step_checkpoint(opts, model, model_old, state, dev)
            trainer = Trainer(model, model_old, device=dev, opts=opts, classes=classes)
            optim = make_optimizer(opts, model)
            model.train()
            r = trainer.train_step(img, labels, optim, None)
            torch.cuda.synchronize()
            res[link] = ({k: v.item() for k, v in r.items()},
                         {n: p.grad.float().clone() for n, p in model.named_parameters() if p.grad is not None})
    finally:
        switches.unset("UCD_BWD_LINK")
        torch.backends.cudnn.deterministic = False
    (la, ga), (lb, gb) = res["1"], res["0"]
    assert la["lde"] > 0
    for k in la:
        assert la[k] == pytest.approx(lb[k], rel=1e-6), k                 # the forward is untouched by the link
    rels = sorted(((_rel(ga[n], gb[n]), n) for n in ga if gb[n].abs().max() > 0), reverse=True)
    print("link on/off through the model: largest gradient differences", rels[:4], "median", rels[len(rels) // 2])
    # two bf16 backward passes that round one map per block differently (d out vs d out * act'), 33 blocks deep
    assert rels[0][0] < 0.1 and rels[len(rels) // 2][0] < 5e-2, rels[:4]


# ---- the benchmark's own layer shapes (B = 24 at 513^2): M = 399 384 (129^2), 101 400 (65^2), 26 136 (33^2) ----------------------
BENCH_LAYERS = [
    # M, K, N, (H, W, dilation) or None
    (399384, 64, 256, None), (399384, 256, 64, None), (399384, 64, 64, (129, 129, 1)),          # mod2: 3121 row tiles
    (101400, 512, 128, None), (101400, 128, 512, None), (101400, 128, 128, (65, 65, 1)),        # mod3
    (26136, 1024, 256, None), (26136, 256, 1024, None), (26136, 256, 256, (33, 33, 1)),         # mod4
    (26136, 2048, 512, None), (26136, 512, 512, (33, 33, 2)), (26136, 2048, 256, (33, 33, 12)),  # mod5, an ASPP branch
]


def _conv3x3_rows_fp32(a, w9, B, H, W, d):
    """Exact fp32 reference of the 3x3 (padding = dilation d) on a row matrix: nine shifted products (plain matmuls only: no
    library convolution whose transform might not be exact on integers).  ``w9`` [N, 9 K] in (kh, kw, k) order."""
    M, K = a.shape
    N = w9.shape[0]
    x = a.float().view(B, H, W, K)
    xp = F.pad(x, (0, 0, d, d, d, d))
    out = torch.zeros(M, N, device=a.device)
    for kh in range(3):
        for kw in range(3):
            sl = xp[:, kh * d:kh * d + H, kw * d:kw * d + W, :].reshape(M, K)
            out += sl @ w9[:, (kh * 3 + kw) * K:(kh * 3 + kw + 1) * K].float().t()
    return out


@pytest.mark.parametrize("M,K,N,sp", BENCH_LAYERS)
def test_bench_shape_products_are_exact_on_integers_and_statistics_hold(M, K, N, sp):
    """Every product form the benchmark step launches, AT the benchmark's row counts (the XCD-aware tile mapping with 3121 /
    793 / 205 row tiles, the clamped last tile, the double-buffered and single-stage forms the host picks per grid): sparse
    small-integer operands make every output an exactly representable integer, so the comparison is bit-exact; then the
    statistics epilogue + finalize and the activation-backward epilogue + partial reduction against torch on the stored map."""
    from ucd_amd import hip
    g = torch.Generator(DEV).manual_seed(M + K + N)
    taps = 9 if sp else 1
    dens = min(0.5, 24.0 / (K * taps))                      # ~24 non-zero products per output: |sum| stays far below 256
    ai = (torch.randint(-1, 2, (M, K), device=DEV, generator=g) * (torch.rand(M, K, device=DEV, generator=g) < dens ** 0.5)).bfloat16()
    wi = (torch.randint(-2, 3, (N, taps * K), device=DEV, generator=g) * (torch.rand(N, taps * K, device=DEV, generator=g) < dens ** 0.5)).bfloat16()
    y = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    if sp:
        H, W, d = sp
        B = M // (H * W)
        hip.conv1x1(ai, wi, y, conv3=(H, W, d))
        exact = _conv3x3_rows_fp32(ai, wi, B, H, W, d)
    else:
        hip.conv1x1(ai, wi, y)
        exact = ai.float() @ wi.float().t()
    assert exact.abs().max().item() <= 256
    assert torch.equal(y.float(), exact)
    # statistics epilogue (out_mode 2) on real-valued operands: the stored map's mean / variance
    a = (torch.randn(M, K, device=DEV, generator=g) * 1.3 + 0.2).bfloat16()
    w = (torch.randn(N, taps * K, device=DEV, generator=g) * (2.0 / (K * taps)) ** 0.5).bfloat16()
    part = hip.conv1x1_stats_partial(M, N, DEV).fill_(float("nan"))
    hip.conv1x1(a, w, y, out_mode=2, partial=part, conv3=sp)
    ref = _conv3x3_rows_fp32(a, w, M // (sp[0] * sp[1]), *sp) if sp else a.float() @ w.float().t()
    assert _rel(y, ref) < 3e-3
    buf = torch.zeros(6 * N, device=DEV)
    rm, rv = torch.zeros(N, device=DEV), torch.ones(N, device=DEV)
    gamma = torch.rand(N, device=DEV, generator=g) + 0.5
    hip._check(hip.load().ucd_conv1x1_stats_finalize(hip.ptr(part), M, N, hip.ptr(gamma), hip.ptr(rm), hip.ptr(rv), 0.1, 1e-5,
                                                     hip.ptr(buf), None, hip.NORM_ABS_GAMMA, hip.stream()), "finalize")
    yf = y.double()
    mean, var = yf.mean(0), yf.var(0, unbiased=False)
    torch.testing.assert_close(buf[3 * N:4 * N].double(), mean, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(buf[4 * N:5 * N].double(), 1 / torch.sqrt(var + 1e-5), rtol=1e-4, atol=1e-6)
    # activation-backward epilogue (out_mode 3) + partial reduction: dz and its two sums
    r = torch.randn(M, N, device=DEV, generator=g).bfloat16()
    om, osc, osh, oinv = (torch.randn(N, device=DEV, generator=g) * 0.3, torch.rand(N, device=DEV, generator=g) + 0.5,
                          torch.randn(N, device=DEV, generator=g) * 0.2, torch.rand(N, device=DEV, generator=g) + 0.5)
    tiles = hip.load().ucd_conv1x1_row_tiles(M)
    part2 = torch.full((tiles, 2, N), float("nan"), device=DEV)
    hip.conv1x1(a, w, y, out_mode=3, out_norm=(om, osc, osh, oinv, hip.ACT_LEAKY_RELU, 0.01), residual=r, partial=part2, conv3=sp)
    z = (r.float() - om) * osc + osh
    assert _rel(y, ref * torch.where(z > 0, 1.0, 0.01)) < 3e-3
    sums = torch.zeros(2 * N, device=DEV)
    hip._check(hip.load().ucd_abn_reduce_partials(hip.ptr(part2), tiles, N, hip.ptr(sums), None, hip.ptr(gamma),
                                                  hip.NORM_ABS_GAMMA, hip.stream()), "reduce")
    dz = y.double()
    torch.testing.assert_close(sums[:N].double(), dz.sum(0), rtol=2e-4, atol=2e-2)
    torch.testing.assert_close(sums[N:].double(), (dz * ((r.double() - om) * oinv)).sum(0), rtol=2e-4, atol=5e-2)


def _worst_param_grad(gf, g32, zero_in_exact_arithmetic=()):
    """Largest relative L2 over the parameter gradients that ARE gradients: behind a batch-statistics norm the gradient of a
    per-channel shift is zero in exact arithmetic (a bias whose layer feeds a 1x1 convolution and then a training-mode norm, in an
    identity-activation chain: ``zero_in_exact_arithmetic`` names them - both sides hold rounding noise there, the independent
    fp32 reference's is simply another noise), and a relative error on rounding noise says nothing - those and parameters whose
    fp32 gradient is below 1e-3 of the largest (rms) are left out."""
    rms = {n: g32[n].pow(2).mean().sqrt().item() for n in g32}
    top = max(rms.values())
    rel = {n: _rel(gf[n], g32[n]) for n in gf if rms[n] > 1e-3 * top and n not in zero_in_exact_arithmetic}
    worst = max(rel, key=rel.get)
    print("parameter gradients (relative L2):", {n: round(v, 5) for n, v in sorted(rel.items(), key=lambda kv: -kv[1])[:4]}, "worst:", worst)
    return rel[worst]


@pytest.mark.parametrize("cin,chans,dil,hw,slope", [(256, (64, 64, 256), 1, 129, 1.0), (256, (64, 64, 256), 1, 129, 0.01),
                                                    (512, (128, 128, 512), 1, 65, 1.0), (1024, (256, 256, 1024), 1, 33, 1.0),
                                                    (1024, (256, 256, 1024), 1, 33, 0.01), (2048, (512, 512, 2048), 2, 33, 1.0)])
def test_bench_shape_block_chain_fused_against_fp32_layer_by_layer(cin, chans, dil, hw, slope):
    """One identity-shortcut bottleneck of mod2 / mod3 / mod4 / mod5 at the benchmark's batch (B = 24: M = 399 384 / 101 400 /
    26 136) as the chain of conv+ABN nodes the benchmark step runs (statistics epilogues, backward links, shortcut fold, weight
    gradients) against an INDEPENDENT fp32 reference: the oracle's functional block (oracle/model.py::residual_block - stock
    F.conv2d / F.batch_norm / F.leaky_relu on cuda tensors, no product code), so the 205- / 793- / 3121-tile grids
    and the bench-size kernel forms meet something that is not the product.
    slope = 1 (leaky_relu(1.0) = identity through the same kernels): the arithmetic of the chain alone - output, input gradient
    and every parameter gradient within 1e-2 / 2.5e-2 in relative L2 (each stored map carries one bf16 rounding, 2^-9).
    slope = 0.01 (the network's): a stored map's rounding flips the leaky-ReLU branch of the elements nearest zero - a fraction
    p ~ 0.8 * 2^-9 = 1.6e-3 of them per activation, each with a 99 % error on its gradient, i.e. sqrt(3 p) ~ 7 % in relative L2
    on the input gradient of a three-activation block for ANY bf16 implementation (measured 7.0-7.1 % on all four shapes, the
    library-kernel path the same).  Asserted: L2 below 0.1 - the slope-1 run is what holds the kernels' arithmetic tightly."""
    from functools import partial
    from ucd_amd import abn, blocks
    from ucd_amd.ddp import DistributedDataParallel
    norm = partial(abn.InPlaceABNSync, activation="leaky_relu", activation_param=slope)
    B = 24
    x0 = synth.t_normal(9, (B, cin, hw, hw), stream=1).to(DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    dy = synth.t_normal(10, (B, chans[2], hw, hw), stream=1).to(DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    from oracle import model as OM
    outs = []
    blk = blocks.ResidualBlock(cin, chans, norm_act=norm, stride=1, dilation=dil)
    state = synth.fill_state_dict(blk.state_dict(), 5)
    blk.load_state_dict(state)
    blk = blk.to(DEV).to(memory_format=torch.channels_last).train()
    mod = DistributedDataParallel(blk, bf16_weights=True)
    x = x0.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = mod(x * 1.0)
    y.backward(dy)
    mod.finish_grad_sync()
    outs.append((y.detach().float(), x.grad.float(), {n: p.grad.float().clone() for n, p in blk.named_parameters()},
                 blk.convs.bn2.running_var.clone(), blk.convs.bn3.running_mean.clone()))
    del blk, mod, x, y
    torch.cuda.empty_cache()
    # the independent leg: the oracle's functional block on the device, parameters by the reference's state_dict names - in float64
    # on torch's native kernels (MIOpen off): MIOpen's fp32 batch-norm backward is NOT a usable reference here - its d bias of bn3,
    # which is nothing but the column sums of dy, is off by 1.5 % (M = 399 384) to 5.4 % (M = 26 136) from those sums taken directly,
    # the product's by 9e-8 (tests/diag/bn3_bias_diag.py, profiles/r05_bn3_bias_diag.txt)
    P = {"blk." + k: (v.to(DEV).double() if v.is_floating_point() else v.to(DEV)) for k, v in state.items()}
    for k, v in P.items():
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    with torch.backends.cudnn.flags(enabled=False):
        x = x0.double().contiguous().clone().requires_grad_(True)
        y = OM.residual_block(x * 1.0, P, "blk", 1, dil, True, slope=slope)
        y.backward(dy.double().contiguous())
    outs.append((y.detach().float(), x.grad.float(), {k[4:]: v.grad.float() for k, v in P.items() if v.requires_grad},
                 P["blk.convs.bn2.running_var"].float(), P["blk.convs.bn3.running_mean"].float()))
    del P, x, y
    torch.cuda.empty_cache()
    (yf, gxf, gf, rvf, rmf), (y32, gx32, g32, rv32, rm32) = outs
    assert set(gf) == set(g32)
    # slope 1 (no activation): a shift of bn2's output passes conv3 (1x1: the same constant on every pixel) and is removed by bn3's
    # batch mean - d bn2.bias is zero in exact arithmetic; a shift of bn1's output meets conv2's zero padding, so d bn1.bias is the
    # (small) border term only: it is held on its own, against the scale of the norm's other gradient
    zero = ("convs.bn2.bias",) if slope == 1.0 else ()
    border = ("convs.bn1.bias",) if slope == 1.0 else ()
    worst = _worst_param_grad({n: v for n, v in gf.items() if n not in border}, {n: v for n, v in g32.items() if n not in border}, zero)
    for n in border:
        err = (gf[n] - g32[n]).pow(2).mean().sqrt().item()
        assert err < 2e-2 * g32["convs.bn1.weight"].pow(2).mean().sqrt().item(), (n, err)
    l2 = _rel(gxf, gx32)
    print("bench-shape chain", cin, chans, hw, "slope", slope, "y", _rel(yf, y32), "dx", l2, "worst param grad", worst)
    assert _rel(yf, y32) < 1e-2
    torch.testing.assert_close(rvf, rv32, rtol=5e-3, atol=1e-5)
    torch.testing.assert_close(rmf, rm32, rtol=5e-3, atol=2e-3)
    if slope == 1.0:
        # measured against the float64 reference: dx 3.7e-3 on all four shapes, every parameter gradient <= 5.3e-3 (one bf16 rounding
        # per stored map); pinned at about twice that - NOT to be loosened to follow a kernel change
        assert l2 < 6e-3, l2
        assert worst < 1e-2, worst
    else:
        assert l2 < 0.1, l2
        assert worst < 0.13, worst


@pytest.mark.parametrize("slope", [1.0, 0.01])
def test_bench_shape_aspp_head_fused_against_fp32(slope):
    """DeeplabV3 in training mode at the benchmark's shape (B = 24, 2048 x 33 x 33): four branches on the own kernels (1x1 and
    three dilated 3x3 implicit GEMMs), map_bn over channel slices, red_conv + statistics, the pooled branch as a plane bias -
    against the oracle's functional head in fp32 on the device (oracle/model.py::deeplab_head: stock torch operators, no product
    code); slopes as in the block test (two activations here: sqrt(2 p) ~ 6 %)."""
    from functools import partial
    from ucd_amd import abn, blocks
    from ucd_amd.ddp import DistributedDataParallel
    norm = partial(abn.InPlaceABNSync, activation="leaky_relu", activation_param=slope)
    B, C, hw = 24, 2048, 33
    x0 = synth.t_normal(19, (B, C, hw, hw), stream=1).to(DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    dy = synth.t_normal(20, (B, 256, hw, hw), stream=1).to(DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    from oracle import model as OM
    outs = []
    head = blocks.DeeplabV3(C, 256, 256, norm_act=norm, out_stride=16, pooling_size=32)
    state = synth.fill_state_dict(head.state_dict(), 21)
    head.load_state_dict(state)
    head = head.to(DEV).to(memory_format=torch.channels_last).train()
    mod = DistributedDataParallel(head, bf16_weights=True)
    x = x0.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = mod(x * 1.0)
    y.backward(dy)
    mod.finish_grad_sync()
    outs.append((y.detach().float(), x.grad.float(), {n: p.grad.float().clone() for n, p in head.named_parameters()}))
    del head, mod, x, y
    torch.cuda.empty_cache()
    # float64 on torch's native kernels (see the block test: MIOpen's fp32 batch-norm backward is no reference)
    P = {"head." + k: (v.to(DEV).double() if v.is_floating_point() else v.to(DEV)) for k, v in state.items()}
    for k, v in P.items():
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    with torch.backends.cudnn.flags(enabled=False):
        x = x0.double().contiguous().clone().requires_grad_(True)
        y = OM.deeplab_head(x * 1.0, P, True, prefix="head.", pooling_size=32, slope=slope)
        y.backward(dy.double().contiguous())
    outs.append((y.detach().float(), x.grad.float(), {k[5:]: v.grad.float() for k, v in P.items() if v.requires_grad}))
    del P, x, y
    torch.cuda.empty_cache()
    (yf, gxf, gf), (y32, gx32, g32) = outs
    assert set(gf) == set(g32)
    # slope 1: map_bn's and the pooled branch's shifts pass a 1x1 convolution and are removed by red_bn's batch mean (zero gradients)
    worst = _worst_param_grad(gf, g32, ("map_bn.bias", "global_pooling_bn.bias") if slope == 1.0 else ())
    l2 = _rel(gxf, gx32)
    print("bench-shape ASPP slope", slope, "y", _rel(yf, y32), "dx", l2, "worst param grad", worst)
    assert _rel(yf, y32) < 1e-2
    if slope == 1.0:
        assert l2 < 1e-2 and worst < 1e-2, (l2, worst)       # measured 5.5e-3 / 4.7e-3 against the float64 reference
    else:
        assert l2 < 0.1 and worst < 0.1, (l2, worst)


@pytest.mark.usefixtures("deterministic_stats")
@pytest.mark.parametrize("cin,chans,hw,dil", [(512, (256, 256, 1024), 33, 1), (256, (64, 64, 256), 65, 1), (1024, (512, 512, 2048), 33, 2)])
def test_block_link_moves_bn3_backward_into_the_next_blocks_first_product(cin, chans, hw, dil):
    """Three bottlenecks in a row (projection block, identity, identity): with the block link the input-gradient product of an
    identity block's conv1 - the one that folds the shortcut's gradient in, i.e. the product that FORMS the gradient w.r.t. the
    previous block's output - runs in out_mode 4: block activation derivative (sign of that output) and the two backward sums
    of the previous block's bn3 in its epilogue.  bn3 then skips its reduction pass and hands d pre on as the shortcut's
    gradient (no second tensor).  Same gradients as with UCD_BLOCK_LINK=0 up to bf16 rounding of one map; ucd_abn_bwd_reduce
    calls drop by two (from 4 = three bn3 + one proj_bn on the mod4 shape); C++ node and Python twin agree."""
    from functools import partial
    from ucd_amd import abn, blocks, hip
    from ucd_amd.ddp import DistributedDataParallel
    norm = partial(abn.InPlaceABNSync, activation="leaky_relu", activation_param=0.01)
    B = 24
    x0 = synth.t_normal(9, (B, cin, hw, hw), stream=1).to(DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    dy = synth.t_normal(10, (B, chans[2], hw, hw), stream=1).to(DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    node = blocks._gemm_node()
    res, counts = {}, {}
    real_reduce = hip.abn_bwd_reduce
    for use_node in (True, False):
        for link in ("1", "0"):
            calls = [0]

            def counting(*a, **k):
                calls[0] += 1
                return real_reduce(*a, **k)
            switches.set("UCD_BLOCK_LINK", link)
            blocks._node_cache[0] = node if use_node else None
            hip.abn_bwd_reduce = counting
            saved_timing = hip._timing
            if not use_node:
                hip._timing = {}
            try:
                net = torch.nn.Sequential(blocks.ResidualBlock(cin, chans, norm_act=norm, stride=1, dilation=dil),
                                          blocks.ResidualBlock(chans[2], chans, norm_act=norm, stride=1, dilation=dil),
                                          blocks.ResidualBlock(chans[2], chans, norm_act=norm, stride=1, dilation=dil))
                net.load_state_dict(synth.fill_state_dict(net.state_dict(), 5))
                net = net.to(DEV).to(memory_format=torch.channels_last).train()
                mod = DistributedDataParallel(net, bf16_weights=True)
                x = x0.clone().requires_grad_(True)
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    y = mod(x * 1.0)
                y.backward(dy)
                mod.finish_grad_sync()
                res[(use_node, link)] = [y.detach().float(), x.grad.float()] + [p.grad.float().clone() for p in net.parameters()]
                counts[(use_node, link)] = calls[0]
            finally:
                switches.unset("UCD_BLOCK_LINK")
                blocks._node_cache[0] = node
                hip.abn_bwd_reduce = real_reduce
                hip._timing = saved_timing
    # the two bn3 in front of an identity block lose their reduction pass (the other reductions of a chain - the last bn3,
    # proj_bn, and bn1 / bn2 where the 3x3 stays with MIOpen - are not the block link's)
    assert counts[(False, "0")] - counts[(False, "1")] == 2, counts
    if chans[0] == 256:
        assert counts[(False, "0")] == 4, counts
    for use_node in (True, False):
        a, b = res[(use_node, "1")], res[(use_node, "0")]
        assert torch.equal(a[0], b[0])                                   # the forward is untouched
        worst = max(_rel(a[i], b[i]) for i in range(1, len(a)) if b[i].abs().max() > 0)
        print("block link", cin, chans, hw, "node" if use_node else "twin", "worst gradient difference on / off", worst)
        assert worst < 3e-2, (use_node, worst)
    for i, (a, b) in enumerate(zip(res[(True, "1")], res[(False, "1")])):
        assert _rel(a, b) < 1e-2, i                                       # node and twin agree with the link on


# ---- weight gradients (csrc/wgrad.hip) ----------------------------------------------------------------------------------------
WGRAD_CASES = [
    # M, N (out), K (in), (H, W, dilation) or None
    (300, 64, 64, None), (1000, 128, 256, None), (4356, 256, 128, (33, 33, 2)), (2 * 9 * 11, 64, 128, (9, 11, 1)),
    (26136, 256, 1024, None), (26136, 1024, 256, None), (26136, 256, 256, (33, 33, 1)), (26136, 512, 512, (33, 33, 2)),
    (26136, 256, 2048, (33, 33, 12)), (101400, 128, 512, None), (101400, 128, 128, (65, 65, 1)),
    (399384, 256, 64, None), (399384, 64, 256, None), (399384, 64, 64, (129, 129, 1)),
]


def _wgrad_ref(dz, x, sp):
    """fp32 reference by plain matmuls: dw[n, t, k] = sum_m dz[m, n] x[shift_t(m), k] (zero outside the map)."""
    M, N = dz.shape
    K = x.shape[1]
    if sp is None:
        return dz.float().t() @ x.float()
    H, W, d = sp
    B = M // (H * W)
    xp = F.pad(x.float().view(B, H, W, K), (0, 0, d, d, d, d))
    out = torch.empty(N, 9, K, device=dz.device)
    for kh in range(3):
        for kw in range(3):
            out[:, kh * 3 + kw] = dz.float().t() @ xp[:, kh * d:kh * d + H, kw * d:kw * d + W, :].reshape(M, K)
    return out.reshape(N, 9 * K)


@pytest.mark.parametrize("M,N,K,sp", WGRAD_CASES)
def test_conv_wgrad_exact_on_integers_and_against_fp32(M, N, K, sp):
    """dW = dZ^T X (1x1) and the 9-tap form of the 3x3 layers through ucd_conv_wgrad at the network's shapes (B = 24 row counts
    included: the chunk / XCD mapping, the chunk that ends inside a 64-row step, rows that shift off the map): sparse
    small-integer operands make every output an exactly representable integer, so the comparison is bit-exact in bf16 AND in the
    fp32 output; then random operands against the fp32 product at bf16 rounding, and the += form into an fp32 buffer."""
    from ucd_amd import hip
    g = torch.Generator(DEV).manual_seed(M + N + K)
    dens = min(0.5, (24.0 / M) ** 0.5)
    zi = (torch.randint(-2, 3, (M, N), device=DEV, generator=g) * (torch.rand(M, N, device=DEV, generator=g) < dens)).bfloat16()
    xi = (torch.randint(-1, 2, (M, K), device=DEV, generator=g) * (torch.rand(M, K, device=DEV, generator=g) < dens)).bfloat16()
    taps = 9 if sp else 1
    dw = torch.full((N, taps * K), float("nan"), device=DEV, dtype=torch.bfloat16)
    dw32 = torch.full((N, taps * K), float("nan"), device=DEV)
    hip.conv_wgrad(zi, xi, dw, conv3=sp, dw32=dw32)
    exact = _wgrad_ref(zi, xi, sp)
    assert exact.abs().max().item() <= 256
    assert torch.equal(dw32, exact)
    assert torch.equal(dw.float(), exact)
    z = torch.randn(M, N, device=DEV, generator=g).bfloat16()
    x = (torch.randn(M, K, device=DEV, generator=g) * 1.3 + 0.2).bfloat16()
    ref = _wgrad_ref(z, x, sp)
    hip.conv_wgrad(z, x, dw, conv3=sp)
    assert _rel(dw, ref) < 3e-3
    base = torch.randn(N, taps * K, device=DEV, generator=g)
    acc = base.clone()
    hip.conv_wgrad(z, x, None, conv3=sp, dw32=acc, accumulate32=True)
    assert _rel(acc - base, ref) < 1e-4


@pytest.mark.parametrize("M,N,K,sp", [(26136, 256, 256, (33, 33, 1)), (26136, 512, 512, (33, 33, 2)), (26136, 256, 2048, (33, 33, 12)),
                                      (23 * 33 * 33, 128, 128, (33, 33, 18)), (24 * 65 * 65, 128, 128, (65, 65, 1))])
def test_three_tap_and_nine_tap_weight_gradients_agree(M, N, K, sp, monkeypatch):
    """The 3x3 weight gradient has two kernels since round 4: one kernel ROW per workgroup (wgrad3_kernel: shared dZ tile, X tile of
    64 + 2 d rows, row-pair masks for the horizontal border, loader waves) and the 9-tap form it replaced (UCD_WGRAD3=0, kept for
    small maps and as the A/B reference).  On sparse small integers both are exact, so they agree bit for bit - in every tap, at
    the image borders (masks), at the chunk ends and for the dilations of the network (1, 2, 12, 18)."""
    from ucd_amd import hip
    g = torch.Generator(DEV).manual_seed(M + N + K + sp[2])
    dens = min(0.5, (24.0 / M) ** 0.5)
    zi = (torch.randint(-2, 3, (M, N), device=DEV, generator=g) * (torch.rand(M, N, device=DEV, generator=g) < dens)).bfloat16()
    xi = (torch.randint(-1, 2, (M, K), device=DEV, generator=g) * (torch.rand(M, K, device=DEV, generator=g) < dens)).bfloat16()
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("UCD_WGRAD3", mode)
        dw32 = torch.full((N, 9 * K), float("nan"), device=DEV)
        hip.conv_wgrad(zi, xi, None, conv3=sp, dw32=dw32)
        out[mode] = dw32
    exact = _wgrad_ref(zi, xi, sp)
    assert torch.equal(out["1"], exact) and torch.equal(out["0"], exact)
    z = torch.randn(M, N, device=DEV, generator=g).bfloat16()
    x = (torch.randn(M, K, device=DEV, generator=g) * 1.3 + 0.2).bfloat16()
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("UCD_WGRAD3", mode)
        res[mode] = hip.conv_wgrad(z, x, None, conv3=sp, dw32=torch.empty(N, 9 * K, device=DEV)).clone()
    assert _rel(res["1"], res["0"].float()) < 1e-5          # fp32 sums of the same products in another order


# ---- the strided layers (first block of a stage: conv2 3x3 stride 2, proj_conv 1x1 stride 2; modules/residual.py:57-82) -------
STRIDED_CASES = [
    # B, H, W, K (in), N (out), taps, dilation
    (2, 9, 11, 64, 64, 1, 1), (2, 9, 11, 128, 128, 9, 1), (3, 10, 8, 128, 256, 9, 1), (24, 129, 129, 256, 512, 1, 1),
    (24, 65, 65, 512, 1024, 1, 1), (24, 129, 129, 128, 128, 9, 1), (24, 65, 65, 256, 256, 9, 1), (3, 65, 65, 256, 256, 9, 2),
]


def _strided_ref(x4, w4, taps, d):
    """fp32 F.conv2d of the NCHW views (padding 0 for 1x1, = dilation for 3x3), rows of the channels-last result."""
    y = F.conv2d(x4.float(), w4.float(), None, 2, d if taps == 9 else 0, d if taps == 9 else 1)
    return y.permute(0, 2, 3, 1).reshape(-1, y.shape[1]), y.shape[2], y.shape[3]


@pytest.mark.parametrize("B,H,W,K,N,taps,d", STRIDED_CASES)
def test_strided_forward_statistics_and_weight_gradient(B, H, W, K, N, taps, d):
    """ucd_conv1x1 with stride 2 (the 1x1 row gather and the 3x3 implicit GEMM over the strided pixels) and
    ucd_conv_wgrad_strided against F.conv2d / its weight gradient: exact on sparse small integers (every sum an exactly
    representable integer), then at bf16 rounding on random operands; the out_mode-2 partials give the batch statistics of the
    strided product."""
    from ucd_amd import hip
    g = torch.Generator(DEV).manual_seed(B * H + K + N + taps)
    k = 3 if taps == 9 else 1
    cl = torch.channels_last

    def operands(integer):
        if integer:
            x = (torch.randint(-2, 3, (B, K, H, W), device=DEV, generator=g) * (torch.rand(B, K, H, W, device=DEV, generator=g) < 0.2))
            w = (torch.randint(-1, 2, (N, K, k, k), device=DEV, generator=g) * (torch.rand(N, K, k, k, device=DEV, generator=g) < 0.3))
        else:
            x = torch.randn(B, K, H, W, device=DEV, generator=g) * 1.3 + 0.2
            w = torch.randn(N, K, k, k, device=DEV, generator=g) * (taps * K) ** -0.5
        return x.bfloat16().contiguous(memory_format=cl), w.bfloat16().contiguous(memory_format=cl)

    for integer in (True, False):
        x4, w4 = operands(integer)
        ref, OH, OW = _strided_ref(x4, w4, taps, d)
        M = B * OH * OW
        xr = x4.permute(0, 2, 3, 1).reshape(B * H * W, K)
        wr = w4.permute(0, 2, 3, 1).reshape(N, taps * K)
        y = torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16)
        partial = hip.conv1x1_stats_partial(M, N, DEV)
        kw = dict(conv3=(H, W, d, 2)) if taps == 9 else dict(strided=(H, W, 2))
        hip.conv1x1(xr, wr, y, out_mode=2, partial=partial, **kw)
        if integer:
            assert ref.abs().max().item() <= 256
            assert torch.equal(y.float(), ref)
        else:
            assert _rel(y, ref) < 4e-3
        buf = torch.zeros(6 * N, device=DEV)
        rm, rv = torch.zeros(N, device=DEV), torch.ones(N, device=DEV)
        hip.conv1x1_stats_finalize(partial, M, N, None, rm, rv, 0.1, 1e-5, buf)
        yf = y.float()
        assert torch.allclose(buf[3 * N:4 * N], yf.mean(0), atol=2e-3 * max(1.0, yf.abs().max().item()), rtol=1e-3)
        assert torch.allclose(buf[4 * N:5 * N], (yf.var(0, unbiased=False) + 1e-5).rsqrt(), rtol=2e-3)
        # weight gradient: dz over the output map, x the input map
        if N % 128 or K % 128:
            continue
        if integer:
            dz = (torch.randint(-2, 3, (M, N), device=DEV, generator=g) * (torch.rand(M, N, device=DEV, generator=g) < min(0.5, (24.0 / M) ** 0.5))).bfloat16()
        else:
            dz = torch.randn(M, N, device=DEV, generator=g).bfloat16()
        dz4 = dz.view(B, OH, OW, N).permute(0, 3, 1, 2)
        pad, dil = (d, d) if taps == 9 else (0, 1)
        gw = torch.ops.aten.convolution_backward(dz4.float(), x4.float(), w4.float(), None, [2, 2], [pad, pad], [dil, dil], False, [0, 0],
                                                 1, [False, True, False])[1]
        gw = gw.permute(0, 2, 3, 1).reshape(N, taps * K)
        dw = torch.full((N, taps * K), float("nan"), device=DEV, dtype=torch.bfloat16)
        dw32 = torch.full((N, taps * K), float("nan"), device=DEV)
        hip.conv_wgrad(dz, xr, dw, dw32=dw32, **kw)
        if integer:
            assert gw.abs().max().item() <= 256
            assert torch.equal(dw32, gw) and torch.equal(dw.float(), gw)
        else:
            assert _rel(dw, gw) < 3e-3 and _rel(dw32, gw) < 1e-4


@pytest.mark.usefixtures("deterministic_stats")
@pytest.mark.parametrize("cin,chans,stride,hw", [(256, (128, 128, 512), 2, 65), (1024, (512, 512, 2048), 1, 33), (64, (64, 64, 256), 1, 65)])
def test_projection_block_alias_equals_the_separate_gradient_add(cin, chans, stride, hw):
    """Projection blocks (stride and / or channel change; modules/residual.py:79-87): with UCD_PROJ_ALIAS (default) proj_conv reads an
    alias of the block input that conv1's node returns, so the block input has one consumer, the projection's input gradient is the
    accumulate operand of conv1's input-gradient product and the block in front can offer its block link.  Same forward bit for
    bit and the same gradients (up to the bf16 rounding of one sum) as with the two consumers and autograd's add."""
    from functools import partial
    from ucd_amd import abn, blocks
    from ucd_amd.ddp import DistributedDataParallel
    norm = partial(abn.InPlaceABNSync, activation="leaky_relu", activation_param=0.01)
    B = 24 if hw <= 33 else 6
    front = (chans[2] // 4 if stride == 1 and cin == chans[2] else max(64, cin // 4))
    x0 = synth.t_normal(21, (B, cin, hw, hw), stream=1).to(DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    res = {}
    for alias in ("1", "0"):
        switches.set("UCD_PROJ_ALIAS", alias)
        try:
            # an identity block in front (its bn3 offers the block link), the projection block, an identity block behind
            net = torch.nn.Sequential(blocks.ResidualBlock(cin, (front, front, cin), norm_act=norm),
                                      blocks.ResidualBlock(cin, chans, norm_act=norm, stride=stride),
                                      blocks.ResidualBlock(chans[2], chans, norm_act=norm))
            net.load_state_dict(synth.fill_state_dict(net.state_dict(), 7))
            net = net.to(DEV).to(memory_format=torch.channels_last).train()
            mod = DistributedDataParallel(net, bf16_weights=True)
            x = x0.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = mod(x * 1.0)
            dy = synth.t_normal(22, tuple(y.shape), stream=1).to(DEV).bfloat16().contiguous(memory_format=torch.channels_last)
            y.backward(dy)
            mod.finish_grad_sync()
            res[alias] = [y.detach().float(), x.grad.float()] + [p.grad.float().clone() for p in net.parameters()]
        finally:
            switches.unset("UCD_PROJ_ALIAS")
    a, b = res["1"], res["0"]
    assert torch.equal(a[0], b[0])
    worst = max(_rel(a[i], b[i]) for i in range(1, len(a)) if b[i].abs().max() > 0)
    print("projection alias", cin, chans, stride, hw, "worst gradient difference on / off", worst)
    assert worst < 3e-2, worst


# ---- every pipeline form of the GEMM kernel (UCD_CONV_PIPE is read once per process: one child process per form) ----------------------
_PIPE_CHILD = r"""
import sys, torch
sys.path.insert(0, %(root)r)
from ucd_amd import hip
dev = torch.device("cuda:0")
g = torch.Generator(dev).manual_seed(11)
def ints(*shape, hi):
    return torch.randint(-hi, hi + 1, shape, device=dev, generator=g).bfloat16()
# (M, K, N, conv3 = (H, W, d) or None): ragged row counts, both tile widths, 1x1 and 3x3, grids below and above 256 tiles
for M, K, N, sp in ((2 * 9 * 11, 128, 64, (9, 11, 1)), (777, 256, 128, None), (2178, 128, 256, (33, 33, 2)), (26136, 256, 256, (33, 33, 1)),
                    (26136, 1024, 256, None), (4 * 33 * 33 - 0, 64, 128, (33, 33, 6))):
    a = ints(M, K, hi=2) * (torch.rand(M, K, device=dev, generator=g) < 0.2)
    taps = 9 if sp else 1
    w = ints(N, taps * K, hi=1) * (torch.rand(N, taps * K, device=dev, generator=g) < 0.2)
    y = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
    if sp:
        H, W, d = sp
        B = M // (H * W)
        ref = torch.nn.functional.conv2d(a.float().view(B, H, W, K).permute(0, 3, 1, 2),
                                         w.float().view(N, 3, 3, K).permute(0, 3, 1, 2), None, 1, d, d).permute(0, 2, 3, 1).reshape(M, N)
    else:
        ref = a.float() @ w.float().t()
    assert ref.abs().max().item() <= 256
    hip.conv1x1(a, w, y, conv3=sp)
    assert torch.equal(y.float(), ref), ("plain", M, K, N, sp)
    part = hip.conv1x1_stats_partial(M, N, dev)
    y.fill_(float("nan"))
    hip.conv1x1(a, w, y, out_mode=2, partial=part, conv3=sp)
    assert torch.equal(y.float(), ref), ("stats", M, K, N, sp)
    tiles = hip.load().ucd_conv1x1_row_tiles(M)
    p = part.view(tiles, 3, N)
    for t in (0, tiles - 1):
        rows = ref[t * 128:(t + 1) * 128]
        k = p[t, 0]
        assert torch.equal(p[t, 1], (rows - k).sum(0)) and torch.equal(p[t, 2], ((rows - k) ** 2).sum(0)), ("partials", M, K, N, sp, t)
    res = ints(M, N, hi=3)
    one, zero = torch.ones(N, device=dev), torch.zeros(N, device=dev)
    hip.conv1x1(a, w, y, out_mode=1, out_norm=(zero, one, zero, None, hip.ACT_LEAKY_RELU, 0.5), residual=res, conv3=sp)
    z = ref + res.float()
    assert torch.equal(y.float(), torch.where(z > 0, z, z * 0.5).bfloat16().float()), ("affine", M, K, N, sp)
print("ok")
"""


@pytest.mark.parametrize("M,K,N,sp", [(26136, 1024, 256, None), (26136, 256, 1024, None), (26136, 256, 256, (33, 33, 1)),
                                      (101400, 128, 512, None), (3267, 512, 128, None), (300, 64, 64, None),
                                      # round 6: the 3-image shapes - the atomic path runs them on 64-row loader-wave tiles (conv_lw_kernel,
                                      # BM = 64), the deterministic path on the 128-row form: same stored products bit for bit
                                      (3267, 1024, 256, None), (3267, 256, 256, (33, 33, 1)), (3267, 2048, 256, (33, 33, 12)),
                                      (1089, 512, 512, (33, 33, 2))])
def test_atomic_statistics_epilogues_and_the_finalising_apply_passes(M, K, N, sp):
    """Round 5: the statistics epilogue (out_mode 2) and the link epilogues (out_mode 3 / 4) add their column sums with fp32 atomics
    into ONE zeroed [2 N] accumulator per layer (``stat_acc``) instead of per-tile rows, and the apply passes finalise it in their
    prologue (``ucd_abn_apply_stats`` / ``ucd_abn_bwd_apply_raw``: no tile_stats_reduce / reduce_bands launch; SURVEY K1,
    modules/residual.py:51-73).  Held against (a) float64 statistics of the stored map, (b) the deterministic per-tile path of the
    same library: same stored products bit for bit, mean / invstd / running statistics / sums to fp32 rounding (1e-6 relative - the
    atomic sums are order dependent in the last bits), dx of the backward apply equal when fed the same sums."""
    from ucd_amd import hip
    g = torch.Generator(DEV).manual_seed(M + 3 * K + N)
    taps = 9 if sp else 1
    a = (torch.randn(M, K, device=DEV, generator=g) * 1.3 + 0.2).bfloat16()
    w = (torch.randn(N, taps * K, device=DEV, generator=g) * (2.0 / (K * taps)) ** 0.5).bfloat16()
    gamma = (torch.rand(N, device=DEV, generator=g) - 0.3)                      # negative entries: the |gamma| + eps convention
    beta = torch.randn(N, device=DEV, generator=g) * 0.1
    shift = torch.randn(N, device=DEV, generator=g) * 0.3                       # the common shift (the layer's running mean)
    res = torch.randn(M, N, device=DEV, generator=g).bfloat16()
    # ---- deterministic path
    z0 = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    part = hip.conv1x1_stats_partial(M, N, DEV)
    hip.conv1x1(a, w, z0, out_mode=2, partial=part, conv3=sp)
    buf0 = torch.zeros(6 * N, device=DEV)
    rm0, rv0 = shift.clone(), torch.ones(N, device=DEV)
    hip.conv1x1_stats_finalize(part, M, N, gamma, rm0, rv0, 0.1, 1e-5, buf0, None, hip.NORM_ABS_GAMMA)
    y0 = torch.empty_like(z0)
    hip.abn_apply(z0, N, y0, N, res, N, M, N, None, 1, buf0[3 * N:4 * N], buf0[5 * N:], beta, hip.ACT_LEAKY_RELU | hip.NORM_ABS_GAMMA, 0.01)
    # ---- atomic path
    z1 = torch.empty_like(z0)
    R = hip.load().ucd_conv1x1_stat_replicas(M)               # row tile t adds into replica t % R: <= 64 adds per address
    assert R >= 1 and (R & (R - 1)) == 0 and (hip.conv1x1_row_tiles(M) <= 64 * R or R == 64)
    acc = torch.zeros(R, 2 * N, device=DEV)
    buf1 = torch.zeros(6 * N, device=DEV)
    rm1, rv1 = shift.clone(), torch.ones(N, device=DEV)
    hip.conv1x1(a, w, z1, out_mode=2, partial=buf1[2 * N:3 * N], conv3=sp, stat_acc=acc, stat_shift=rm1, stat_rep=R)
    assert torch.equal(z0, z1)
    assert torch.equal(buf1[2 * N:3 * N], shift)                                # the snapshot of the shift
    zf = z1.double()
    assert R == 1 or acc[R - 1].abs().sum() > 0                                 # every replica took adds
    torch.testing.assert_close(acc.sum(0)[:N].double(), (zf - shift.double()).sum(0), rtol=2e-5, atol=2e-2)
    y1 = torch.empty_like(z0)
    hip.abn_apply_stats(z1, y1, res, M, N, acc, buf1[2 * N:3 * N], float(M), gamma, beta, rm1, rv1, 0.1, 1e-5, buf1,
                        hip.ACT_LEAKY_RELU | hip.NORM_ABS_GAMMA, 0.01, reps=R)
    mean, var = zf.mean(0), zf.var(0, unbiased=False)
    torch.testing.assert_close(buf1[3 * N:4 * N].double(), mean, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(buf1[4 * N:5 * N].double(), 1 / torch.sqrt(var + 1e-5), rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(buf1[3 * N:], buf0[3 * N:], rtol=1e-5, atol=2e-6)            # mean | invstd | scale: both paths
    torch.testing.assert_close(rm1, rm0, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rv1, rv0, rtol=1e-5, atol=1e-6)
    assert _rel(y1, y0) < 5e-4 and (y1.float() - y0.float()).abs().max() <= 2.0 ** -7 * y0.float().abs().max()
    # ---- link epilogue (out_mode 3) into an accumulator, the backward apply on the raw sums
    om, osc, osh, oinv = (torch.randn(N, device=DEV, generator=g) * 0.3, torch.rand(N, device=DEV, generator=g) + 0.5,
                          torch.randn(N, device=DEV, generator=g) * 0.2, torch.rand(N, device=DEV, generator=g) + 0.5)
    tiles = hip.conv1x1_row_tiles(M)
    norm = (om, osc, osh, oinv, hip.ACT_LEAKY_RELU, 0.01)
    p3 = torch.empty(tiles, 2, N, device=DEV)
    dz0 = torch.empty_like(z0)
    hip.conv1x1(a, w, dz0, out_mode=3, out_norm=norm, residual=res, partial=p3, conv3=sp)
    sums0 = torch.zeros(2 * N, device=DEV)
    hip._check(hip.load().ucd_abn_reduce_partials(hip.ptr(p3), tiles, N, hip.ptr(sums0), None, hip.ptr(gamma), hip.NORM_ABS_GAMMA,
                                                  hip.stream()), "reduce")
    acc3, acc3b = torch.zeros(R, 2 * N, device=DEV), torch.zeros(R, 2 * N, device=DEV)
    dz1 = torch.empty_like(z0)
    hip.conv1x1(a, w, dz1, out_mode=3, out_norm=norm, residual=res, partial=p3, conv3=sp, stat_acc=acc3, stat_acc2=acc3b, stat_rep=R)
    assert torch.equal(dz0, dz1)
    sign = torch.where(gamma < 0, -1.0, 1.0)
    torch.testing.assert_close(acc3.sum(0)[:N], sums0[:N], rtol=2e-5, atol=2e-3)
    torch.testing.assert_close(acc3.sum(0)[N:] * sign, sums0[N:], rtol=2e-5, atol=2e-3)
    torch.testing.assert_close(acc3b, acc3, rtol=1e-4, atol=1e-3)               # the second accumulator took the same adds (in its own order)
    # the producer's backward: identity activation on d pre, mean / invstd / scale of ITS statistics (buf0), parameter gradients
    dx0, dx1 = torch.empty_like(z0), torch.empty_like(z0)
    act = hip.ACT_IDENTITY | hip.NORM_ABS_GAMMA
    hip.abn_bwd_apply(z0, N, dz0, N, None, 0, dx0, N, None, 0, M, N, None, 1, buf0[3 * N:4 * N], buf0[4 * N:5 * N], buf0[5 * N:], beta,
                      gamma, sums0, float(M), 0, act, 0.0)
    # fed the SAME sums (sign removed again): bit-identical dx, and the parameter gradients written by the launch
    raw = torch.cat([sums0[:N], sums0[N:] * sign])
    gout = torch.full((2 * N,), float("nan"), device=DEV)
    hip.abn_bwd_apply_raw(z0, dz0, None, dx1, None, M, N, buf0[3 * N:4 * N], buf0[4 * N:5 * N], buf0[5 * N:], beta, gamma, raw, None, gout,
                          float(M), act, 0.0)
    assert torch.equal(dx0, dx1)
    assert torch.equal(gout, sums0)
    # and from the replicated accumulators (summed in the prologue): the same dx up to the rounding of the sums
    dx2 = torch.empty_like(z0)
    hip.abn_bwd_apply_raw(z0, dz0, None, dx2, None, M, N, buf0[3 * N:4 * N], buf0[4 * N:5 * N], buf0[5 * N:], beta, gamma, acc3, acc3b, gout,
                          float(M), act, 0.0, reps=R)
    assert _rel(dx2, dx0) < 5e-4
    torch.testing.assert_close(gout, sums0, rtol=1e-4, atol=5e-3)
    if sp is None:
        # ---- block link (out_mode 4)
        z3 = torch.randn(M, N, device=DEV, generator=g).bfloat16()
        p4 = torch.empty(tiles, 2, N, device=DEV)
        y4a, y4b = res.clone(), res.clone()
        kw = dict(out_mode=4, out_norm=(om, None, None, oinv, hip.ACT_LEAKY_RELU, 0.01), residual=z3, side2=res, accumulate=True)
        hip.conv1x1(a, w, y4a, partial=p4, **kw)
        s4 = torch.zeros(2 * N, device=DEV)
        hip._check(hip.load().ucd_abn_reduce_partials(hip.ptr(p4), tiles, N, hip.ptr(s4), None, None, 0, hip.stream()), "reduce")
        acc4 = torch.zeros(R, 2 * N, device=DEV)
        hip.conv1x1(a, w, y4b, partial=p4, stat_acc=acc4, stat_rep=R, **kw)
        assert torch.equal(y4a, y4b)
        torch.testing.assert_close(acc4.sum(0), s4, rtol=2e-5, atol=5e-3)


def test_atomic_links_survive_an_arena_reset_and_refuse_a_second_fill():
    """The arena slots behind the atomic links (csrc/abn_node.cpp) carry the arena's generation and a fill state: (1) a statistics
    arena reset between a graph's forward and its backward (another training forward of the model - here an explicit reset)
    invalidates the slots, the consumers do not serve the links and the producers run their own reduction passes - same gradients
    as an undisturbed run up to the order of the sums; (2) a retain_graph replay of the whole backward finds the slots already read
    and does not fill them a second time (a consumer replayed ALONE against a filled, unread slot raises: link_servable)."""
    from functools import partial
    from ucd_amd import abn, blocks, hip
    from ucd_amd.ddp import DistributedDataParallel
    node = blocks._gemm_node()
    if node is None or not hasattr(node, "stat_arena_reset"):
        pytest.skip("C++ node not built")
    norm = partial(abn.InPlaceABNSync, activation="leaky_relu", activation_param=0.01)
    x0 = synth.t_normal(9, (24, 1024, 33, 33), stream=1).to(DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    dy = synth.t_normal(10, (24, 1024, 33, 33), stream=1).to(DEV).bfloat16().contiguous(memory_format=torch.channels_last)

    def make():
        net = torch.nn.Sequential(blocks.ResidualBlock(1024, (256, 256, 1024), norm_act=norm), blocks.ResidualBlock(1024, (256, 256, 1024), norm_act=norm))
        net.load_state_dict(synth.fill_state_dict(net.state_dict(), 5))
        net = net.to(DEV).to(memory_format=torch.channels_last).train()
        return net, DistributedDataParallel(net, bf16_weights=True)

    grads = []
    for reset_between in (False, True):
        net, mod = make()
        x = x0.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = mod(x * 1.0)
        if reset_between:
            node.stat_arena_reset(torch.cuda.current_device(), hip.stream())
        y.backward(dy)
        mod.finish_grad_sync()
        torch.cuda.synchronize()
        grads.append([x.grad.float().clone()] + [p.grad.float().clone() for p in net.parameters()])
    for a, b in zip(*grads):
        # a link served or not: one more bf16 rounding of a gradient map (the link tests' bound: 3e-2; here, with fp32-atomic sums on
        # top, 2.2e-2 - 4.5e-2 from run to run on the d bias vectors, whose terms cancel); a lost or doubled link term is off by O(1)
        assert torch.isfinite(a).all() and _rel(a, b) < 8e-2
    # (2) the whole backward run twice (retain_graph): the slots are in state "read" after the first pass, so the second pass's
    # consumers do not serve the links (a second fill would double the sums) and the producers reduce for themselves - the same
    # gradients again, not twice the link terms
    net, mod = make()
    x = x0.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = mod(x * 1.0)
    y.backward(dy, retain_graph=True)
    mod.finish_grad_sync()
    torch.cuda.synchronize()
    first = [x.grad.float().clone()] + [p.grad.float().clone() for p in net.parameters()]
    x.grad = None
    mod.zero_grad()
    y.backward(dy)
    mod.finish_grad_sync()
    torch.cuda.synchronize()
    second = [x.grad.float().clone()] + [p.grad.float().clone() for p in net.parameters()]
    for a, b in zip(first, second):
        assert torch.isfinite(b).all() and _rel(b, a) < 8e-2


@pytest.mark.parametrize("pipe", ["2x64", "4x32", "4x64", "lw32", "lw64", "lw64x2", "lw256"])
def test_every_pipeline_form_of_the_gemm_kernel_is_exact_on_integers(pipe):
    """The GEMM / implicit-GEMM kernel has seven pipeline forms since round 4 (double buffer, two four-stage forms, loader waves on
    128-row tiles with 2 / 3 / 4 stages and on 256-row tiles); the library picks by grid, ``UCD_CONV_PIPE`` forces one.  Each form runs
    the plain, statistics and affine + residual + activation epilogues on sparse small integers (every output exactly representable:
    bit-exact against fp32 products, the per-tile statistics partials included) at ragged row counts, in a process of its own."""
    import subprocess
    import sys
    env = dict(os.environ, UCD_CONV_PIPE=pipe)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _PIPE_CHILD % {"root": root}], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (pipe, r.stdout[-500:], r.stderr[-1500:])

_BN64_VS_BN128 = r"""
import sys, torch
from ucd_amd import hip
dev = torch.device("cuda:0")
torch.manual_seed(0)
outs = {}
cases = [(2178, 256, 256, None), (2178, 1024, 256, None), (2178, 256, 256, (33, 33, 2)), (2178, 512, 512, None), (8450, 512, 128, None),
         (8450, 128, 128, (65, 65, 1)), (2178, 2048, 256, (33, 33, 12)), (2, 2048, 256, None)]
for (M, K, N, conv3) in cases:
    a = (torch.randn(M, K, device=dev) * 0.3 + 1.0).abs().bfloat16()          # post-activation-like: positive, mean >> spread
    KK = 9 * K if conv3 else K
    w = (torch.randn(N, KK, device=dev) * (1.0 / KK) ** 0.5 + 0.5 / KK).bfloat16()
    mean = torch.randn(N, device=dev) * 0.1; scale = torch.rand(N, device=dev) + 0.5; shift = torch.randn(N, device=dev) * 0.1
    invstd = torch.rand(N, device=dev) + 0.5
    res = torch.randn(M, N, device=dev).bfloat16(); z3 = torch.randn(M, N, device=dev).bfloat16()
    tiles = hip.conv1x1_row_tiles(M)
    kw = dict(conv3=conv3) if conv3 else {}
    r = {}
    wide = torch.zeros(M, 640, device=dev, dtype=torch.bfloat16)              # the product lands in a channel slice of a wider buffer
    hip.conv1x1(a, w, wide[:, 128:128 + N], **kw); r["plain_slice"] = wide.clone()
    y0 = res.clone(); hip.conv1x1(a, w, y0, accumulate=True, **kw); r["accumulate"] = y0
    y1 = torch.empty_like(res); hip.conv1x1(a, w, y1, out_mode=1, out_norm=(mean, scale, shift, None, 1, 0.01), residual=res, **kw); r["affine_res"] = y1
    p2 = hip.conv1x1_stats_partial(M, N, dev); y2 = torch.empty_like(res)
    hip.conv1x1(a, w, y2, out_mode=2, partial=p2, **kw); r["stats_y"] = y2; r["stats_partial"] = p2.clone()
    buf = torch.zeros(6 * N, device=dev)
    hip.conv1x1_stats_finalize(p2, M, N, torch.ones(N, device=dev), torch.zeros(N, device=dev), torch.ones(N, device=dev), 0.1, 1e-5, buf)
    r["mean_invstd"] = buf[3 * N:5 * N].clone()
    r["true_mean_invstd"] = torch.cat([y2.float().mean(0), 1.0 / torch.sqrt(y2.float().var(0, unbiased=False) + 1e-5)])
    p3 = torch.zeros(tiles, 2, N, device=dev); y3 = torch.empty_like(res)
    hip.conv1x1(a, w, y3, out_mode=3, out_norm=(mean, scale, shift, invstd, 1, 0.01), residual=res, partial=p3, **kw); r["link_y"] = y3; r["link_partial"] = p3.clone()
    if not conv3:
        p4 = torch.zeros(tiles, 2, N, device=dev); y4 = res.clone()
        hip.conv1x1(a, w, y4, out_mode=4, out_norm=(mean, None, None, invstd, 1, 0.01), residual=z3, side2=res, partial=p4, accumulate=True)
        r["block_y"] = y4; r["block_partial"] = p4.clone()
    torch.cuda.synchronize()
    outs[(M, K, N, conv3)] = {k: v.float().cpu() for k, v in r.items()}
torch.save(outs, sys.argv[1])
"""


def test_small_grids_on_64_column_tiles_equal_the_128_column_tiles(tmp_path):
    """Round 4: launches of at most 128 (128 x 128) tiles - the 3 - 6 images per GPU of the multi-GPU split - run on 128 x 64 tiles
    (``UCD_CONV_BN64_TILES``, csrc/conv1x1.hip).  Every output mode (plain into a channel slice, accumulate, affine + residual,
    statistics, backward link, block link; 1x1, 3x3, a dilated ASPP branch, a two-row product) must give the SAME bf16 outputs bit for
    bit as the 128-column tiles; the per-tile partial sums are taken in a different order (agreement to fp32 rounding), and the
    finalised mean / invstd of both agree with torch's to 1e-5.  Two child processes: the switch is read once per process."""
    import subprocess
    import sys
    script = tmp_path / "bn64_vs_bn128.py"
    script.write_text(_BN64_VS_BN128)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for mode, v in (("bn64", "128"), ("bn128", "0")):
        env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""), UCD_CONV_BN64_TILES=v)
        env.pop("UCD_CONV_PIPE", None)
        out = tmp_path / f"{mode}.pt"
        subprocess.run([sys.executable, str(script), str(out)], env=env, check=True, timeout=600)
        res[mode] = torch.load(out)
    for case, a in res["bn64"].items():
        b = res["bn128"][case]
        for name in a:
            if name.endswith("partial"):
                rel = ((a[name] - b[name]).norm() / (b[name].norm() + 1e-30)).item()
                assert rel < 1e-6, (case, name, rel)
            elif name == "mean_invstd":
                N = a[name].numel() // 2
                t = a["true_mean_invstd"]
                for got in (a[name], b[name]):
                    assert ((got[N:] - t[N:]).abs() / t[N:]).max().item() < 1e-5, (case, "invstd")
                    assert ((got[:N] - t[:N]).abs() * t[N:]).max().item() < 1e-5, (case, "mean")
            elif name != "true_mean_invstd":
                assert torch.equal(a[name], b[name]), (case, name, int((a[name] != b[name]).sum()))

@pytest.mark.parametrize("B,C,h,w,Ct", [(2, 256, 33, 33, 21), (3, 256, 17, 19, 16), (2, 256, 9, 9, 64), (1, 128, 5, 7, 6)])
def test_classifier_heads_on_the_own_kernels_match_the_convolution(B, C, h, w, Ct):
    """The classifier heads (segmentation_module.py:72-74, 102-105: Ct = 16 + 5 ... classes, off the kernels' 64-channel grid) as
    one product on a zero-padded 64-row weight (``_HeadProduct``): logits, input gradient, weight gradient (fp32) and bias gradient
    against the fp32 convolution on the same bf16 inputs, at bf16 resolution."""
    from ucd_amd.segmentation_module import _HeadProduct, _own_heads_ok
    dev = torch.device("cuda:0")
    x = (synth.t_normal(70 + Ct, (B, C, h, w), stream=1) * 0.8 + 0.2).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
    wt = (synth.t_normal(71 + Ct, (Ct, C, 1, 1), stream=2) * C ** -0.5).to(dev)
    bias = (synth.t_normal(72 + Ct, (Ct,), stream=3) * 0.3).to(dev)
    g = synth.t_normal(73 + Ct, (B, Ct, h, w), stream=4).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
    assert _own_heads_ok(x, wt)
    xa = x.clone().requires_grad_(True); wa = wt.clone().requires_grad_(True); ba = bias.clone().requires_grad_(True)
    y = _HeadProduct.apply(xa, wa, ba)
    assert y.shape == (B, Ct, h, w) and y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
    y.backward(g)
    xr = x.float().requires_grad_(True); wr = wt.bfloat16().float().requires_grad_(True); br = bias.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br)
    yr.backward(g.float())
    rel = lambda a, b: ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()
    assert rel(y, yr) < 4e-3, rel(y, yr)                       # one bf16 rounding of the output
    assert rel(xa.grad, xr.grad) < 4e-3, rel(xa.grad, xr.grad)
    assert wa.grad.dtype == torch.float32 and rel(wa.grad, wr.grad) < 2e-3, rel(wa.grad, wr.grad)
    assert rel(ba.grad, br.grad) < 1e-4, rel(ba.grad, br.grad)
    # a frozen head: no weight / bias gradient asked for
    xa2 = x.clone().requires_grad_(True)
    _HeadProduct.apply(xa2, wt, bias).backward(g)
    assert torch.equal(xa2.grad, xa.grad)


# ---- round 6: what the bf16 whole-step tests cannot see, at the benchmark's batch, against float64 ---------------
# The whole-step bf16 tests hold the body's gradients to a factor of two and the 20-step updates to cosine 0.75 (chaotic random-weight
# network); the identity-block and ASPP chain tests above pin the kernels of ONE identity block per stage and the head.  Below: the
# PROJECTION block of every stage (stride 2 / dilation change, the shortcut projection, conv1's alias for the second consumer of the
# block input, the strided kernels and their library input gradients), the stem (7x7 / 2 convolution from the fp32 image, norm + max
# pool in one pass both ways) and the classifier heads (zero-padded 64-row product) - each at B = 24 with the same slope-1 bounds.
@pytest.mark.parametrize("cin,chans,dil,hw,stride", [(64, (64, 64, 256), 1, 129, 1), (256, (128, 128, 512), 1, 129, 2),
                                                     (512, (256, 256, 1024), 1, 65, 2), (1024, (512, 512, 2048), 2, 33, 1)])
def test_bench_shape_projection_blocks_against_float64(cin, chans, dil, hw, stride):
    """mod2 / mod3 / mod4 / mod5 ``block1`` (modules/residual.py:51-97 with ``proj_conv`` / ``proj_bn``; models/resnet.py:97-101
    decides stride and dilation) at B = 24 as the fused chain the benchmark step runs, against oracle/model.py::residual_block in
    float64 on torch's native kernels.  slope = 1: the arithmetic of the chain alone (one bf16 rounding per stored map)."""
    from functools import partial
    from ucd_amd import abn, blocks
    from ucd_amd.ddp import DistributedDataParallel
    from oracle import model as OM
    slope = 1.0
    norm = partial(abn.InPlaceABNSync, activation="leaky_relu", activation_param=slope)
    B, ho = 24, (hw - 1) // stride + 1
    x0 = synth.t_normal(29, (B, cin, hw, hw), stream=1).to(DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    dy = synth.t_normal(30, (B, chans[2], ho, ho), stream=1).to(DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    blk = blocks.ResidualBlock(cin, chans, norm_act=norm, stride=stride, dilation=dil)
    assert hasattr(blk, "proj_conv")
    state = synth.fill_state_dict(blk.state_dict(), 7)
    blk.load_state_dict(state)
    blk = blk.to(DEV).to(memory_format=torch.channels_last).train()
    mod = DistributedDataParallel(blk, bf16_weights=True)
    x = x0.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = mod(x * 1.0)
    y.backward(dy)
    mod.finish_grad_sync()
    yf, gxf = y.detach().float(), x.grad.float()
    gf = {n: p.grad.float().clone() for n, p in blk.named_parameters()}
    rvf, rmf = blk.proj_bn.running_var.clone(), blk.convs.bn3.running_mean.clone()
    del blk, mod, x, y
    torch.cuda.empty_cache()
    P = {"blk." + k: (v.to(DEV).double() if v.is_floating_point() else v.to(DEV)) for k, v in state.items()}
    for k, v in P.items():
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    with torch.backends.cudnn.flags(enabled=False):
        x = x0.double().contiguous().clone().requires_grad_(True)
        y = OM.residual_block(x * 1.0, P, "blk", stride, dil, True, slope=slope)
        y.backward(dy.double().contiguous())
    y64, gx64 = y.detach().float(), x.grad.float()
    g64 = {k[4:]: v.grad.float() for k, v in P.items() if v.requires_grad}
    rv64, rm64 = P["blk.proj_bn.running_var"].float(), P["blk.convs.bn3.running_mean"].float()
    del P, x, y
    torch.cuda.empty_cache()
    assert set(gf) == set(g64)
    # slope 1: a shift of bn2's output passes conv3 (1x1) and is removed by bn3's batch mean (zero gradient in exact arithmetic);
    # a shift of bn1's output meets conv2's zero padding: d bn1.bias is the border term only, held on the scale of the norm's weight
    border = ("convs.bn1.bias",)
    worst = _worst_param_grad({n: v for n, v in gf.items() if n not in border}, {n: v for n, v in g64.items() if n not in border},
                              ("convs.bn2.bias",))
    for n in border:
        err = (gf[n] - g64[n]).pow(2).mean().sqrt().item()
        assert err < 2e-2 * g64["convs.bn1.weight"].pow(2).mean().sqrt().item(), (n, err)
    l2 = _rel(gxf, gx64)
    print("bench-shape projection block", cin, chans, hw, "stride", stride, "dil", dil, "y", _rel(yf, y64), "dx", l2, "worst param grad", worst)
    assert _rel(yf, y64) < 1e-2
    torch.testing.assert_close(rvf, rv64, rtol=5e-3, atol=1e-5)
    torch.testing.assert_close(rmf, rm64, rtol=5e-3, atol=2e-3)
    assert l2 < 6e-3, l2
    assert worst < 1e-2, worst


def test_bench_shape_stem_against_float64(monkeypatch):
    """mod1 of the body at the benchmark's batch (models/resnet.py:58-64: 7x7 / 2 convolution of the fp32 image, InPlaceABNSync,
    3x3 / 2 max pool; B = 24 at 513 x 513) on the own kernels (csrc/stem.hip: implicit GEMM from the fp32 image, norm + pool as one
    pass forward and backward; the weight gradient is the library's) against float64 torch operators, STAGE BY STAGE on the product's
    own intermediate maps: a max pool decides its argmax on the values it is given, and two candidates of a window within one bf16
    rounding of each other send the gradient to different pixels in a bf16 and a float64 chain (measured: 8 % relative L2 on the
    convolution's weight gradient end to end - every bf16 implementation's, not a kernel's).  So (a) the convolution output against
    float64, (b) norm + pool forward and backward against float64 FROM the stored bf16 convolution output, with the pool deciding
    on the bf16-rounded activation as the product (and any bf16 pipeline) does, (c) the weight gradient against float64 from the
    product's own d z."""
    from functools import partial
    from ucd_amd import abn, backbone
    from ucd_amd.ddp import DistributedDataParallel
    from oracle import model as OM
    norm = partial(abn.InPlaceABNSync, activation="leaky_relu", activation_param=1.0)

    class Stem(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.mod1 = backbone.ResNet([1, 1, 1, 1], True, norm_act=norm).mod1
        forward = backbone.ResNet._stem

    seen = {}
    inner = backbone._stem_conv

    def spy(conv, x):
        z = inner(conv, x)
        z.retain_grad()
        seen["z"] = z
        return z
    monkeypatch.setattr(backbone, "_stem_conv", spy)
    B, H = 24, 513
    img = synth.t_normal(41, (B, 3, H, H), stream=1).to(DEV)
    stem = Stem()
    state = synth.fill_state_dict(stem.state_dict(), 11)
    stem.load_state_dict(state)
    stem = stem.to(DEV).to(memory_format=torch.channels_last).train()
    mod = DistributedDataParallel(stem, bf16_weights=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = mod(img)
    assert y.shape == (B, 64, 129, 129) and y.dtype == torch.bfloat16 and "z" in seen
    dy = synth.t_normal(42, tuple(y.shape), stream=1).to(DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    y.backward(dy)
    mod.finish_grad_sync()
    yf, z, dz = y.detach().float(), seen["z"].detach(), seen["z"].grad.detach()
    gf = {n: p.grad.float().clone() for n, p in stem.named_parameters()}
    rmf, rvf = stem.mod1.bn1.running_mean.clone(), stem.mod1.bn1.running_var.clone()
    del stem, mod, y, seen
    torch.cuda.empty_cache()
    w64 = state["mod1.conv1.weight"].to(DEV).bfloat16().double()                # the working copy's values
    with torch.backends.cudnn.flags(enabled=False):
        z64 = F.conv2d(img.double(), w64, stride=2, padding=3)
        e_z = _rel(z.float(), z64.float())
        del z64
        P = {"s." + k: (v.to(DEV).double() if v.is_floating_point() else v.to(DEV)) for k, v in state.items() if "bn1" in k}
        for k, v in P.items():
            if not k.endswith(("running_mean", "running_var")):
                v.requires_grad_(True)
        zin = z.double().requires_grad_(True)
        a64 = OM.abn(zin, P, "s.mod1.bn1", True, slope=1.0)
        # the pool sees the activation as a bf16 map (what a separate apply pass stores, and what csrc/stem.hip reproduces bit for bit
        # in its one-pass form): neighbours that round to the same bf16 value TIE and the first of the window wins - pooled on the
        # float64 values the reference sends 0.01 % of the gradients to another pixel of their window (3 % of d z in relative L2,
        # tests/diag/stem_dz_diag.py).  Straight-through rounding: decisions on the rounded values, derivative of the identity.
        a64 = a64 + (a64.detach().float().bfloat16().double() - a64.detach())
        y64 = F.max_pool2d(a64, 3, stride=2, padding=1)
        y64.backward(dy.double())
        e_y, e_dz = _rel(yf, y64.float()), _rel(dz.float(), zin.grad.float())
        e_bn = {n: _rel(gf["mod1.bn1." + n], P["s.mod1.bn1." + n].grad.float()) for n in ("weight", "bias")}
        rm64, rv64 = P["s.mod1.bn1.running_mean"].float(), P["s.mod1.bn1.running_var"].float()
        del y64, zin
        torch.cuda.empty_cache()
        dw64 = torch.nn.grad.conv2d_weight(img.double(), w64.shape, dz.double(), stride=2, padding=3)
        e_dw = _rel(gf["mod1.conv1.weight"], dw64.float())
    print("bench-shape stem: conv", e_z, "norm + pool y", e_y, "dz", e_dz, "bn grads", e_bn, "conv weight gradient", e_dw)
    assert e_z < 6e-3, e_z                      # one bf16 rounding of the image, one of the output
    assert e_y < 4e-3 and e_dz < 6e-3, (e_y, e_dz)
    assert max(e_bn.values()) < 1e-2, e_bn
    torch.testing.assert_close(rmf, rm64, rtol=5e-3, atol=2e-3)
    torch.testing.assert_close(rvf, rv64, rtol=5e-3, atol=1e-5)
    assert e_dw < 1e-2, e_dw


def test_bench_shape_classifier_heads_against_float64():
    """The classifier heads at the benchmark's shape (segmentation_module.py:104-111: 1x1 convolutions with bias on the 256-channel
    head output; VOC 15-5 step 1: 16 + 5 classes, B = 24 at 33 x 33) as ONE zero-padded 64-row product on the own kernels
    (ucd_amd.segmentation_module._HeadProduct) against float64 ``F.conv2d``: logits, input gradient, weight and bias gradients."""
    from ucd_amd.segmentation_module import _HeadProduct, _own_heads_ok
    B, C, hw, Ct = 24, 256, 33, 21
    x0 = synth.t_normal(51, (B, C, hw, hw), stream=1).to(DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    w0 = (synth.t_normal(52, (Ct, C, 1, 1), stream=1) * (1.0 / C) ** 0.5).to(DEV)
    b0 = (synth.t_normal(53, (Ct,), stream=1) * 0.1).to(DEV)
    g0 = synth.t_normal(54, (B, Ct, hw, hw), stream=1).to(DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    x = x0.clone().requires_grad_(True)
    w, b = w0.bfloat16().requires_grad_(True), b0.clone().requires_grad_(True)
    assert _own_heads_ok(x, w)
    y = _HeadProduct.apply(x, w, b)
    y.backward(g0)
    x64 = x0.double().requires_grad_(True)
    w64, b64 = w0.bfloat16().double().requires_grad_(True), b0.double().requires_grad_(True)
    y64 = F.conv2d(x64, w64, b64)
    y64.backward(g0.double())
    errs = {"y": _rel(y.float(), y64.float()), "dx": _rel(x.grad.float(), x64.grad.float()), "dw": _rel(w.grad.float(), w64.grad.float()),
            "db": _rel(b.grad.float(), b64.grad.float())}
    print("bench-shape classifier heads", {k: round(v, 6) for k, v in errs.items()})
    assert errs["y"] < 4e-3 and errs["dx"] < 4e-3 and errs["dw"] < 4e-3 and errs["db"] < 1e-4, errs


def test_deferred_slab_sums_ride_in_the_next_weight_gradient_launch():
    """C ABI of round 6's deferral (include/ucd_hip.h: ucd_conv_wgrad_ex flags bit 0, ucd_conv_wgrad_defer / _flush / _drop): under
    deferral a call leaves its gradient unwritten until the next weight-gradient call on the stream (which carries the sum as extra
    workgroups, every kernel form: 1x1, 9-tap, three-tap) or the flush; results bit-identical to the immediate sums; without
    ucd_conv_wgrad_defer(1), or without the flag, nothing is deferred; a dropped pending sum is never written."""
    from ucd_amd import hip
    g = torch.Generator(DEV).manual_seed(77)
    cases = [(26136, 1024, 256, None), (26136, 256, 256, (33, 33, 1)), (3267, 256, 256, (33, 33, 1)), (3267, 256, 1024, None),
             (12675, 128, 128, (65, 65, 1))]
    ops = []
    for M, N, K, sp in cases:
        dz = torch.randn(M, N, device=DEV, generator=g).bfloat16()
        x = torch.randn(M, K, device=DEV, generator=g).bfloat16()
        ref = torch.empty(N, (9 if sp else 1) * K, device=DEV, dtype=torch.bfloat16)
        hip.conv_wgrad(dz, x, dw=ref, conv3=sp)
        ops.append((dz, x, sp, ref))
    torch.cuda.synchronize()
    assert hip.wgrad_defer(True) == 0
    try:
        outs = [torch.full_like(ref, float("nan")) for _, _, _, ref in ops]
        for i, (dz, x, sp, ref) in enumerate(ops):
            hip.conv_wgrad(dz, x, dw=outs[i], conv3=sp, defer=True)
            torch.cuda.synchronize()
            assert bool(torch.isnan(outs[i]).all()), i                    # this call's sum is pending ...
            if i:
                assert torch.equal(outs[i - 1], ops[i - 1][3]), i         # ... the previous one's rode in this launch
        hip.wgrad_flush()
        torch.cuda.synchronize()
        assert torch.equal(outs[-1], ops[-1][3])
        hip.wgrad_flush()                                                 # nothing pending: a no-op
        # without the flag: summed at once, and a pending sum of an earlier call still rides along
        a, b = torch.full_like(ops[0][3], float("nan")), torch.full_like(ops[1][3], float("nan"))
        hip.conv_wgrad(ops[0][0], ops[0][1], dw=a, conv3=ops[0][2], defer=True)
        hip.conv_wgrad(ops[1][0], ops[1][1], dw=b, conv3=ops[1][2])
        torch.cuda.synchronize()
        assert torch.equal(a, ops[0][3]) and torch.equal(b, ops[1][3])
        # a dropped pending sum is never written
        c = torch.full_like(ops[0][3], float("nan"))
        hip.conv_wgrad(ops[0][0], ops[0][1], dw=c, conv3=ops[0][2], defer=True)
        hip.wgrad_drop()
        hip.wgrad_flush()
        hip.conv_wgrad(ops[1][0], ops[1][1], dw=b, conv3=ops[1][2])
        torch.cuda.synchronize()
        assert bool(torch.isnan(c).all())
    finally:
        hip.wgrad_drop()
        assert hip.wgrad_defer(False) == 1
    # deferral off: the flag alone defers nothing
    d = torch.full_like(ops[0][3], float("nan"))
    hip.conv_wgrad(ops[0][0], ops[0][1], dw=d, conv3=ops[0][2], defer=True)
    torch.cuda.synchronize()
    assert torch.equal(d, ops[0][3])


def test_weight_gradients_on_the_side_stream_of_the_library():
    """C ABI of round 6's side stream (include/ucd_hip.h: ucd_conv_wgrad_ex flags bit 1 under ucd_conv_wgrad_defer(mode & 2)): the calls
    record their fork point, are launched one call later on a stream of the library, and joined by ucd_conv_wgrad_flush - the gradients are
    the bits of the plain calls; without the mode bit the flag changes nothing; a drop never launches what is still queued."""
    from ucd_amd import hip
    g = torch.Generator(DEV).manual_seed(11)
    shapes = [(3267, 256, 1024, None), (3267, 256, 256, (33, 33, 2)), (3267, 1024, 256, None), (3267, 128, 128, (33, 33, 1)),
              (3267, 512, 128, None)] * 3                                       # 15 calls
    ops, want = [], []
    for M, K, N, c3 in shapes:
        dz = (torch.randn(M, N, device=DEV, generator=g) * 0.1).bfloat16()
        x = torch.randn(M, K, device=DEV, generator=g).bfloat16()
        ops.append((dz, x, c3))
        want.append(hip.conv_wgrad(dz, x, dw=torch.empty(N, (9 if c3 else 1) * K, device=DEV, dtype=torch.bfloat16), conv3=c3).clone())
    torch.cuda.synchronize()
    lib = hip.load()

    def run():
        out = [torch.full_like(w, float("nan")) for w in want]
        for (dz, x, c3), d in zip(ops, out):
            hip.conv_wgrad(dz, x, dw=d, conv3=c3, defer=True, side=True)
        return out

    assert lib.ucd_conv_wgrad_mode() == 0
    out = run()                                                                  # mode 0: plain calls, final at once
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(out, want))
    assert hip.wgrad_defer(3) == 0
    try:
        assert lib.ucd_conv_wgrad_mode() == 3
        out = run()
        hip.wgrad_flush()                                                        # launches the rest of the queue, sums, joins
        hip.transpose_bf16(out[-1][:64, :64].contiguous(), torch.empty(64, 64, device=DEV, dtype=torch.bfloat16))   # a reader on this stream
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(out, want))
        # side stream without deferred sums
        hip.wgrad_defer(2)
        out = run()
        hip.wgrad_flush()
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(out, want))
        # a drop: what still waits in the queue is never launched, a pending sum never written - every gradient is either final
        # (its group went out and its sum rode in a later launch) or untouched, and the last call's is untouched
        hip.wgrad_defer(3)
        out = run()
        hip.wgrad_drop()
        hip.wgrad_flush()
        torch.cuda.synchronize()
        assert all(bool(torch.isnan(a.float()).all()) or torch.equal(a, b) for a, b in zip(out, want))
        assert bool(torch.isnan(out[-1].float()).all())
    finally:
        hip.wgrad_drop()
        assert hip.wgrad_defer(0) == 3


def test_deferred_weight_gradients_are_final_when_any_backward_pass_ends():
    """A weight gradient that was deferred / moved to the side stream is final only behind ucd_conv_wgrad_flush.  The gradient-bucket
    wrapper flushes in front of its copies; a pass it does not see to its end (``torch.autograd.grad``: no AccumulateGrad, no hook)
    is covered by the C++ node itself - its first such call queues the flush as an engine callback (csrc/abn_node.cpp:
    queue_end_of_pass_flush).  Without it the gradient below would be the uninitialised output buffer of a queued launch."""
    from ucd_amd import abn as _abn, hip
    node = _abn._abn_node()
    if node is None or not hasattr(node, "conv_stride1"):
        pytest.skip("C++ autograd nodes not built")
    g = torch.Generator(DEV).manual_seed(5)
    x = torch.randn(3, 256, 33, 33, device=DEV, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    ws = [(torch.randn(256, 256, 3, 3, device=DEV, generator=g) * 0.02).bfloat16().contiguous(memory_format=torch.channels_last)
          .requires_grad_() for _ in range(3)]

    def grads():
        y = x
        for w in ws:
            y = node.conv_stride1(y, w, 2, None, True, False, hip.stream(), True)
        return torch.autograd.grad(y.float().square().mean(), ws)

    n0 = node.pass_flushes()
    want = [t.clone() for t in grads()]                       # mode 0: every call final at once, no callback
    torch.cuda.synchronize()
    assert node.pass_flushes() == n0
    for i, mode in enumerate((1, 3)):
        assert hip.wgrad_defer(mode) == 0
        try:
            got = grads()
            assert hip.load().ucd_conv_wgrad_mode() == mode   # nobody switched the mode: only the pass's own callback flushed
            assert node.pass_flushes() == n0 + i + 1          # ... once
            torch.cuda.synchronize()
            # the last layer's gradient (the FIRST call of the pass: its product rode through the whole machinery) bit for bit;
            # the others sit behind the library's input-gradient solver, which is not bit-reproducible run to run (1 ulp)
            assert torch.equal(got[2], want[2])
            for a, b in zip(got, want):
                torch.testing.assert_close(a.float(), b.float(), rtol=2e-2, atol=2e-5)
        finally:
            hip.wgrad_drop()
            hip.wgrad_defer(0)
