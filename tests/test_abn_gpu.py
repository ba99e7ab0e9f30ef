"""GPU: the HIP ABN kernels (through the C ABI) against the oracle's restatement of inplace_abn (oracle/abn.py: CPU fp32
batch norm + activation; raw gamma for ABN, |gamma| + eps for the in-place variants - third-party wheel, parity unpinned
by the reference).  Tolerance 1e-5 abs/rel in fp32; bf16 I/O is checked against the same fp32
oracle at bf16 resolution."""
import numpy as np
import os

import pytest
import torch
import torch.nn.functional as F

from ucd_amd import synth

pytestmark = pytest.mark.gpu


def _ref(x, w, b, rm, rv, training, act, slope, residual=None, plane_bias=None, eps=1e-5, momentum=0.1, abs_gamma=False):
    from oracle.abn import abn_forward
    return abn_forward(x, w, b, rm, rv, training, momentum, eps, act, slope, abs_gamma, residual, plane_bias)


def _mk(seed, shape, dev=None, dtype=torch.float32):
    t = synth.t_normal(seed, shape, stream=1)
    return t


CASES = [
    # B, C, H, W
    (3, 64, 17, 19),
    (2, 256, 9, 9),
    (4, 16, 5, 7),
    (24, 256, 1, 1),
    (2, 2048, 5, 5),
    (2, 48, 6, 4),
]


@pytest.mark.parametrize("B,C,H,W", CASES)
@pytest.mark.parametrize("act", ["leaky_relu", "identity"])
@pytest.mark.parametrize("training", [True, False])
def test_abn_forward_backward_fp32(B, C, H, W, act, training):
    from ucd_amd.abn import ABN
    dev = torch.device("cuda:0")
    x = (synth.t_normal(B * C + H, (B, C, H, W), stream=1) * 1.7 + 0.3)
    m = ABN(C, activation=act)
    st = synth.fill_state_dict(m.state_dict(), seed=C + H)
    m.load_state_dict(st)
    m.train(training)
    w, b = st["weight"].clone().requires_grad_(True), st["bias"].clone().requires_grad_(True)
    rm, rv = st["running_mean"].clone(), st["running_var"].clone()
    xr = x.clone().requires_grad_(True)
    yr = _ref(xr, w, b, rm, rv, training, act, 0.01)
    g = synth.t_normal(11, (B, C, H, W), stream=2)
    yr.backward(g)

    m = m.to(dev)
    xg = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    yg = m(xg)
    yg.backward(g.to(dev))
    torch.cuda.synchronize()
    torch.testing.assert_close(yg.detach().cpu(), yr.detach(), rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(xg.grad.cpu(), xr.grad, rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(m.weight.grad.cpu(), w.grad, rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(m.bias.grad.cpu(), b.grad, rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(m.running_mean.cpu(), rm, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(m.running_var.cpu(), rv, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("B,C,H,W", [(3, 64, 9, 11), (2, 256, 5, 5)])
def test_abn_fused_residual_and_plane_bias(B, C, H, W):
    from ucd_amd.abn import InPlaceABN
    dev = torch.device("cuda:0")
    x = synth.t_normal(5, (B, C, H, W), stream=1)
    r = synth.t_normal(5, (B, C, H, W), stream=2)
    pb = synth.t_normal(5, (B, C, 1, 1), stream=3)
    g = synth.t_normal(5, (B, C, H, W), stream=4)
    m = InPlaceABN(C, activation="identity")
    st = synth.fill_state_dict(m.state_dict(), seed=3)
    m.load_state_dict(st)
    # CPU reference: identity-ABN -> + residual -> leaky_relu (modules/residual.py:84-97) and
    # out += pool -> ABN (modules/deeplab.py:68-69)
    w, b = st["weight"].clone().requires_grad_(True), st["bias"].clone().requires_grad_(True)
    xr, rr, pr = x.clone().requires_grad_(True), r.clone().requires_grad_(True), pb.clone().requires_grad_(True)
    yr = F.leaky_relu(_ref(xr, w, b, st["running_mean"].clone(), st["running_var"].clone(), True, "identity", 0.01,
                           plane_bias=pr, abs_gamma=True) + rr, 0.01)
    yr.backward(g)
    m = m.to(dev).train()
    cl = torch.channels_last
    xg = x.to(dev).contiguous(memory_format=cl).requires_grad_(True)
    rg = r.to(dev).contiguous(memory_format=cl).requires_grad_(True)
    pg = pb.to(dev).requires_grad_(True)
    yg = m(xg, residual=rg, activation="leaky_relu", activation_param=0.01, plane_bias=pg)
    yg.backward(g.to(dev))
    torch.cuda.synchronize()
    torch.testing.assert_close(yg.detach().cpu(), yr.detach(), rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(xg.grad.cpu(), xr.grad, rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(rg.grad.cpu(), rr.grad, rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(pg.grad.cpu(), pr.grad, rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(m.weight.grad.cpu(), w.grad, rtol=2e-4, atol=2e-4)


def test_abn_branches_equal_cat():
    from ucd_amd.abn import ABN
    dev = torch.device("cuda:0")
    B, H, W = 2, 7, 9
    xs = [synth.t_normal(20 + i, (B, 32, H, W), stream=1) for i in range(4)]
    g = synth.t_normal(9, (B, 128, H, W), stream=2)
    m = ABN(128)
    st = synth.fill_state_dict(m.state_dict(), seed=8)
    m.load_state_dict(st)
    w, b = st["weight"].clone().requires_grad_(True), st["bias"].clone().requires_grad_(True)
    xr = [x.clone().requires_grad_(True) for x in xs]
    yr = _ref(torch.cat(xr, 1), w, b, st["running_mean"].clone(), st["running_var"].clone(), True, "leaky_relu", 0.01)
    yr.backward(g)
    m = m.to(dev).train()
    xg = [x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True) for x in xs]
    yg = m.forward_branches(xg)
    yg.backward(g.to(dev))
    torch.cuda.synchronize()
    torch.testing.assert_close(yg.detach().cpu(), yr.detach(), rtol=2e-5, atol=2e-5)
    for a, r in zip(xg, xr):
        torch.testing.assert_close(a.grad.cpu(), r.grad, rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(m.weight.grad.cpu(), w.grad, rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(m.bias.grad.cpu(), b.grad, rtol=2e-4, atol=2e-4)


def test_abn_bf16_io():
    from ucd_amd.abn import ABN
    dev = torch.device("cuda:0")
    B, C, H, W = 4, 128, 13, 13
    x = synth.t_normal(31, (B, C, H, W), stream=1).bfloat16()
    g = synth.t_normal(31, (B, C, H, W), stream=2).bfloat16()
    m = ABN(C)
    st = synth.fill_state_dict(m.state_dict(), seed=31)
    m.load_state_dict(st)
    w, b = st["weight"].clone().requires_grad_(True), st["bias"].clone().requires_grad_(True)
    xr = x.float().requires_grad_(True)
    yr = _ref(xr, w, b, st["running_mean"].clone(), st["running_var"].clone(), True, "leaky_relu", 0.01)
    yr.backward(g.float())
    m = m.to(dev).train()
    xg = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    yg = m(xg)
    assert yg.dtype == torch.bfloat16
    yg.backward(g.to(dev))
    torch.cuda.synchronize()
    # outputs are rounded to bf16 once: half an ulp = 2^-9 relative
    torch.testing.assert_close(yg.detach().float().cpu(), yr.detach(), rtol=4e-3, atol=4e-3)
    torch.testing.assert_close(xg.grad.float().cpu(), xr.grad, rtol=8e-3, atol=8e-3)
    torch.testing.assert_close(m.weight.grad.cpu(), w.grad, rtol=2e-3, atol=2e-2)


def test_inplace_contract_and_eval_no_grad():
    from ucd_amd.abn import InPlaceABNSync
    dev = torch.device("cuda:0")
    m = InPlaceABNSync(64).to(dev).eval()
    x = synth.t_normal(40, (2, 64, 8, 8), stream=1).to(dev).contiguous(memory_format=torch.channels_last)
    keep = x.clone()
    with torch.no_grad():
        y = m(x)
    assert y.data_ptr() == x.data_ptr()          # overwritten in place, like inplace_abn
    ref = _ref(keep.cpu(), m.weight.detach().cpu(), m.bias.detach().cpu(), m.running_mean.cpu(), m.running_var.cpu(), False,
               "leaky_relu", 0.01, abs_gamma=True)
    torch.testing.assert_close(y.cpu(), ref, rtol=2e-5, atol=2e-5)


def test_plane_mean_and_attmap():
    from ucd_amd import hip
    from ucd_amd.abn import global_avg_pool
    dev = torch.device("cuda:0")
    x = synth.t_normal(50, (3, 64, 6, 5), stream=1)
    xg = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    p = global_avg_pool(xg)
    torch.testing.assert_close(p.detach().cpu(), x.mean(dim=(2, 3), keepdim=True), rtol=1e-5, atol=1e-6)
    p.sum().backward()
    torch.testing.assert_close(xg.grad.cpu(), torch.full_like(x, 1.0 / 30), rtol=1e-6, atol=1e-7)
    # attention map (segmentation_module.py:86-94)
    y = torch.empty_like(xg)
    xv, M, C, HW, ld = hip.rows_view(xg.detach())
    hip.attmap(xv, ld, y, C, 3, HW, C)
    a = (x ** 2).sum(1)
    a = a / a.flatten(1).norm(dim=1)[:, None, None]
    torch.testing.assert_close(y.cpu(), a.unsqueeze(1) * x, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_sync_forward_backward_kernels_match_global_batch(dtype):
    """The library calls around the SyncBN collectives (ucd_abn_sync_stats / _sync_forward / _sync_bwd_reduce), driven
    here for three 'ranks' held in one process, against batch norm over the concatenated batch (fp32 torch) and the
    oracle's combination formula."""
    from oracle.syncbn import combine_rank_moments
    from ucd_amd import hip
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    world, B, C, H, W = 3, 2, 64, 9, 7
    M, HW = B * H * W, H * W
    xs = [(torch.randn(B, C, H, W, device=dev) * 2 + 5).to(dtype).contiguous(memory_format=torch.channels_last)
          for _ in range(world)]
    weight = torch.rand(C, device=dev) + 0.5
    bias = torch.randn(C, device=dev)
    packs, bufs = [], []
    for x in xs:
        buf = torch.zeros(8 * C, device=dev)
        hip.abn_sync_stats(x, C, M, C, None, HW, buf[:2 * C], buf[2 * C:3 * C], buf[6 * C:])
        packs.append(buf[6 * C:].clone()); bufs.append(buf)
    gathered = torch.stack(packs).contiguous()                       # [world][2C] = what all_gather returns
    full = torch.cat([x.float() for x in xs])                        # the global batch
    ref_mean = full.mean(dim=(0, 2, 3)); ref_var = full.var(dim=(0, 2, 3), unbiased=False)
    om, ov, _ = combine_rank_moments(gathered.view(world, 2, C).cpu().numpy(), M)
    np.testing.assert_allclose(om, ref_mean.cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ov, ref_var.cpu().numpy(), rtol=1e-4, atol=1e-5)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    ys = []
    for x, buf in zip(xs, bufs):
        y = torch.empty_like(x)
        rm_r, rv_r = rm.clone(), rv.clone()
        hip.abn_sync_forward(x, C, y, C, None, 0, M, C, None, HW, gathered, world, weight, bias, rm_r, rv_r, 0.1, 1e-5, buf,
                             hip.ACT_CODES["leaky_relu"], 0.01)
        ys.append(y)
        np.testing.assert_allclose(buf[3 * C:4 * C].cpu().numpy(), om, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(buf[4 * C:5 * C].cpu().numpy(), 1 / np.sqrt(ov + 1e-5), rtol=1e-4)
    n = world * M
    np.testing.assert_allclose(rm_r.cpu().numpy(), 0.1 * om, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rv_r.cpu().numpy(), 0.9 + 0.1 * ov * n / (n - 1), rtol=1e-4)
    fr = full.clone().requires_grad_(True)
    wr, br = weight.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    yr = torch.nn.functional.leaky_relu(torch.nn.functional.batch_norm(fr, None, None, wr, br, True, 0.1, 1e-5), 0.01)
    tol = dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    np.testing.assert_allclose(torch.cat([y.float() for y in ys]).cpu().numpy(), yr.detach().cpu().numpy(), **tol)
    # backward: per-rank sums -> "all_reduce" -> bwd_apply with the global count
    dys = [torch.randn(B, C, H, W, device=dev).to(dtype).contiguous(memory_format=torch.channels_last) for _ in range(world)]
    yr.backward(torch.cat([d.float() for d in dys]))
    sums_r, local_r = [], []
    for x, dy, buf in zip(xs, dys, bufs):
        s4 = torch.zeros(4 * C, device=dev)
        hip.abn_sync_bwd_reduce(x, C, dy, C, None, 0, M, C, None, HW, buf[3 * C:4 * C], buf[4 * C:5 * C], buf[5 * C:6 * C], bias,
                                hip.ACT_CODES["leaky_relu"], 0.01, s4[:2 * C], s4[2 * C:])
        assert torch.equal(s4[:2 * C], s4[2 * C:])
        sums_r.append(s4[:2 * C]); local_r.append(s4[2 * C:])
    total = torch.stack(sums_r).sum(0)
    gtol = dict(rtol=1e-3, atol=1e-3) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-1)
    np.testing.assert_allclose(total[:C].cpu().numpy(), br.grad.cpu().numpy(), **gtol)
    np.testing.assert_allclose(total[C:].cpu().numpy(), wr.grad.cpu().numpy(), **gtol)
    dxs = []
    for x, dy, buf in zip(xs, dys, bufs):
        dx = torch.empty_like(x)
        hip.abn_bwd_apply(x, C, dy, C, None, 0, dx, C, None, 0, M, C, None, HW, buf[3 * C:4 * C], buf[4 * C:5 * C],
                          buf[5 * C:6 * C], bias, weight, total, float(n), 0, hip.ACT_CODES["leaky_relu"], 0.01)
        dxs.append(dx.float())
    got, ref = torch.cat(dxs), fr.grad
    assert ((got - ref).norm() / ref.norm()).item() < (1e-4 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("with_res", [False, True])
def test_cpp_autograd_node_equals_python_function(dtype, with_res):
    """ucd_amd/csrc/abn_node.cpp (the host path of the student's training-mode layers) against the Python autograd
    Function it shadows: same library calls, so outputs, gradients and running statistics are bit-identical."""
    from ucd_amd import abn
    node = abn._abn_node()
    assert node is not None, "the C++ autograd node is not built (python ucd_amd/csrc/build_node.py)"
    dev = torch.device("cuda:0")
    shape = (3, 64, 13, 11)
    x0 = synth.t_normal(21, shape, stream=1).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    r0 = synth.t_normal(22, shape, stream=1).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    dy = synth.t_normal(23, shape, stream=1).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    outs = []
    for use_node in (True, False):
        abn._node_mod = node if use_node else None
        try:
            m = abn.InPlaceABN(64).to(dev).train()
            with torch.no_grad():
                m.weight.copy_(torch.linspace(0.5, 1.5, 64)); m.bias.copy_(torch.linspace(-1, 1, 64))
            x = x0.clone().requires_grad_(True)
            r = r0.clone().requires_grad_(True) if with_res else None
            y = m(x * 1.0, residual=None if r is None else r * 1.0, activation="leaky_relu", activation_param=0.01)
            y.backward(dy)
            outs.append([y.detach(), x.grad, m.weight.grad, m.bias.grad, m.running_mean.clone(), m.running_var.clone()]
                        + ([r.grad] if with_res else []))
        finally:
            abn._node_mod = node
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_direct_parameter_gradients_under_the_reducer():
    """Under ucd_amd.ddp the ABN parameters' .grad views of a layer are laid out [d bias | d weight] and the backward
    kernels write them in place (no gradient tensors, no accumulate adds): same values as the plain autograd path."""
    from ucd_amd import abn
    from ucd_amd.ddp import DistributedDataParallel
    dev = torch.device("cuda:0")

    def net():
        torch.manual_seed(0)
        m = torch.nn.Sequential(torch.nn.Conv2d(8, 64, 3, padding=1, bias=False), abn.InPlaceABN(64),
                                torch.nn.Conv2d(64, 32, 1, bias=False), abn.InPlaceABN(32, activation="identity")).to(dev)
        m = m.to(memory_format=torch.channels_last)
        with torch.no_grad():
            m[1].weight.copy_(torch.linspace(0.5, 1.5, 64)); m[1].bias.copy_(torch.linspace(-1, 1, 64))
        return m.train()

    x = synth.t_normal(31, (4, 8, 12, 10), stream=1).to(dev).contiguous(memory_format=torch.channels_last)
    dy = synth.t_normal(32, (4, 32, 12, 10), stream=1).to(dev).contiguous(memory_format=torch.channels_last)
    plain = net()
    plain(x).backward(dy)
    wrapped = DistributedDataParallel(net())
    assert wrapped.reducer.direct_flat is not None and wrapped.module[1]._direct_grad_ptr() != 0
    for step in range(2):                      # twice: the second step must not accumulate on top of the first
        wrapped.zero_grad()
        wrapped(x).backward(dy)
        wrapped.finish_grad_sync()
        for (n, p), (_, q) in zip(plain.named_parameters(), wrapped.module.named_parameters()):
            assert q.grad is not None, n
            # convolution weight gradients come from MIOpen (atomics in its fp32 wrw solvers): looser than the ABN ones
            tol = dict(rtol=1e-5, atol=1e-6) if q.dim() == 1 else dict(rtol=1e-3, atol=1e-4)
            torch.testing.assert_close(q.grad, p.grad, msg=n, **tol)
    flat = wrapped.reducer.direct_flat
    assert torch.equal(flat[:64], wrapped.module[1].bias.grad) and torch.equal(flat[64:128], wrapped.module[1].weight.grad)


@pytest.mark.parametrize("shape", [(24, 256, 129, 129), (24, 2048, 33, 33), (24, 64, 257, 257)])
def test_full_size_batchnorm_invariants(shape):
    """The largest layers of the benchmark workload (up to 204 MB in bf16), through properties batch norm has at any size:
    with identity activation the output has per-channel mean = bias and variance = weight^2 (up to eps), and the input
    gradient is orthogonal to the constant and to the normalised input (sum dx = 0, sum dx*xhat = 0)."""
    from ucd_amd import abn
    dev = torch.device("cuda:0")
    B, C, H, W = shape
    g = torch.Generator(dev).manual_seed(C + H)
    x = (torch.randn(shape, device=dev, generator=g) * 1.5 + 0.7).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True)
    m = abn.InPlaceABN(C, activation="identity").to(dev).train()
    with torch.no_grad():
        m.weight.copy_(torch.linspace(0.5, 2.0, C)); m.bias.copy_(torch.linspace(-1, 1, C))
    y = m(x * 1.0)
    yf = y.float()
    mean, var = yf.mean(dim=(0, 2, 3)), yf.var(dim=(0, 2, 3), unbiased=False)
    torch.testing.assert_close(mean, m.bias.detach(), rtol=0, atol=2e-2)              # bf16 output rounding: 2^-9 relative
    torch.testing.assert_close(var.sqrt(), m.weight.detach(), rtol=1e-2, atol=1e-3)
    dy = torch.randn(shape, device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y.backward(dy)
    dx = x.grad.float()
    xf = x.detach().float()
    xhat = (xf - xf.mean(dim=(0, 2, 3), keepdim=True)) / xf.std(dim=(0, 2, 3), keepdim=True, unbiased=False)
    n = B * H * W
    scale = dx.abs().mean().item() * n
    assert (dx.sum(dim=(0, 2, 3)).abs().max().item()) / scale < 2e-3                  # bf16 dx: sums cancel to rounding noise
    assert ((dx * xhat).sum(dim=(0, 2, 3)).abs().max().item()) / scale < 2e-3


def _signed_state(m, seed):
    """A state dict whose gammas have both signs (as stored gammas of a trained inplace_abn checkpoint may)."""
    st = synth.fill_state_dict(m.state_dict(), seed=seed)
    sign = torch.where(synth.t_normal(seed + 1, st["weight"].shape, stream=9) < 0.3, -1.0, 1.0)
    st["weight"] = st["weight"] * sign
    assert (st["weight"] < 0).any() and (st["weight"] > 0).any()
    return st


@pytest.mark.parametrize("cls_name,abs_gamma", [("ABN", False), ("InPlaceABN", True), ("InPlaceABNSync", True)])
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("with_res", [False, True])
def test_negative_gamma_parameterisation(cls_name, abs_gamma, training, with_res):
    """Stored gammas of either sign: ABN is F.batch_norm (raw gamma); InPlaceABN / InPlaceABNSync normalise with
    |gamma| + eps and return d gamma = sign(gamma) * sum dz*xhat (inplace_abn's published algorithm).  Through the Python
    Function and (training, no plane bias) the C++ node, with and without the fused residual."""
    from ucd_amd import abn
    dev = torch.device("cuda:0")
    B, C, H, W = 3, 64, 9, 11
    x = synth.t_normal(61, (B, C, H, W), stream=1) * 1.3 + 0.2
    r = synth.t_normal(62, (B, C, H, W), stream=1)
    g = synth.t_normal(63, (B, C, H, W), stream=1)
    m = getattr(abn, cls_name)(C, activation="identity" if with_res else "leaky_relu")
    st = _signed_state(m, 17)
    m.load_state_dict(st)
    w, b = st["weight"].clone().requires_grad_(True), st["bias"].clone().requires_grad_(True)
    rm, rv = st["running_mean"].clone(), st["running_var"].clone()
    xr = x.clone().requires_grad_(True)
    rr = r.clone().requires_grad_(True) if with_res else None
    yr = _ref(xr, w, b, rm, rv, training, "identity" if with_res else "leaky_relu", 0.01, abs_gamma=abs_gamma)
    if with_res:
        yr = F.leaky_relu(yr + rr, 0.01)
    yr.backward(g)
    m = m.to(dev).train(training)
    cl = torch.channels_last
    xg = x.to(dev).contiguous(memory_format=cl).requires_grad_(True)
    rg = r.to(dev).contiguous(memory_format=cl).requires_grad_(True) if with_res else None
    yg = m(xg * 1.0, residual=None if rg is None else rg * 1.0, activation="leaky_relu", activation_param=0.01)
    yg.backward(g.to(dev))
    torch.cuda.synchronize()
    torch.testing.assert_close(yg.detach().cpu(), yr.detach(), rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(xg.grad.cpu(), xr.grad, rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(m.weight.grad.cpu(), w.grad, rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(m.bias.grad.cpu(), b.grad, rtol=2e-4, atol=2e-4)
    if with_res:
        torch.testing.assert_close(rg.grad.cpu(), rr.grad, rtol=2e-4, atol=2e-5)
    if abs_gamma and not with_res:     # the sign matters: the raw-gamma formula gives different activations
        y_raw = _ref(x, st["weight"], st["bias"], st["running_mean"].clone(), st["running_var"].clone(), training,
                     "leaky_relu", 0.01, abs_gamma=False)
        assert (y_raw - yr.detach()).abs().max() > 0.1


def test_negative_gamma_state_dict_whole_block_and_branches():
    """A ResidualBlock and the ASPP branch norm with mixed-sign gammas loaded through load_state_dict (the way a
    reference checkpoint arrives): the product's InPlaceABNSync layers against the oracle's |gamma| + eps."""
    from ucd_amd import abn
    dev = torch.device("cuda:0")
    B, H, W = 2, 7, 9
    xs = [synth.t_normal(70 + i, (B, 32, H, W), stream=1) for i in range(4)]
    g = synth.t_normal(79, (B, 128, H, W), stream=2)
    for training in (True, False):
        m = abn.InPlaceABNSync(128)
        st = _signed_state(m, 23)
        m.load_state_dict(st)
        w, b = st["weight"].clone().requires_grad_(True), st["bias"].clone().requires_grad_(True)
        xr = [x.clone().requires_grad_(True) for x in xs]
        yr = _ref(torch.cat(xr, 1), w, b, st["running_mean"].clone(), st["running_var"].clone(), training, "leaky_relu", 0.01,
                  abs_gamma=True)
        yr.backward(g)
        m = m.to(dev).train(training)
        xg = [x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True) for x in xs]
        yg = m.forward_branches(xg)
        yg.backward(g.to(dev))
        torch.cuda.synchronize()
        torch.testing.assert_close(yg.detach().cpu(), yr.detach(), rtol=2e-5, atol=2e-5)
        for a, r in zip(xg, xr):
            torch.testing.assert_close(a.grad.cpu(), r.grad, rtol=2e-4, atol=2e-5)
        if training:
            torch.testing.assert_close(m.weight.grad.cpu(), w.grad, rtol=2e-4, atol=2e-4)
            torch.testing.assert_close(m.bias.grad.cpu(), b.grad, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("with_res", [False, True])
def test_elu_activation(dtype, with_res):
    """activation="elu" (modules/residual.py:94-95 applies it after the residual add; inplace_abn's ACT_ELU)."""
    from ucd_amd import abn
    dev = torch.device("cuda:0")
    B, C, H, W = 2, 64, 9, 7
    x = synth.t_normal(81, (B, C, H, W), stream=1).to(dtype)
    r = synth.t_normal(82, (B, C, H, W), stream=1).to(dtype)
    g = synth.t_normal(83, (B, C, H, W), stream=1).to(dtype)
    m = abn.InPlaceABN(C, activation="identity" if with_res else "elu", activation_param=0.9)
    st = synth.fill_state_dict(m.state_dict(), seed=5)
    m.load_state_dict(st)
    w, b = st["weight"].clone().requires_grad_(True), st["bias"].clone().requires_grad_(True)
    xr = x.float().clone().requires_grad_(True)
    rr = r.float().clone().requires_grad_(True)
    yr = _ref(xr, w, b, st["running_mean"].clone(), st["running_var"].clone(), True, "identity" if with_res else "elu", 0.9,
              abs_gamma=True)
    if with_res:
        yr = F.elu(yr + rr, 0.9)
    yr.backward(g.float())
    m = m.to(dev).train()
    cl = torch.channels_last
    xg = x.to(dev).contiguous(memory_format=cl).requires_grad_(True)
    rg = r.to(dev).contiguous(memory_format=cl).requires_grad_(True)
    yg = m(xg * 1.0, residual=rg * 1.0 if with_res else None, activation="elu", activation_param=0.9)
    yg.backward(g.to(dev))
    torch.cuda.synchronize()
    tol = dict(rtol=2e-5, atol=2e-5) if dtype == torch.float32 else dict(rtol=8e-3, atol=8e-3)
    gtol = dict(rtol=2e-4, atol=5e-5) if dtype == torch.float32 else dict(rtol=1.6e-2, atol=1.6e-2)
    torch.testing.assert_close(yg.detach().float().cpu(), yr.detach(), **tol)
    torch.testing.assert_close(xg.grad.float().cpu(), xr.grad, **gtol)
    if with_res:
        torch.testing.assert_close(rg.grad.float().cpu(), rr.grad, **gtol)
    if dtype == torch.float32:
        torch.testing.assert_close(m.weight.grad.cpu(), w.grad, rtol=2e-4, atol=2e-4)
        torch.testing.assert_close(m.bias.grad.cpu(), b.grad, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("fused", [True, False])
def test_eval_constants_follow_training_and_optimizer_steps(fused):
    """eval -> train step (kernels update running_var through raw pointers, fused SGD updates gamma without bumping its
    version counter) -> eval: the cached evaluation-mode constants must be rebuilt (run.py validates after every epoch)."""
    from ucd_amd import abn
    dev = torch.device("cuda:0")
    C = 64
    m = abn.InPlaceABNSync(C).to(dev)
    m.load_state_dict(synth.fill_state_dict(m.state_dict(), seed=9))
    opt = torch.optim.SGD(m.parameters(), lr=0.5, momentum=0.9, fused=fused)
    x = (synth.t_normal(91, (4, C, 8, 8), stream=1) * 2 + 1).to(dev).contiguous(memory_format=torch.channels_last)

    def eval_out():
        m.eval()
        with torch.no_grad():
            y = m(x.clone())
        ref = _ref(x.cpu(), m.weight.detach().cpu(), m.bias.detach().cpu(), m.running_mean.cpu(), m.running_var.cpu(), False,
                   "leaky_relu", 0.01, abs_gamma=True)
        torch.testing.assert_close(y.cpu(), ref, rtol=2e-5, atol=2e-5)
        return y

    y0 = eval_out()
    for _ in range(2):
        m.train()
        opt.zero_grad()
        m(x.clone().requires_grad_(True) * 1.0).square().mean().backward()
        opt.step()
        y1 = eval_out()
        assert (y1 - y0).abs().max() > 1e-3          # the parameters and running statistics did move
        y0 = y1
    # a frozen layer (the teacher) keeps its cached constants across optimiser steps of OTHER parameters
    t = abn.InPlaceABNSync(C).to(dev).eval()
    for p in t.parameters():
        p.requires_grad = False
    with torch.no_grad():
        t(x.clone())
    c0 = t._eval_constants()
    opt.step()
    assert t._eval_constants() is c0


@pytest.mark.parametrize("zero_where", ["before", "after"])
def test_wrapper_under_the_references_plain_loop(zero_where):
    """``optim.zero_grad(); loss.backward(); optim.step()`` (train.py:104,137-138,149) on the wrapped model with NO
    reducer-specific call: the parameters follow the unwrapped model's trajectory, the kernel-written ABN gradients and the
    bucket views survive ``set_to_none``."""
    from ucd_amd import abn
    from ucd_amd.ddp import DistributedDataParallel
    dev = torch.device("cuda:0")

    def net():
        torch.manual_seed(0)
        m = torch.nn.Sequential(torch.nn.Conv2d(8, 64, 3, padding=1, bias=False), abn.InPlaceABN(64),
                                torch.nn.Conv2d(64, 32, 1, bias=False), abn.InPlaceABN(32, activation="identity")).to(dev)
        return m.to(memory_format=torch.channels_last).train()

    x = synth.t_normal(31, (4, 8, 12, 10), stream=1).to(dev).contiguous(memory_format=torch.channels_last)
    dy = synth.t_normal(32, (4, 32, 12, 10), stream=1).to(dev).contiguous(memory_format=torch.channels_last)
    plain, wrapped = net(), DistributedDataParallel(net())
    opts = [torch.optim.SGD(m.parameters(), lr=0.05, momentum=0.9, nesterov=True) for m in (plain, wrapped)]
    for step in range(3):
        for m, opt in zip((plain, wrapped), opts):
            if zero_where == "before":
                opt.zero_grad()
                y = m(x)
            else:
                y = m(x)
                opt.zero_grad()
            (y * dy).sum().backward()
            opt.step()
        for (n, p), (_, q) in zip(plain.named_parameters(), wrapped.module.named_parameters()):
            # MIOpen's weight-gradient solvers use atomics: the two trajectories drift apart by rounding noise
            torch.testing.assert_close(q, p, msg=f"{n} step {step}", rtol=1e-3, atol=1e-4)
    if zero_where == "before":        # the fast path stayed on: gradients live in the reducer's flat buffers
        assert wrapped.module[1]._direct_grad_ptr() != 0
        assert wrapped.module[0].weight.grad.data_ptr() == wrapped.reducer._bucket_of[wrapped.module[0].weight].flat.data_ptr() \
            or wrapped.module[0].weight.grad.untyped_storage().data_ptr() == wrapped.reducer._bucket_of[wrapped.module[0].weight].flat.untyped_storage().data_ptr()


@pytest.mark.parametrize("B,C,H,W,slope", [(2, 64, 65, 67, 0.01), (3, 64, 33, 32, 0.01), (2, 128, 9, 11, 1.0), (24, 64, 257, 257, 0.01)])
def test_stem_norm_pool_fused_equals_the_two_layers(B, C, H, W, slope):
    """mod1.bn1 + mod1.pool1 (models/resnet.py:58-64) as one kernel forward and two backward (csrc/stem.hip) against the two
    layers run one after the other (HIP ABN, then ATen's max_pool2d): the forward is BIT-identical (same apply arithmetic, values
    rounded to bf16 before the comparison, first maximum wins), the gradients agree to bf16 rounding, and both agree with the fp32
    composition batch_norm -> leaky_relu -> max_pool2d on the same bf16 input; evaluation mode with the running statistics too."""
    import torch.nn.functional as F
    from ucd_amd import abn
    dev = torch.device("cuda:0")
    z0 = (synth.t_normal(31 + H, (B, C, H, W), stream=1) * 1.5 + 0.3).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
    PH, PW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    dp = synth.t_normal(32 + H, (B, C, PH, PW), stream=2).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
    res = {}
    for mode in ("fused", "layers"):
        bn = abn.InPlaceABNSync(C, activation="leaky_relu", activation_param=slope).to(dev)
        with torch.no_grad():
            bn.weight.copy_(synth.t_normal(5, (C,), stream=3) * 0.3 + 1.0)
            bn.weight[1] = -0.7                                               # |gamma| + eps semantics on a negative stored scale
            bn.bias.copy_(synth.t_normal(6, (C,), stream=4) * 0.2)
        bn.train()
        z = z0.clone().requires_grad_(True)
        y = abn.stem_norm_pool(bn, z * 1.0) if mode == "fused" else F.max_pool2d(bn(z * 1.0), 3, 2, 1)
        assert y is not None and y.shape == (B, C, PH, PW)
        y.backward(dp)
        bn.eval()
        with torch.no_grad():
            ye = abn.stem_norm_pool(bn, z0.clone()) if mode == "fused" else F.max_pool2d(bn(z0.clone()), 3, 2, 1)
        res[mode] = (y.detach().clone(), z.grad.float(), bn.weight.grad.clone(), bn.bias.grad.clone(), bn.running_mean.clone(),
                     bn.running_var.clone(), ye.clone())
    a, b = res["fused"], res["layers"]
    assert torch.equal(a[0], b[0]) and torch.equal(a[6], b[6])                  # forward (train and eval): bit-identical
    torch.testing.assert_close(a[4], b[4], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(a[5], b[5], rtol=1e-5, atol=1e-6)
    rel = lambda u, v: ((u.float() - v.float()).norm() / v.float().norm()).item()
    assert rel(a[1], b[1]) < 1e-2 and rel(a[2], b[2]) < 2e-3 and rel(a[3], b[3]) < 2e-3
    # fp32 composition on the same input
    zf = z0.float().clone().requires_grad_(True)
    w = (res["layers"][2] * 0).clone()                                            # shapes only
    bn32 = abn.InPlaceABNSync(C, activation="leaky_relu", activation_param=slope).to(dev)
    with torch.no_grad():
        bn32.weight.copy_(synth.t_normal(5, (C,), stream=3) * 0.3 + 1.0); bn32.weight[1] = -0.7
        bn32.bias.copy_(synth.t_normal(6, (C,), stream=4) * 0.2)
    g = (bn32.weight.abs() + bn32.eps).detach().requires_grad_(True)
    beta = bn32.bias.detach().clone().requires_grad_(True)
    yf = F.max_pool2d(F.leaky_relu(F.batch_norm(zf, None, None, g, beta, True, 0.1, bn32.eps), slope), 3, 2, 1)
    yf.backward(dp.float())
    assert rel(a[0], yf.detach()) < 5e-3
    assert rel(a[1], zf.grad) < 3e-2          # bf16 ties in a window route the gradient to another (equal-valued) position
    sgn = torch.where(bn32.weight < 0, -1.0, 1.0)
    assert rel(a[2], g.grad * sgn) < 1e-2 and rel(a[3], beta.grad) < 1e-2
    del w


@pytest.mark.parametrize("B,H,W,nchw", [(2, 65, 65, False), (3, 129, 97, True), (24, 513, 513, False), (1, 7, 9, False)])
def test_stem_conv7x7_matches_conv2d(B, H, W, nchw):
    """ucd_stem_conv7x7 (models/resnet.py:58 conv1: 7x7, stride 2, padding 3, 3 -> 64, no bias) against F.conv2d: exact on small
    integers (every product and partial sum is exactly representable), then against the fp32 convolution of the bf16-rounded
    operands; image in either memory format, map sizes whose last tile holds one column."""
    from ucd_amd import hip
    DEV = "cuda:0"
    g = torch.Generator(DEV).manual_seed(B * H + W)
    cl = torch.channels_last
    xi = torch.randint(-3, 4, (B, 3, H, W), device=DEV, generator=g).float()
    wi = torch.randint(-2, 3, (64, 3, 7, 7), device=DEV, generator=g).float()
    if not nchw:
        xi = xi.contiguous(memory_format=cl)
    z = hip.stem_conv7x7(xi, wi.bfloat16().contiguous(memory_format=cl))
    if B * H * W <= 40000:
        # the reference on the CPU in float64 (the library's fp32 solvers - Winograd / FFT forms - are not exact on integers)
        ref = F.conv2d(xi.cpu().double(), wi.cpu().double(), None, 2, 3).float().to(DEV)
        assert z.shape == ref.shape and z.is_contiguous(memory_format=cl)
        assert torch.equal(z.float(), ref.bfloat16().float())
    else:
        ref = F.conv2d(xi, wi, None, 2, 3)
        assert z.shape == ref.shape and z.is_contiguous(memory_format=cl)
        assert ((z.float() - ref).norm() / ref.norm()).item() < 3e-3
    x = torch.randn(B, 3, H, W, device=DEV, generator=g) * 1.5 + 0.3
    w = torch.randn(64, 3, 7, 7, device=DEV, generator=g) * 0.1
    if not nchw:
        x = x.contiguous(memory_format=cl)
    z = hip.stem_conv7x7(x, w.bfloat16().contiguous(memory_format=cl))
    ref = F.conv2d(x.bfloat16().float(), w.bfloat16().float(), None, 2, 3)
    err = ((z.float() - ref).norm() / ref.norm()).item()
    assert err < 3e-3, err

@pytest.mark.parametrize("B,H,W,nchw,slope", [(2, 65, 65, False, 0.01), (3, 129, 97, True, 0.01), (24, 513, 513, False, 0.01),
                                              (1, 9, 11, False, 1.0), (2, 33, 61, False, 0.2), (1, 8, 8, False, 0.01)])
def test_stem_conv_norm_pool_in_one_kernel_equals_the_two_kernels(B, H, W, nchw, slope):
    """Round 5: ucd_stem_conv_pool (the frozen-statistics stem of the teacher: conv1 -> norm -> activation -> 3x3 / 2 max pool,
    models/resnet.py:58-64 in evaluation mode, as ONE kernel that never writes the convolution output) against ucd_stem_conv7x7
    followed by ucd_stem_apply_pool: bit-identical, at map sizes whose pooled tiles (3 x 15) end everywhere - one row, one column,
    full - and in both image memory formats."""
    from ucd_amd import hip
    DEV = "cuda:0"
    g = torch.Generator(DEV).manual_seed(B * H + W + 7)
    cl = torch.channels_last
    x = torch.randn(B, 3, H, W, device=DEV, generator=g) * 1.5 + 0.3
    if not nchw:
        x = x.contiguous(memory_format=cl)
    w = (torch.randn(64, 3, 7, 7, device=DEV, generator=g) * 0.1).bfloat16().contiguous(memory_format=cl)
    mean = torch.randn(64, device=DEV, generator=g) * 0.2
    scale = torch.rand(64, device=DEV, generator=g) + 0.5
    scale[3] = -0.8                                                      # a negative scale: the maximum moves to the smallest z
    beta = torch.randn(64, device=DEV, generator=g) * 0.3
    act = 1 if slope != 1.0 else 0
    z = hip.stem_conv7x7(x, w)
    ref, _ = hip.stem_apply_pool(z, mean, scale, beta, act, slope, False)
    out = hip.stem_conv_pool(x, w, mean, scale, beta, act, slope)
    assert out.shape == ref.shape and out.is_contiguous(memory_format=cl)
    assert torch.equal(out, ref), (out.float() - ref.float()).abs().max().item()
    out2 = hip.stem_conv_pool(x, w, mean, scale, None, act, slope)
    ref2, _ = hip.stem_apply_pool(z, mean, scale, None, act, slope, False)
    assert torch.equal(out2, ref2)


def test_teacher_stem_takes_the_one_kernel_path_and_matches():
    """The evaluation-mode stem of the backbone (ucd_amd/backbone.py::_stem) under no_grad + bf16 autocast goes through
    ucd_stem_conv_pool and returns what the two-kernel path (UCD_STEM_EVAL_FUSED=0) returns, bit for bit."""
    from functools import partial
    from ucd_amd import abn, backbone, hip, switches
    DEV = "cuda:0"
    norm = partial(abn.InPlaceABNSync, activation="leaky_relu", activation_param=0.01)
    net = backbone.ResNet([1, 1, 1, 1], True, norm_act=norm, output_stride=16) if hasattr(backbone, "ResNet") else None
    if net is None:
        pytest.skip("backbone.ResNet not exposed")
    net = net.to(DEV).to(memory_format=torch.channels_last).eval()
    with torch.no_grad():
        net.mod1.bn1.running_mean.normal_(0, 0.2); net.mod1.bn1.running_var.uniform_(0.5, 1.5)
        net.mod1.bn1.weight.normal_(1, 0.3); net.mod1.bn1.bias.normal_(0, 0.2)
    x = torch.randn(2, 3, 129, 129, device=DEV).contiguous(memory_format=torch.channels_last)
    calls = []
    orig = hip.stem_conv_pool
    hip.stem_conv_pool = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            y1 = net._stem(x)
            switches.set("UCD_STEM_EVAL_FUSED", "0")
            y0 = net._stem(x)
    finally:
        hip.stem_conv_pool = orig
        switches.unset("UCD_STEM_EVAL_FUSED")
    assert len(calls) == 1
    assert torch.equal(y1, y0)


_PACKED_VS_GENERIC = r"""
import sys, torch
from ucd_amd import hip
dev = torch.device("cuda:0")
torch.manual_seed(0)
outs = {}
for (B, C, H, W, slope, act) in ((3, 64, 33, 32, 0.01, 1), (2, 64, 65, 67, 0.01, 1), (2, 256, 33, 33, 0.01, 1), (3, 1024, 17, 19, 1.0, 0), (1, 8, 5, 7, 0.2, 1), (2, 128, 9, 11, 0.01, 1 | 0x100)):
    x = (torch.randn(B, C, H, W, device=dev) * 1.5 + 0.3).bfloat16().contiguous(memory_format=torch.channels_last)
    dy = torch.randn_like(x); r = torch.randn_like(x)
    M, HW = B * H * W, H * W
    buf = torch.zeros(6 * C, device=dev); w = torch.randn(C, device=dev) * 0.3 + 1; w[1] = -0.7; b = torch.randn(C, device=dev) * 0.2
    sums, ks, mean, invstd, scale = buf[:2*C], buf[2*C:3*C], buf[3*C:4*C], buf[4*C:5*C], buf[5*C:]
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    hip.abn_stats_finalize(x, C, M, C, None, HW, sums, ks, w, rm, rv, 0.1, 1e-5, mean, invstd, scale)
    y = torch.empty_like(x); y2 = torch.empty_like(x); y3 = torch.empty_like(x); dx = torch.empty_like(x); dz = torch.empty_like(x); dx2 = torch.empty_like(x); dx3 = torch.empty_like(x)
    hip.abn_apply(x, C, y, C, None, 0, M, C, None, HW, mean, scale, b, act, slope)
    hip.abn_apply(x, C, y2, C, r, C, M, C, None, HW, mean, scale, b, act, slope)
    hip.abn_apply(x, C, y3, C, None, 0, M, C, None, HW, mean, scale, None, act, slope)
    s2 = torch.zeros(2 * C, device=dev); s3 = torch.zeros(2 * C, device=dev)
    hip.abn_bwd_reduce(x, C, dy, C, None, 0, M, C, None, HW, mean, invstd, scale, b, act, slope, s2)
    hip.abn_bwd_apply(x, C, dy, C, None, 0, dx, C, None, 0, M, C, None, HW, mean, invstd, scale, b, w, s2, M, 0, act, slope)
    hip.abn_bwd_reduce(x, C, dy, C, y, C, M, C, None, HW, mean, invstd, scale, b, act, slope, s3)
    hip.abn_bwd_apply(x, C, dy, C, y, C, dx2, C, dz, C, M, C, None, HW, mean, invstd, scale, b, w, s3, M, 0, act, slope)
    hip.abn_bwd_apply(x, C, dy, C, y, C, dx3, C, None, 0, M, C, None, HW, mean, invstd, scale, b, w, s3, M, 1, act, slope)   # frozen
    torch.cuda.synchronize()
    outs[(B, C, H, W, act)] = [t.cpu() for t in (y, y2, y3, s2, dx, s3, dx2, dz, dx3)]
torch.save(outs, sys.argv[1])
"""


def test_packed_math_kernels_equal_the_per_element_kernels_bit_for_bit(tmp_path):
    """Round 4: the bf16 apply / backward passes of the step run on float pairs (csrc/abn.hip: abn_*_fast_kernel, packed-fp32
    instructions, the nullable operands as template flags).  Same operations in the same order as the per-element kernels
    (the library is built without fp contraction), so every output - activations, both backward sums, input gradients with the
    sign from x or from the stored output, dz, the frozen form - must be BIT-identical; UCD_ABN_GENERIC=1 (read once per process,
    hence the two child processes) selects the per-element kernels."""
    import subprocess
    import sys
    script = tmp_path / "packed_vs_generic.py"
    script.write_text(_PACKED_VS_GENERIC)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for mode in ("packed", "generic"):
        env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        env.pop("UCD_ABN_GENERIC", None)
        if mode == "generic":
            env["UCD_ABN_GENERIC"] = "1"
        out = tmp_path / f"{mode}.pt"
        subprocess.run([sys.executable, str(script), str(out)], env=env, check=True, timeout=600)
        res[mode] = torch.load(out)
    names = "y y_res y_noshift sums dx sums_y dx_y dz dx_frozen".split()
    for shape, tensors in res["packed"].items():
        for name, a, b in zip(names, tensors, res["generic"][shape]):
            assert torch.equal(a, b), f"{name} at {shape}: {(a.float() != b.float()).sum().item()} values differ"
