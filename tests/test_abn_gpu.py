"""GPU: the HIP ABN kernels (through the C ABI) against torch's batch_norm + leaky_relu on CPU fp32
(the documented semantics of inplace_abn.ABN - SURVEY.md section 8-c; tolerance 1e-5 abs/rel in
fp32, bf16 I/O checked against the same fp32 reference at bf16 resolution)."""
import pytest
import torch
import torch.nn.functional as F

from ucd_amd import synth

pytestmark = pytest.mark.gpu


def _ref(x, w, b, rm, rv, training, act, slope, residual=None, plane_bias=None, eps=1e-5, momentum=0.1):
    if plane_bias is not None:
        x = x + plane_bias
    y = F.batch_norm(x, rm, rv, w, b, training, momentum, eps)
    if residual is not None:
        y = y + residual
    return F.leaky_relu(y, slope) if act == "leaky_relu" else y


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
                           plane_bias=pr) + rr, 0.01)
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
    ref = F.leaky_relu(F.batch_norm(keep.cpu(), m.running_mean.cpu(), m.running_var.cpu(), m.weight.detach().cpu(),
                                    m.bias.detach().cpu(), False, 0.1, 1e-5), 0.01)
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
