"""CPU: the oracle's restatement of the third-party inplace_abn layer (oracle/abn.py; parity unpinned by the reference -
the wheel's source is not in the reference tree) against the wheel's published formulas written out by hand."""
import pytest
import torch

from oracle.abn import abn_forward
from ucd_amd import synth


@pytest.mark.parametrize("abs_gamma", [False, True])
@pytest.mark.parametrize("act,param", [("leaky_relu", 0.01), ("elu", 0.9), ("identity", 0.0)])
def test_oracle_abn_matches_published_formulas(abs_gamma, act, param):
    B, C, H, W = 3, 8, 5, 4
    x = synth.t_normal(1, (B, C, H, W), stream=1) * 2 + 0.5
    w = (synth.t_normal(2, (C,), stream=1) + 0.1).requires_grad_(True)        # both signs
    b = synth.t_normal(3, (C,), stream=1).requires_grad_(True)
    g = synth.t_normal(4, (B, C, H, W), stream=1)
    rm, rv = torch.zeros(C), torch.ones(C)
    xr = x.clone().requires_grad_(True)
    y = abn_forward(xr, w, b, rm, rv, True, 0.1, 1e-5, act, param, abs_gamma)
    y.backward(g)
    n = B * H * W
    mean = x.mean(dim=(0, 2, 3))
    var = x.var(dim=(0, 2, 3), unbiased=False)
    xhat = (x - mean[None, :, None, None]) * torch.rsqrt(var + 1e-5)[None, :, None, None]
    gamma = (w.detach().abs() + 1e-5) if abs_gamma else w.detach()
    z = xhat * gamma[None, :, None, None] + b.detach()[None, :, None, None]
    if act == "leaky_relu":
        yy, dz = torch.where(z > 0, z, z * param), g * torch.where(z > 0, 1.0, param)
    elif act == "elu":
        yy, dz = torch.where(z > 0, z, param * torch.expm1(z)), g * torch.where(z > 0, torch.ones_like(z), param * torch.exp(z))
    else:
        yy, dz = z, g
    torch.testing.assert_close(y.detach(), yy, rtol=1e-5, atol=1e-5)
    s1, s2 = dz.sum(dim=(0, 2, 3)), (dz * xhat).sum(dim=(0, 2, 3))
    torch.testing.assert_close(b.grad, s1, rtol=1e-4, atol=1e-4)
    sign = torch.where(w.detach() < 0, -1.0, 1.0) if abs_gamma else torch.ones(C)
    torch.testing.assert_close(w.grad, sign * s2, rtol=1e-4, atol=1e-4)
    dx = (dz - s1[None, :, None, None] / n - xhat * s2[None, :, None, None] / n) * (gamma * torch.rsqrt(var + 1e-5))[None, :, None, None]
    torch.testing.assert_close(xr.grad, dx, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rm, 0.1 * mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rv, 0.9 + 0.1 * var * n / (n - 1), rtol=1e-5, atol=1e-6)
