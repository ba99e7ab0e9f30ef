"""GPU: the wide 1x1 convolutions as row-matrix GEMMs (ucd_amd/blocks.py: Conv1x1, split-K weight gradient) against
the plain fp32 convolution of the reference (modules/residual.py:57-63 builds them as nn.Conv2d(k=1)).
Tolerance: bf16 operands, fp32 accumulation -> 1e-2 relative L2 on outputs and gradients."""
import pytest
import torch
import torch.nn.functional as F

from ucd_amd.blocks import Conv1x1, _wgrad_split

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,hw,ci,co", [(24, 33, 1024, 256), (8, 33, 256, 1024), (3, 33, 2048, 512), (2, 17, 1024, 2048)])
def test_conv1x1_gemm_path_matches_conv2d(B, hw, ci, co):
    dev = torch.device("cuda:0")
    torch.manual_seed(ci + co + B)
    conv = Conv1x1(ci, co).to(dev)
    assert conv.as_gemm
    x32 = torch.randn(B, ci, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    dy32 = torch.randn(B, co, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    x = x32.to(torch.bfloat16).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = conv(x)
    assert y.dtype == torch.bfloat16 and y.shape == (B, co, hw, hw)
    y.backward(dy32.to(torch.bfloat16))
    xr = x32.to(torch.bfloat16).float().requires_grad_(True)
    wr = conv.weight.detach().to(torch.bfloat16).float().requires_grad_(True)
    yr = F.conv2d(xr, wr)
    yr.backward(dy32.to(torch.bfloat16).float())
    rel = lambda a, b: ((a.float() - b).norm() / b.norm()).item()
    assert rel(y, yr) < 1e-2
    assert rel(x.grad, xr.grad) < 1e-2
    assert conv.weight.grad.dtype == torch.float32
    assert rel(conv.weight.grad, wr.grad) < 1e-2
    assert _wgrad_split(24 * 33 * 33) == 8 and _wgrad_split(3 * 33 * 33) == 1
