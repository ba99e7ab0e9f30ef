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


def test_cpp_gemm_node_equals_python_function():
    """The C++ autograd node of the 1x1-conv GEMM (ucd_amd/csrc/abn_node.cpp: Gemm1x1Node) issues the same library calls as
    the Python Function it shadows: outputs and gradients are bit-identical."""
    from ucd_amd import blocks
    dev = torch.device("cuda:0")
    node = blocks._gemm_node()
    assert node is not None, "C++ node not built (python ucd_amd/csrc/build_node.py)"
    torch.manual_seed(3)
    outs = []
    for use_node in (True, False):
        blocks._node_cache[0] = node if use_node else None
        try:
            conv = Conv1x1(1024, 256).to(dev)
            with torch.no_grad():
                conv.weight.copy_(torch.linspace(-1, 1, conv.weight.numel()).view_as(conv.weight) * 0.05)
            x = torch.randn(3, 1024, 33, 33, device=dev, generator=torch.Generator(dev).manual_seed(5)).to(torch.bfloat16) \
                .contiguous(memory_format=torch.channels_last).requires_grad_(True)
            dy = torch.randn(3, 256, 33, 33, device=dev, generator=torch.Generator(dev).manual_seed(6)).to(torch.bfloat16) \
                .contiguous(memory_format=torch.channels_last)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = conv(x)
            y.backward(dy)
            outs.append((y.detach().clone(), x.grad.clone(), conv.weight.grad.clone()))
        finally:
            blocks._node_cache[0] = node
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("ci,co,hw,d,B", [(256, 256, 33, 1, 8), (512, 512, 33, 2, 8), (2048, 256, 33, 12, 8), (64, 64, 65, 1, 8),
                                           (2048, 256, 33, 12, 24), (256, 256, 33, 1, 24), (2048, 256, 33, 6, 24),
                                           (2048, 256, 33, 18, 24)])
def test_conv3x3_input_gradient_on_the_forward_solver(ci, co, hw, d, B):
    """ucd_amd/blocks.py::Conv3x3 computes dx with the forward solver on the flipped / transposed weight: the same
    arithmetic as conv2d's own backward (reference: nn.Conv2d(k=3, padding=dilation), modules/residual.py:69,
    modules/deeplab.py:27-29), compared in fp32 on bf16-rounded operands."""
    from ucd_amd.blocks import Conv3x3
    dev = torch.device("cuda:0")
    torch.manual_seed(ci + co + d)
    # B = 24: maps large enough for the own implicit-GEMM kernel (forward and input gradient of the stand-alone layer: the
    # ASPP branches); B = 8: MIOpen's forward solver for both
    conv = Conv3x3(ci, co, 3, stride=1, padding=d, dilation=d, bias=False).to(dev).to(memory_format=torch.channels_last)
    x32 = torch.randn(B, ci, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    dy32 = torch.randn(B, co, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    x = x32.to(torch.bfloat16).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = conv(x)
    assert "StrideOneConv" in y.grad_fn.name(), y.grad_fn.name()
    y.backward(dy32.to(torch.bfloat16))
    xr = x32.to(torch.bfloat16).float().requires_grad_(True)
    wr = conv.weight.detach().to(torch.bfloat16).float().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, 1, d, d)
    yr.backward(dy32.to(torch.bfloat16).float())
    rel = lambda a, b: ((a.float() - b).norm() / b.norm()).item()
    assert rel(y, yr) < 1e-2 and rel(x.grad, xr.grad) < 1e-2 and rel(conv.weight.grad, wr.grad) < 1e-2


@pytest.mark.parametrize("K,N,hw,d,B", [(64, 128, 33, 18, 3), (64, 128, 33, 12, 5), (128, 256, 33, 6, 4), (64, 64, 33, 16, 2),
                                         (64, 128, 33, 32, 2), (128, 256, 17, 12, 9), (256, 256, 33, 18, 24)])
def test_dilated_3x3_skips_only_kernel_rows_that_read_padding(K, N, hw, d, B):
    """Round 5: a tile of the implicit-GEMM 3x3 kernels drops the K steps of a kernel ROW whose every tap reads padding for all of the
    tile's rows (csrc/conv1x1.hip live_taps: dy = -d for tiles within d rows of the top edge, dy = +d near the bottom edge) - 12 / 24 /
    33 % of the work of the ASPP branches (modules/deeplab.py:27-29, dilations 6 / 12 / 18 on 33 x 33).  Every output element against
    F.conv2d on the bf16-rounded operands, at batch sizes whose 128- and 256-row tiles start at every phase of the image (tiles
    inside one image, tiles across an image boundary, the last partial tile), dilations up to the map size."""
    from ucd_amd import hip
    dev = torch.device("cuda:0")
    g = torch.Generator(dev).manual_seed(K + N + d + B)
    cl = torch.channels_last
    x = torch.randn(B, K, hw, hw, device=dev, generator=g).bfloat16().contiguous(memory_format=cl)
    w = (torch.randn(N, K, 3, 3, device=dev, generator=g) * (2.0 / (9 * K)) ** 0.5).bfloat16().contiguous(memory_format=cl)
    y = torch.empty(B, N, hw, hw, device=dev, dtype=torch.bfloat16).contiguous(memory_format=cl)
    rows = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])
    hip.conv1x1(rows(x), w.permute(0, 2, 3, 1).reshape(N, 9 * K), rows(y), conv3=(hw, hw, d))
    ref = F.conv2d(x.float(), w.float(), None, 1, d, d)
    err = (y.float() - ref).abs().max().item()
    assert err < 0.03 * ref.abs().max().item(), (err, ref.abs().max().item())       # one bf16 rounding of the output
    # and the input gradient (the same kernel on the flipped / transposed weight)
    dy = torch.randn(B, N, hw, hw, device=dev, generator=g).bfloat16().contiguous(memory_format=cl)
    wt = w.flip(2, 3).transpose(0, 1).contiguous(memory_format=cl)
    dx = torch.empty_like(x)
    hip.conv1x1(rows(dy), wt.permute(0, 2, 3, 1).reshape(K, 9 * N), rows(dx), conv3=(hw, hw, d))
    refdx = torch.nn.grad.conv2d_input(x.shape, w.float(), dy.float(), 1, d, d)
    err = (dx.float() - refdx).abs().max().item()
    assert err < 0.03 * refdx.abs().max().item(), (err, refdx.abs().max().item())


@pytest.mark.parametrize("ci,co,hw", [(256, 64, 65), (64, 256, 65), (512, 128, 33), (256, 40, 65), (72, 256, 33)])
def test_narrow_conv1x1_standalone_forward_and_gradients(ci, co, hw):
    """A narrow 1x1 convolution called on its own (outside the conv+ABN node): 64-aligned channel counts run on the row
    matrix like the wide layers; the others stay with MIOpen, their input gradient on the forward solver (transposed
    weight).  Either way: the arithmetic of F.conv2d on the bf16-rounded operands."""
    dev = torch.device("cuda:0")
    torch.manual_seed(ci + co)
    B = 8
    conv = Conv1x1(ci, co).to(dev).to(memory_format=torch.channels_last)
    aligned = ci % 64 == 0 and co % 64 == 0
    assert conv.as_gemm == aligned and not conv.wide
    x32 = torch.randn(B, ci, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    dy32 = torch.randn(B, co, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    x = x32.to(torch.bfloat16).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = conv(x)
    assert aligned or "StrideOneConv" in y.grad_fn.name(), y.grad_fn.name()
    y.backward(dy32.to(torch.bfloat16))
    xr = x32.to(torch.bfloat16).float().requires_grad_(True)
    wr = conv.weight.detach().to(torch.bfloat16).float().requires_grad_(True)
    yr = F.conv2d(xr, wr)
    yr.backward(dy32.to(torch.bfloat16).float())
    rel = lambda a, b: ((a.float() - b).norm() / b.norm()).item()
    assert rel(y, yr) < 1e-2 and rel(x.grad, xr.grad) < 1e-2 and rel(conv.weight.grad, wr.grad) < 1e-2


def test_block_shortcut_gradient_folded_into_the_first_conv():
    """Identity-shortcut bottleneck block (modules/residual.py:84-97): with the C++ skip node the shortcut's gradient is
    accumulated by the input-gradient GEMM of conv1 (beta = 1) instead of a separate autograd add - same block output,
    same gradients (bf16 rounding of one vs two roundings apart)."""
    from functools import partial
    from ucd_amd import abn, blocks
    dev = torch.device("cuda:0")
    node = blocks._gemm_node()
    assert node is not None and hasattr(node, "gemm1x1_skip")
    x0 = torch.randn(6, 1024, 33, 33, device=dev, generator=torch.Generator(dev).manual_seed(1)).to(torch.bfloat16) \
        .contiguous(memory_format=torch.channels_last)
    dy = torch.randn(6, 1024, 33, 33, device=dev, generator=torch.Generator(dev).manual_seed(2)).to(torch.bfloat16) \
        .contiguous(memory_format=torch.channels_last)
    outs = []
    for use_node in (True, False):
        blocks._node_cache[0] = node if use_node else None
        try:
            torch.manual_seed(7)
            blk = blocks.ResidualBlock(1024, [256, 256, 1024], norm_act=partial(abn.InPlaceABN, activation="leaky_relu",
                                                                               activation_param=0.01)).to(dev)
            blk = blk.to(memory_format=torch.channels_last).train()
            x = x0.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = blk(x * 1.0)
            y.backward(dy)
            outs.append((y.detach().float(), x.grad.float(), blk.convs.conv1.weight.grad.float(), blk.convs.conv3.weight.grad.float()))
        finally:
            blocks._node_cache[0] = node
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    # two bf16 implementations of the same block: with the node the 1x1 products come from the fused GEMM (statistics in
    # its epilogue), without it from the library GEMM + separate statistics kernels - every stored map differs by bf16
    # rounding (2^-9), amplified by the three normalisations of the block
    assert rel(outs[0][0], outs[1][0]) < 1e-2, rel(outs[0][0], outs[1][0])
    assert rel(outs[0][1], outs[1][1]) < 3e-2
    assert rel(outs[0][2], outs[1][2]) < 3e-2 and rel(outs[0][3], outs[1][3]) < 3e-2
