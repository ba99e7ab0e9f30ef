"""Oracle (test infrastructure, not product): functional CPU forward of the UCD segmentation network.

A *functional* restatement - plain functions over a flat ``{name: tensor}`` parameter dict with the
reference's ``state_dict`` key names - of
  * ``ResNet.forward`` (ResNet-101, output stride 16)             models/resnet.py:48-121
  * ``ResidualBlock.forward``                                     modules/residual.py:84-97
  * ``DeeplabV3.forward`` / ``_global_pooling``                   modules/deeplab.py:54-89
  * ``IncrementalSegmentationModule._network/forward/att_map/init_new_classifier``
                                                                  segmentation_module.py:86-136
  * the norm/activation layer: inplace_abn.{ABN, InPlaceABN, InPlaceABNSync} (third-party wheel,
    source not under /root/reference -> parity unpinned by the reference) restated as batch norm
    (biased batch variance, eps 1e-5, momentum 0.1) with the in-place variants' ``|weight| + eps`` scale
    (``--norm_act iabn_sync``, the reference's default) followed by ``leaky_relu(0.01)`` or identity.
Being functional (no nn.Module tree) it shares no code with the product's modules and doubles as a
check that the product's parameter names are the reference's.  Pinned by tests/golden/model_*.npz.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

EPS = 1e-5
MOMENTUM = 0.1
SLOPE = 0.01
STRUCTURE_101 = (3, 4, 23, 3)


ABS_GAMMA = True      # the network is built with --norm_act iabn_sync (InPlaceABNSync): gamma~ = |gamma| + eps, oracle/abn.py


def abn(x, P, name, training, activation="leaky_relu", slope=SLOPE):
    """Batch norm (batch statistics + running-stat update when ``training``) with the in-place variants' ``|weight| + eps``
    scale, then the activation (oracle/abn.py holds the layer's full restatement)."""
    w = P[name + ".weight"]
    y = F.batch_norm(x, P[name + ".running_mean"], P[name + ".running_var"], w.abs() + EPS if ABS_GAMMA else w,
                     P[name + ".bias"], training, MOMENTUM, EPS)
    if activation == "leaky_relu":
        return F.leaky_relu(y, slope)
    if activation == "identity":
        return y
    raise ValueError(activation)


def residual_block(x, P, name, stride, dilation, training, slope=SLOPE):
    """Bottleneck: 1x1 -> ABN -> 3x3(stride, dilation) -> ABN -> 1x1 -> ABN(identity); projection
    shortcut (1x1 stride + identity ABN) when present; sum; leaky_relu  (residual.py:84-97).
    ``slope``: the activation parameter of the block's norm_act layers (residual.py:92-93 reads it from bn1)."""
    if name + ".proj_conv.weight" in P:
        r = F.conv2d(x, P[name + ".proj_conv.weight"], stride=stride)
        r = abn(r, P, name + ".proj_bn", training, "identity")
    else:
        r = x
    c = name + ".convs."
    y = abn(F.conv2d(x, P[c + "conv1.weight"]), P, c + "bn1", training, slope=slope)
    y = F.conv2d(y, P[c + "conv2.weight"], stride=stride, padding=dilation, dilation=dilation)
    y = abn(y, P, c + "bn2", training, slope=slope)
    y = abn(F.conv2d(y, P[c + "conv3.weight"]), P, c + "bn3", training, "identity")
    return F.leaky_relu(y + r, slope)


def resnet_body(x, P, training, prefix="body.", structure=STRUCTURE_101, output_stride=16):
    dil = {16: (1, 1, 1, 2), 8: (1, 1, 2, 4)}[output_stride]
    x = F.conv2d(x, P[prefix + "mod1.conv1.weight"], stride=2, padding=3)          # resnet.py:58-61
    x = abn(x, P, prefix + "mod1.bn1", training)
    x = F.max_pool2d(x, 3, stride=2, padding=1)                                    # resnet.py:62-63
    for stage, depth in enumerate(structure):                                      # resnet.py:72-84
        for b in range(depth):
            stride = 2 if (dil[stage] == 1 and b == 0 and stage > 0) else 1        # resnet.py:97-101
            x = residual_block(x, P, f"{prefix}mod{stage + 2}.block{b + 1}", stride, dil[stage], training)
    return x


def deeplab_head(x, P, training, prefix="head.", pooling_size=32, output_stride=16, slope=SLOPE):
    dils = {16: (6, 12, 18), 8: (12, 24, 32)}[output_stride]
    branches = [F.conv2d(x, P[prefix + "map_convs.0.weight"])]
    for i, d in enumerate(dils, 1):
        branches.append(F.conv2d(x, P[prefix + f"map_convs.{i}.weight"], padding=d, dilation=d))
    out = abn(torch.cat(branches, dim=1), P, prefix + "map_bn", training, slope=slope)           # deeplab.py:56-57
    out = F.conv2d(out, P[prefix + "red_conv.weight"])                             # :58
    if training or pooling_size is None:                                           # :72-76
        pool = x.reshape(x.shape[0], x.shape[1], -1).mean(dim=-1)[:, :, None, None]
    else:                                                                          # :77-88
        ph, pw = min(pooling_size, x.shape[2]), min(pooling_size, x.shape[3])
        pad = ((pw - 1) // 2, (pw - 1) // 2 if pw % 2 == 1 else (pw - 1) // 2 + 1,
               (ph - 1) // 2, (ph - 1) // 2 if ph % 2 == 1 else (ph - 1) // 2 + 1)
        pool = F.pad(F.avg_pool2d(x, (ph, pw), stride=1), pad=pad, mode="replicate")
    pool = F.conv2d(pool, P[prefix + "global_pooling_conv.weight"])                # :61
    pool = abn(pool, P, prefix + "global_pooling_bn", training, slope=slope)                    # :62
    pool = F.conv2d(pool, P[prefix + "pool_red_conv.weight"])                      # :63
    if training or pooling_size is None:
        pool = pool.repeat(1, 1, x.shape[2], x.shape[3])                           # :65-66
    out = out + pool                                                               # :68
    return abn(out, P, prefix + "red_bn", training, slope=slope)                                # :69


def att_map(x):
    """segmentation_module.py:86-94: spatial attention ``a = sum_c x^2`` normalised per image by its
    Frobenius norm, detached, multiplied back."""
    a = (x ** 2).sum(dim=1)
    a = a / a.flatten(1).norm(dim=1)[:, None, None]
    return a.unsqueeze(1).detach() * x


def segmentation_forward(x, P, n_heads, training, pooling_size=32):
    """``IncrementalSegmentationModule.forward``: returns (logits upsampled to the input size,
    {"body", "pre_logits", "sem"}) (segmentation_module.py:95-108,125-136)."""
    x_b = resnet_body(x, P, training)
    x_pl = deeplab_head(x_b, P, training, pooling_size=pooling_size)
    sem = torch.cat([F.conv2d(x_pl, P[f"cls.{i}.weight"], P[f"cls.{i}.bias"]) for i in range(n_heads)], dim=1)
    logits = F.interpolate(sem, size=x.shape[-2:], mode="bilinear", align_corners=False)
    return logits, {"body": att_map(x_b), "pre_logits": att_map(x_pl), "sem": sem, "raw_body": x_b, "raw_pre_logits": x_pl}


def init_new_classifier(P, n_heads, n_new):
    """Balanced initialisation of the newest head (segmentation_module.py:111-123): weight <- the
    background row of head 0, bias <- b_bkg - log(n_new + 1), and head 0's background bias is
    overwritten with the same value."""
    w0 = P["cls.0.weight"][0]
    with torch.no_grad():
        new_bias = P["cls.0.bias"][0] - torch.log(torch.tensor([n_new + 1.0]))[0]   # float32 log
        P[f"cls.{n_heads - 1}.weight"].copy_(w0.expand_as(P[f"cls.{n_heads - 1}.weight"]))
        P[f"cls.{n_heads - 1}.bias"].fill_(float(new_bias))
        P["cls.0.bias"][0] = float(new_bias)
