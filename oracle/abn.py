"""Oracle (test infrastructure, not product): the norm + activation layer of the reference's network.

PARITY UNPINNED BY THE REFERENCE.  The reference delegates this layer to the third-party wheel ``inplace-abn==1.0.7``
(``requirements.txt:38``; selected at ``segmentation_module.py:15-20``, instantiated at ``models/resnet.py:60``,
``modules/residual.py:51,56,64,68,71,81``, ``modules/deeplab.py:30,33,37``); its source is not under /root/reference
and the package is not installed here, so nothing the reference ships can pin these numbers.  This file restates the
wheel's published algorithm (mapillary/inplace_abn 1.0.x):

  ``ABN``            ``F.batch_norm(x, running_mean, running_var, weight, bias, training, momentum, eps)`` followed by
                     the activation - raw ``weight``;
  ``InPlaceABN`` /   statistics: mean and BIASED variance over (B, H, W) [all ranks for Sync]; running statistics updated
  ``InPlaceABNSync`` with ``momentum`` and the UNBIASED variance (count / (count - 1)); forward
                     ``y = act((x - mean) * rsqrt(var + eps) * (|weight| + eps) + bias)``; the backward recovers xhat from
                     y and returns ``d weight = sign(weight) * sum dz*xhat`` (negated where ``weight < 0``), ``d bias =
                     sum dz``; evaluation mode uses the running statistics in the same formula;
  activations        ``leaky_relu(slope)``, ``elu(alpha)``, ``identity``.

Written with differentiable torch ops, so autograd supplies the backward the tests compare with (``d|w|/dw`` =
sign(w); torch uses 0 at w = 0 where the wheel uses +1 - a measure-zero difference the tests avoid).
"""
from __future__ import annotations

import torch.nn.functional as F


def activation(z, name, param):
    if name == "leaky_relu":
        return F.leaky_relu(z, param)
    if name == "elu":
        return F.elu(z, param)
    if name == "identity":
        return z
    raise ValueError(name)


def abn_forward(x, weight, bias, running_mean, running_var, training, momentum=0.1, eps=1e-5, act="leaky_relu",
                param=0.01, abs_gamma=False, residual=None, plane_bias=None):
    """``act(norm(x [+ plane_bias]) [+ residual])``; ``running_mean`` / ``running_var`` are updated in place when
    ``training``.  ``abs_gamma`` selects the in-place variants' ``|weight| + eps``.  ``residual`` and ``plane_bias`` are the
    glue the product fuses into the layer (modules/residual.py:90-97, modules/deeplab.py:65-68)."""
    if plane_bias is not None:
        x = x + plane_bias
    gamma = None if weight is None else (weight.abs() + eps if abs_gamma else weight)
    y = F.batch_norm(x, running_mean, running_var, gamma, bias, training, momentum, eps)
    if residual is not None:
        y = y + residual
    return activation(y, act, param)
