"""Test helpers: a stock-PyTorch ABN stand-in so the product's module tree (names, wiring, literal
fallback paths) can be exercised on CPU without the HIP library."""
from functools import partial

import torch.nn as nn
import torch.nn.functional as F

from ucd_amd.backbone import net_resnet101
from ucd_amd.blocks import DeeplabV3


class ShimABN(nn.BatchNorm2d):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, activation="leaky_relu",
                 activation_param=0.01):
        super().__init__(num_features, eps=eps, momentum=momentum, affine=affine)
        self.activation, self.activation_param = activation, activation_param

    def forward(self, x):
        # InPlaceABN semantics (|weight| + eps), like the stand-in the goldens were captured with
        y = F.batch_norm(x, self.running_mean, self.running_var, self.weight.abs() + self.eps, self.bias, self.training,
                         self.momentum, self.eps)
        return F.leaky_relu(y, self.activation_param) if self.activation == "leaky_relu" else y


class _Net(nn.Module):
    def __init__(self, classes):
        super().__init__()
        norm = partial(ShimABN, activation="leaky_relu", activation_param=0.01)
        self.body = net_resnet101(norm_act=norm, output_stride=16)
        self.head = DeeplabV3(2048, 256, 256, norm_act=norm, out_stride=16, pooling_size=32)
        self.cls = nn.ModuleList([nn.Conv2d(256, c, 1) for c in classes])


def build_cpu_net(classes):
    return _Net(classes)
