"""Oracle (test infrastructure, not product): parameter-name/shape table of the UCD network, written out
by rule (ResNet-101 body, DeepLab-V3 head, per-step 1x1 classifiers) with the reference's state_dict
keys - models/resnet.py:56-89, modules/residual.py:45-82, modules/deeplab.py:24-37,
segmentation_module.py:72-74 - so the oracle needs nothing from the product package."""
from __future__ import annotations

import torch

from . import step as OS
from . import model as OM


def _abn(d, name, c):
    d[name + ".weight"] = torch.ones(c)
    d[name + ".bias"] = torch.zeros(c)
    d[name + ".running_mean"] = torch.zeros(c)
    d[name + ".running_var"] = torch.ones(c)


def template_state(classes, structure=(3, 4, 23, 3), hidden=256):
    d = {}
    d["body.mod1.conv1.weight"] = torch.zeros(64, 3, 7, 7)
    _abn(d, "body.mod1.bn1", 64)
    cin, width = 64, (64, 64, 256)
    for stage, depth in enumerate(structure):
        for b in range(depth):
            p = f"body.mod{stage + 2}.block{b + 1}"
            stride = 2 if (stage in (1, 2) and b == 0) else 1      # output stride 16: mod5 is dilated instead
            d[p + ".convs.conv1.weight"] = torch.zeros(width[0], cin, 1, 1)
            _abn(d, p + ".convs.bn1", width[0])
            d[p + ".convs.conv2.weight"] = torch.zeros(width[1], width[0], 3, 3)
            _abn(d, p + ".convs.bn2", width[1])
            d[p + ".convs.conv3.weight"] = torch.zeros(width[2], width[1], 1, 1)
            _abn(d, p + ".convs.bn3", width[2])
            if stride != 1 or cin != width[2]:
                d[p + ".proj_conv.weight"] = torch.zeros(width[2], cin, 1, 1)
                _abn(d, p + ".proj_bn", width[2])
            cin = width[2]
        width = tuple(2 * c for c in width)
    d["head.map_convs.0.weight"] = torch.zeros(hidden, cin, 1, 1)
    for i in (1, 2, 3):
        d[f"head.map_convs.{i}.weight"] = torch.zeros(hidden, cin, 3, 3)
    _abn(d, "head.map_bn", 4 * hidden)
    d["head.global_pooling_conv.weight"] = torch.zeros(hidden, cin, 1, 1)
    _abn(d, "head.global_pooling_bn", hidden)
    d["head.red_conv.weight"] = torch.zeros(256, 4 * hidden, 1, 1)
    d["head.pool_red_conv.weight"] = torch.zeros(256, hidden, 1, 1)
    _abn(d, "head.red_bn", 256)
    for i, c in enumerate(classes):
        d[f"cls.{i}.weight"] = torch.zeros(c, 256, 1, 1)
        d[f"cls.{i}.bias"] = torch.zeros(c)
    return d


def student_teacher_params(classes, seed=42, fill=None):
    """(student, teacher) oracle parameter dicts initialised like run.py:207-233 does from a step
    checkpoint: both load the same (synthetic) previous-step weights, the student's new head is
    initialised by ``init_new_classifier``; the teacher is frozen."""
    from ucd_amd import synth          # closed-form input generator only (no compute path)
    sd = (fill or synth.fill_state_dict)(template_state(classes[:-1]), seed)
    Pt = OS.make_params(sd, requires_grad=False)
    st = template_state(classes)
    # fresh head of the current step: small deterministic values, then the balanced initialisation
    new = {k: v for k, v in synth.fill_state_dict(st, seed + 1).items() if k.startswith(f"cls.{len(classes) - 1}.")}
    st.update({k: v.clone() for k, v in sd.items()})
    st.update(new)
    Ps = OS.make_params(st)
    Ps["cls.0.weight"].requires_grad_(False)
    Ps["cls.0.bias"].requires_grad_(False)
    OM.init_new_classifier(Ps, len(classes), classes[-1])
    return Ps, Pt
