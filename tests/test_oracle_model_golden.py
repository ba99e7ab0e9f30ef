"""CPU: the functional oracle network (oracle/model.py, oracle/step.py) reproduces the golden vectors
captured from the reference's models/ + modules/ + segmentation_module.py (ABN = BatchNorm2d +
leaky_relu stand-in) and from the hand-composed UCD step."""
import numpy as np
import pytest
import torch

from conftest import load_golden, sample_idx
from oracle import model as OM
from oracle import step as OS
from ucd_amd import synth
from ucd_amd.blocks import DeeplabV3, ResidualBlock
from oracle_shims import ShimABN, build_cpu_net


def test_blocks_and_head_match_reference():
    g = load_golden("model_blocks.npz")
    from functools import partial
    norm = partial(ShimABN, activation="leaky_relu", activation_param=0.01)
    # parameter dicts with the reference's key names come from the product's module tree (names only)
    blk = ResidualBlock(32, (16, 16, 64), norm_act=norm, stride=2, dilation=1)
    blk2 = ResidualBlock(64, (16, 16, 64), norm_act=norm, stride=1, dilation=2)
    P1 = {"b." + k: v for k, v in synth.fill_state_dict(blk.state_dict(), 11).items()}
    P2 = {"b." + k: v for k, v in synth.fill_state_dict(blk2.state_dict(), 12).items()}
    x = synth.t_normal(400, (3, 32, 13, 13), stream=1)
    for mode in ("train", "eval"):
        y = OM.residual_block(x, P1, "b", 2, 1, mode == "train")
        y = OM.residual_block(y, P2, "b", 1, 2, mode == "train")
        np.testing.assert_allclose(y.numpy(), g[f"block_{mode}"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(P1["b.convs.bn1.running_mean"].numpy(), g["block_rm_after"], rtol=1e-6, atol=1e-7)

    head = DeeplabV3(48, 24, 16, norm_act=norm, out_stride=16, pooling_size=4)
    Ph = {"head." + k: v for k, v in synth.fill_state_dict(head.state_dict(), 13).items()}
    xh = synth.t_normal(401, (2, 48, 7, 9), stream=1)
    np.testing.assert_allclose(OM.deeplab_head(xh, Ph, True, pooling_size=4).numpy(), g["head_train"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(OM.deeplab_head(xh, Ph, False, pooling_size=4).numpy(), g["head_eval"], rtol=1e-5, atol=1e-5)


def _student_teacher_params(seed=42, classes=(16, 5), calibrated=False):
    classes = list(classes)
    teacher = build_cpu_net(classes[:-1])
    student = build_cpu_net(classes)
    sd = synth.fill_state_dict(teacher.state_dict(), seed, calibrated=calibrated)
    Pt = OS.make_params(sd, requires_grad=False)
    st = {k: v.clone() for k, v in student.state_dict().items()}
    st.update({k: v.clone() for k, v in sd.items()})
    Ps = OS.make_params(st)
    OM.init_new_classifier(Ps, len(classes), classes[-1])
    return Ps, Pt


def test_full_network_matches_reference():
    g = load_golden("model_full.npz")
    Ps, Pt = _student_teacher_params()
    img = synth.images(500, 2, 65)
    with torch.no_grad():
        lt, ft = OM.segmentation_forward(img, Pt, 1, training=False)
        np.testing.assert_allclose(ft["sem"].numpy(), g["teacher_sem"], rtol=2e-4, atol=2e-4)
        assert lt.double().abs().sum().item() == pytest.approx(float(g["teacher_logits_abs"]), rel=1e-4)
        assert ft["pre_logits"].double().abs().sum().item() == pytest.approx(float(g["teacher_pl_abs"]), rel=1e-4)
        assert ft["body"].double().abs().sum().item() == pytest.approx(float(g["teacher_body_abs"]), rel=1e-4)
        ls_eval, fs_eval = OM.segmentation_forward(img, Ps, 2, training=False)
        np.testing.assert_allclose(fs_eval["sem"].numpy(), g["student_eval_sem"], rtol=2e-4, atol=2e-4)
        ls, fs = OM.segmentation_forward(img, Ps, 2, training=True)
    np.testing.assert_allclose(fs["sem"].numpy(), g["student_train_sem"], rtol=5e-4, atol=5e-4)
    np.testing.assert_allclose(ls.flatten()[g["sample_idx"]].numpy(), g["student_train_logits_sample"], rtol=5e-4, atol=5e-4)
    np.testing.assert_allclose(Ps["cls.1.bias"].detach().numpy(), g["new_head_bias"], rtol=1e-6)
    assert Ps["cls.0.bias"][0].item() == pytest.approx(float(g["head0_bias0"]), rel=1e-6)


def test_ucd_step_matches_reference():
    g = load_golden("ucd_step.npz")
    Ps, Pt = _student_teacher_params()
    img = synth.images(501, 2, 129)
    labels = synth.seg_labels(501, 2, 129, 129, range(16, 21))
    r = OS.ucd_losses(Ps, Pt, img, labels, [16, 5])
    assert r["A"] == int(g["A"]) and r["C"] == int(g["C"])
    assert r["ce"].item() == pytest.approx(float(g["ce"]), rel=1e-4)
    assert r["con"].item() == pytest.approx(float(g["con"]), rel=1e-4)
    assert r["loss"].item() == pytest.approx(float(g["loss"]), rel=1e-4)
    assert r["lkd"].item() == pytest.approx(float(g["lkd"]), rel=1e-4)
    (r["loss"] + r["lkd"]).backward()
    names = [k.split("::")[1] for k in g if k.startswith("grad_abs::")]
    for n in names:
        gr = Ps[n].grad
        assert gr.double().abs().sum().item() == pytest.approx(float(g[f"grad_abs::{n}"]), rel=2e-3), n
    # one SGD step: three groups, momentum 0.9, nesterov, wd 1e-4, lr 1e-3; frozen cls.0
    groups = []
    for pre in ("body.", "head.", "cls."):
        ps = [v for k, v in Ps.items() if k.startswith(pre) and v.requires_grad and not k.startswith("cls.0.")]
        groups.append({"params": ps, "weight_decay": 1e-4})
    opt = torch.optim.SGD(groups, lr=1e-3, momentum=0.9, nesterov=True)
    opt.step()
    for n in names:
        np.testing.assert_allclose(Ps[n].detach().flatten()[:16].numpy(), g[f"after_step::{n}"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(Ps["body.mod1.bn1.running_mean"].numpy(), g["running_mean_after"], rtol=1e-5, atol=1e-6)


def test_config0_voc_19_1_step0_ft_matches_reference():
    """BASELINE.json configs[0]: VOC 19-1 step 0, --method FT (plain cross entropy, no teacher), 2 synthetic 256x256
    images on the CPU - the reference's own CPU-runnable case (golden: tests/golden/make_goldens.py::gold_cfg0)."""
    from oracle.params import template_state
    g = load_golden("cfg0_step.npz")
    P = OS.make_params(synth.fill_state_dict(template_state([20]), 43))
    P["cls.0.weight"].requires_grad_(False); P["cls.0.bias"].requires_grad_(False)      # segmentation_module.py:75-78
    img = synth.images(777, 2, 256)
    labels = synth.seg_labels(777, 2, 256, 256, range(1, 20))
    out, feat = OM.segmentation_forward(img, P, 1, training=True)
    loss = torch.nn.functional.cross_entropy(out, labels, ignore_index=255, reduction="none").mean()     # train.py:30,116
    assert loss.item() == pytest.approx(float(g["loss"]), rel=1e-4)
    assert out.double().abs().sum().item() == pytest.approx(float(g["logits_abs"]), rel=1e-4)
    np.testing.assert_allclose(feat["sem"].detach().numpy()[:, :, ::4, ::4], g["sem"], rtol=1e-3, atol=1e-3)
    loss.backward()
    for k in g:
        if k.startswith("grad_abs::"):
            n = k.split("::")[1]
            assert P[n].grad.double().abs().sum().item() == pytest.approx(float(g[k]), rel=2e-3), n


def _check_step_golden(g, r, Ps, rel=1e-4):
    from conftest import assert_matches_compact
    assert r["A"] == int(g["A"]) and r["C"] == int(g["C"])
    for k in ("ce", "con", "loss", "lkd"):
        assert r[k].item() == pytest.approx(float(g[k]), rel=rel), k
    np.testing.assert_allclose(r["logits"].detach().flatten()[g["sample_idx"]].numpy(), g["logits_sample"], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(r["logits_old"].flatten()[sample_idx(r["logits_old"].numel(), 256)].numpy(),
                               g["teacher_logits_sample"], rtol=1e-3, atol=1e-3)
    (r["loss"] + r["lkd"]).backward()
    for k in g:
        if k.startswith("grad_abs::"):
            n = k.split("::")[1]
            assert Ps[n].grad.double().abs().sum().item() == pytest.approx(float(g[k]), rel=2e-3), n


def test_ucd_step_at_the_benchmark_crop_matches_reference():
    """configs[1] at 513^2 (2 images): 33 x 33 stride-16 maps, so the teacher's ASPP takes the sliding-window pooling branch
    (modules/deeplab.py:77-88) that the 129^2 golden never reaches."""
    g = load_golden("ucd_step_513.npz")
    Ps, Pt = _student_teacher_params()
    img = synth.images(502, 2, 513)
    labels = synth.seg_labels(502, 2, 513, 513, range(16, 21))
    r = OS.ucd_losses(Ps, Pt, img, labels, [16, 5])
    _check_step_golden(g, r, Ps)


def test_ucd_step_multi_head_15_5s_step3_matches_reference():
    """configs[2]: a later overlapped step - student heads [16, 1, 1, 1], teacher [16, 1, 1]."""
    g = load_golden("ucd_step_15_5s_step3.npz")
    Ps, Pt = _student_teacher_params(seed=44, classes=(16, 1, 1, 1))
    img = synth.images(503, 2, 129)
    labels = synth.seg_labels(503, 2, 129, 129, [18])
    r = OS.ucd_losses(Ps, Pt, img, labels, [16, 1, 1, 1])
    _check_step_golden(g, r, Ps)


@pytest.mark.parametrize("gname,classes,new_ids,max_label", [
    ("ucd_step_513_cal.npz", (16, 5), range(16, 21), 20),            # configs[1], calibrated checkpoint (the bf16 tests' golden)
    ("ucd_step_ade_512.npz", (101, 50), range(101, 151), 150),       # configs[3] per-rank shape: 3 x 512^2, 151 classes, K = 101
    ("ucd_step_city_768.npz", (14, 6), range(14, 20), 20)])          # configs[4] per-rank shape: 2 x 768^2, 48 x 48 maps
def test_whole_steps_of_the_other_configs_match_reference(gname, classes, new_ids, max_label):
    """The oracle's whole UCD step against goldens captured through the reference's classes at the per-rank shapes of the 8-GPU
    configurations, from the calibrated synthetic checkpoint (tests/golden/make_goldens.py::gold_step513_cal / gold_cfg3_ade /
    gold_cfg4_city)."""
    g = load_golden(gname)
    seed, B, S = [int(v) for v in g["cfg"]]
    Ps, Pt = _student_teacher_params(classes=classes, calibrated=True)
    img = synth.images(seed, B, S)
    labels = synth.seg_labels(seed, B, S, S, new_ids)
    r = OS.ucd_losses(Ps, Pt, img, labels, list(classes), max_label=max_label)
    _check_step_golden(g, r, Ps)


def test_aspp_eval_pooling_matches_reference():
    """DeeplabV3 eval mode on maps larger than the pooling window (33 x 33, 48 x 48, odd map with an even window)."""
    from functools import partial
    from conftest import assert_matches_compact
    g = load_golden("aspp_eval.npz")
    norm = partial(ShimABN, activation="leaky_relu", activation_param=0.01)
    for tag in ("33", "48", "odd"):
        seed, B, C, H, W, pool = [int(v) for v in g[f"cfg_{tag}"]]
        head = DeeplabV3(C, 32, 16, norm_act=norm, out_stride=16, pooling_size=pool)
        Ph = {"head." + k: v for k, v in synth.fill_state_dict(head.state_dict(), 21).items()}
        x = synth.t_normal(seed, (B, C, H, W), stream=1)
        y = OM.deeplab_head(x, Ph, False, pooling_size=pool)
        assert_matches_compact(g, f"eval_{tag}", y.numpy(), rtol=1e-4, atol=1e-4)
        # the product's module tree on the CPU shim (literal fallback path of ucd_amd.blocks) agrees too
        head.load_state_dict(synth.fill_state_dict(head.state_dict(), 21))
        head.eval()
        with torch.no_grad():
            assert_matches_compact(g, f"eval_{tag}", head(x).numpy(), rtol=1e-4, atol=1e-4)
