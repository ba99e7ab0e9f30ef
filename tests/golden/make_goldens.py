"""Regenerates tests/golden/*.npz by running the REFERENCE's own Python (imported from
/root/reference, which exists only in the build container) on closed-form synthetic inputs
(ucd_amd.synth).  Only numeric inputs/outputs are stored; no reference source travels.

    python tests/golden/make_goldens.py            # writes the .npz files next to this script

Import recipe: utils/loss.py and utils/loss_new.py load by file path with
torch only; models/ + modules/ import with sys.path=/root/reference (models first); the third-party
wheels the reference needs but this image lacks (inplace_abn, torchvision, apex, wandb, cv2) are
replaced by import-time stand-ins *in this process only* - inplace_abn.ABN becomes
BatchNorm2d + leaky_relu, inplace_abn.InPlaceABN / InPlaceABNSync the same with the wheel's published
``|weight| + eps`` scale (the model-level goldens use the latter: --norm_act iabn_sync is the reference's default).
"""
import importlib.util
import os
import sys
import types
from functools import partial

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
from ucd_amd import synth  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)


def load_by_path(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


ref_loss = load_by_path("ref_loss", "utils/loss.py")
ref_loss_new = load_by_path("ref_loss_new", "utils/loss_new.py")


# ---------------------------------------------------------------------------------------------
class ShimABN(nn.BatchNorm2d):
    """inplace_abn.ABN stand-in: BatchNorm2d (eps 1e-5, momentum 0.1) + activation."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, activation="leaky_relu",
                 activation_param=0.01):
        super().__init__(num_features, eps=eps, momentum=momentum, affine=affine)
        self.activation, self.activation_param = activation, activation_param

    def forward(self, x):
        y = super().forward(x)
        if self.activation == "leaky_relu":
            return F.leaky_relu(y, self.activation_param)
        return y


class ShimInPlaceABN(ShimABN):
    """inplace_abn.InPlaceABN / InPlaceABNSync stand-in: the in-place kernels of the wheel normalise with
    ``|weight| + eps`` (its published forward), everything else as ShimABN.  This is what ``--norm_act iabn_sync`` (the
    reference's default, argparser.py:132) instantiates, so the model-level goldens are captured with it."""

    def forward(self, x):
        y = F.batch_norm(x, self.running_mean, self.running_var, self.weight.abs() + self.eps, self.bias,
                         self.training, self.momentum, self.eps)
        if self.activation == "leaky_relu":
            return F.leaky_relu(y, self.activation_param)
        return y


def import_reference_model():
    shim = types.ModuleType("inplace_abn")
    shim.ABN = ShimABN
    shim.InPlaceABN = shim.InPlaceABNSync = ShimInPlaceABN
    sys.modules["inplace_abn"] = shim
    tv = types.ModuleType("torchvision"); tvt = types.ModuleType("torchvision.transforms")
    tvf = types.ModuleType("torchvision.transforms.functional"); tvf.normalize = lambda *a, **k: None
    tv.transforms = tvt; tvt.functional = tvf
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt,
                        "torchvision.transforms.functional": tvf})
    for name in ("tensorboardX", "wandb", "cv2"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.path.insert(0, REF)
    import models  # noqa: F401  (must precede modules: circular import)
    import modules  # noqa: F401
    import segmentation_module
    return models, modules, segmentation_module


OUT_DIR = os.environ.get("UCD_GOLDEN_OUT", HERE)      # the regeneration test writes into a scratch directory


def save(name, **arrays):
    path = os.path.join(OUT_DIR, name)
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def sample_idx(n, k=64, seed=5):
    return synth.randint(seed, (min(k, n),), 0, n, stream=77)


def compact(prefix, arr, full_below=6000, k=512):
    """Small arrays are stored whole; big ones as (sum, abs-sum, row sums over the last axis, k samples at
    closed-form indices sample_idx(size, k)) so the fixtures stay KB-sized."""
    arr = np.asarray(arr)
    if arr.size <= full_below:
        return {prefix: arr}
    flat = arr.reshape(-1)
    return {prefix + "::sum": flat.astype(np.float64).sum(), prefix + "::abs": np.abs(flat.astype(np.float64)).sum(),
            prefix + "::rowsum": arr.reshape(-1, arr.shape[-1]).astype(np.float64).sum(axis=1).astype(np.float32)
            if arr.ndim > 1 and arr.size // arr.shape[-1] <= 4096 else np.zeros(0, np.float32),
            prefix + "::shape": np.array(arr.shape), prefix + "::samples": flat[sample_idx(flat.size, k)]}


# ---------------------------------------------------------------------------------------------
PIXCON_CASES = {
    # name: (seed, B, N, h, w, K, H, W, new_ids)
    "voc_15_5": (101, 2, 32, 9, 9, 16, 129, 129, list(range(16, 21))),
    "city_13_6": (102, 3, 64, 12, 12, 14, 190, 190, list(range(14, 20))),
    "voc_15_5s_step2": (103, 2, 48, 8, 8, 17, 128, 128, [17]),
    "voc_19_1_odd": (104, 4, 16, 7, 11, 20, 100, 171, [20]),
}


def gold_pixcon():
    for name, (seed, B, N, h, w, K, H, W, new_ids) in PIXCON_CASES.items():
        f_n, f_o, l_po, labels = synth.contrastive_case(seed, B, N, h, w, K, H, W, new_ids)
        f_n = f_n.clone().requires_grad_(True)
        a, c, la, lc, P = ref_loss.pre_contrastive_pixel(f_n, labels.clone(), l_po=l_po, f_o=f_o)
        crit = ref_loss.PixelConLossV2(temperature=0.07)
        loss = crit(a, c, la, lc, P)
        loss.backward()
        grad = f_n.grad.detach().clone()
        loss_nop = crit(a.detach(), c, la, lc, None)
        label_ds = F.interpolate(labels.float().unsqueeze(1), size=(h, w), mode="bilinear",
                                 align_corners=False)[:, 0]
        save(f"pixcon_{name}.npz", cfg=np.array([seed, B, N, h, w, K, H, W] + new_ids),
             A=a.shape[0], C=c.shape[0], la=la.numpy().astype(np.int16), lc=lc.numpy().astype(np.int16),
             loss=loss.item(), loss_noP=loss_nop.item(), label_interp=label_ds.numpy(),
             **compact("a", a.detach().numpy()), **compact("c", c.numpy()), **compact("P", P.numpy()),
             **compact("grad_f_n", grad.numpy()))


def gold_logit_losses():
    out = {}
    for tag, (Ctot, K) in {"voc": (21, 16), "city": (20, 14), "ade": (151, 101)}.items():
        seed = 200 + Ctot
        x = synth.t_normal(seed, (2, Ctot, 16, 16), stream=1, scale=2.0).requires_grad_(True)
        t = synth.t_normal(seed, (2, K, 16, 16), stream=2, scale=2.0)
        lab = synth.randint(seed, (2, 16, 16), 0, Ctot + 3, stream=3)
        lab = np.where(lab >= Ctot, 255, lab)
        lab = torch.from_numpy(lab)
        ce = ref_loss.UnbiasedCrossEntropy(old_cl=K, ignore_index=255, reduction="none")(x, lab.clone())
        ce_mean = ce.mean()
        g_ce, = torch.autograd.grad(ce_mean, x)
        kd = ref_loss.UnbiasedKnowledgeDistillationLoss(alpha=1.0)(x, t)
        g_kd, = torch.autograd.grad(kd, x)
        out.update({f"{tag}_cfg": np.array([seed, Ctot, K]), f"{tag}_ce": ce.detach().numpy(),
                    f"{tag}_ce_mean": ce_mean.item(), f"{tag}_kd": kd.item(),
                    **compact(f"{tag}_g_ce", g_ce.numpy()), **compact(f"{tag}_g_kd", g_kd.numpy())})
    save("logit_losses.npz", **out)


def gold_v1_losses():
    seed = 300
    n, d = 96, 32
    f = F.normalize(synth.t_normal(seed, (n, d), stream=1), dim=1)
    lab = torch.from_numpy(synth.randint(seed, (n,), 0, 5, stream=2))
    pix = ref_loss_new.PixelConLoss(temperature=0.07)(f[:, None, :], lab)
    pix_t1 = ref_loss_new.PixelConLoss()(f[:, None, :], lab)
    f2 = F.normalize(synth.t_normal(seed, (n, 2, d), stream=3), dim=2)
    sup = ref_loss_new.SupConLoss(temperature=0.07)(f2, lab)
    sup_one = ref_loss_new.SupConLoss(temperature=0.1, contrast_mode="one")(f2, lab)
    simclr = ref_loss_new.SupConLoss()(f2)
    save("v1_losses.npz", cfg=np.array([seed, n, d]), pixcon_T007=pix.item(), pixcon_T1=pix_t1.item(),
         supcon=sup.item(), supcon_one=sup_one.item(), simclr=simclr.item())


# ---------------------------------------------------------------------------------------------
def gold_model():
    models, modules, segm = import_reference_model()
    norm = partial(ShimInPlaceABN, activation="leaky_relu", activation_param=0.01)

    # (a) one projection bottleneck + one identity bottleneck, train and eval
    blk = modules.ResidualBlock(32, (16, 16, 64), norm_act=norm, stride=2, dilation=1)
    blk2 = modules.ResidualBlock(64, (16, 16, 64), norm_act=norm, stride=1, dilation=2)
    blk.load_state_dict(synth.fill_state_dict(blk.state_dict(), 11))
    blk2.load_state_dict(synth.fill_state_dict(blk2.state_dict(), 12))
    x = synth.t_normal(400, (3, 32, 13, 13), stream=1)
    out = {}
    for mode in ("train", "eval"):
        getattr(blk, mode)(); getattr(blk2, mode)()
        y = blk2(blk(x.clone()))
        out[f"block_{mode}"] = y.detach().numpy()
    out["block_rm_after"] = blk.convs.bn1.running_mean.numpy().copy()

    # (b) DeepLab-V3 head on a small map, train and eval (pooling 4 < map size: sliding window)
    head = modules.DeeplabV3(48, 24, 16, norm_act=norm, out_stride=16, pooling_size=4)
    head.load_state_dict(synth.fill_state_dict(head.state_dict(), 13))
    xh = synth.t_normal(401, (2, 48, 7, 9), stream=1)
    head.train(); out["head_train"] = head(xh.clone()).detach().numpy()
    head.eval(); out["head_eval"] = head(xh.clone()).detach().numpy()
    save("model_blocks.npz", **out)

    # (c) full ResNet-101 + DeepLab-V3 + incremental heads [16, 5], 2x3x65x65
    def build(classes):
        body = models.net_resnet101(norm_act=norm, output_stride=16)
        head = modules.DeeplabV3(body.out_channels, 256, 256, norm_act=norm, out_stride=16, pooling_size=32)
        return segm.IncrementalSegmentationModule(body, head, 256, classes=classes)

    student, teacher = build([16, 5]), build([16])
    sd = synth.fill_state_dict(teacher.state_dict(), 42)
    teacher.load_state_dict(sd)
    student.load_state_dict(sd, strict=False)
    with torch.no_grad():
        student.init_new_classifier(torch.device("cpu"))
    teacher.eval()
    img = synth.images(500, 2, 65)
    with torch.no_grad():
        lt, ft = teacher(img)
    student.eval()
    with torch.no_grad():
        ls_eval, fs_eval = student(img)
    student.train()
    ls, fs = student(img)
    idx = sample_idx(ls.numel(), 256)
    save("model_full.npz", img_cfg=np.array([500, 2, 65]),
         teacher_logits_sum=lt.double().sum().item(), teacher_logits_abs=lt.double().abs().sum().item(),
         teacher_sem=ft["sem"].numpy(), teacher_pl_abs=ft["pre_logits"].double().abs().sum().item(),
         teacher_body_abs=ft["body"].double().abs().sum().item(),
         student_eval_sem=fs_eval["sem"].numpy(),
         student_train_sem=fs["sem"].detach().numpy(),
         student_train_logits_sample=ls.detach().flatten()[idx].numpy(), sample_idx=idx,
         student_train_pl_abs=fs["pre_logits"].double().abs().sum().item(),
         new_head_bias=student.cls[1].bias.detach().numpy(), head0_bias0=student.cls[0].bias[0].item())

    # (d) the composed UCD step on 2x3x129x129 (train.py:95-151 as intended)
    student, teacher = build([16, 5]), build([16])
    teacher.load_state_dict(sd)
    student.load_state_dict(sd, strict=False)
    with torch.no_grad():
        student.init_new_classifier(torch.device("cpu"))
    for p in teacher.parameters():
        p.requires_grad = False
    teacher.eval(); student.train()
    img = synth.images(501, 2, 129)
    labels = synth.seg_labels(501, 2, 129, 129, range(16, 21))
    groups = [{"params": [p for p in m.parameters() if p.requires_grad], "weight_decay": 1e-4}
              for m in (student.body, student.head, student.cls)]
    opt = torch.optim.SGD(groups, lr=1e-3, momentum=0.9, nesterov=True)
    with torch.no_grad():
        out_old, feat_old = teacher(img)
    opt.zero_grad()
    outp, feat = student(img)
    a, c, la, lc, P = ref_loss.pre_contrastive_pixel(feat["pre_logits"], labels.clone(), l_po=feat_old["sem"],
                                                     f_o=feat_old["pre_logits"])
    ce = ref_loss.UnbiasedCrossEntropy(old_cl=16, ignore_index=255, reduction="none")(outp, labels.clone()).mean()
    con = ref_loss.PixelConLossV2(temperature=0.07)(a, c, la, lc, P)
    loss = ce + con / 100
    lkd = 10 * ref_loss.UnbiasedKnowledgeDistillationLoss(alpha=1.0)(outp, out_old)
    (loss + lkd).backward()
    names = ["body.mod1.conv1.weight", "body.mod3.block2.convs.bn2.weight", "body.mod5.block3.convs.conv3.weight",
             "head.map_convs.2.weight", "head.red_bn.bias", "head.global_pooling_bn.weight", "cls.1.weight",
             "cls.1.bias"]
    params = dict(student.named_parameters())
    grads = {f"grad_abs::{n}": params[n].grad.double().abs().sum().item() for n in names}
    grads.update({f"grad_sum::{n}": params[n].grad.double().sum().item() for n in names})
    opt.step()
    upd = {f"after_step::{n}": params[n].detach().flatten()[:16].numpy().copy() for n in names}
    save("ucd_step.npz", cfg=np.array([501, 2, 129]), ce=ce.item(), con=con.item(), loss=loss.item(),
         lkd=lkd.item(), A=a.shape[0], C=c.shape[0], **grads, **upd,
         running_mean_after=student.body.mod1.bn1.running_mean.numpy().copy())


def gold_cfg0():
    """BASELINE.json configs[0]: VOC 19-1 step 0, --method FT (plain cross entropy, no teacher), 2 synthetic 256x256
    images, single-process CPU fp32, through the reference's own model classes (train.py:108,116 with model_old None;
    run.py:175-186 optimiser with the step-0 learning rate 0.01)."""
    models, modules, segm = import_reference_model()
    norm = partial(ShimInPlaceABN, activation="leaky_relu", activation_param=0.01)
    body = models.net_resnet101(norm_act=norm, output_stride=16)
    head = modules.DeeplabV3(body.out_channels, 256, 256, norm_act=norm, out_stride=16, pooling_size=32)
    model = segm.IncrementalSegmentationModule(body, head, 256, classes=[20])
    model.load_state_dict(synth.fill_state_dict(model.state_dict(), 43))
    model.train()
    img = synth.images(777, 2, 256)
    labels = synth.seg_labels(777, 2, 256, 256, range(1, 20))
    groups = [{"params": [p for p in m.parameters() if p.requires_grad], "weight_decay": 1e-4}
              for m in (model.body, model.head, model.cls)]
    opt = torch.optim.SGD(groups, lr=1e-2, momentum=0.9, nesterov=True)
    opt.zero_grad()
    out, feat = model(img)
    loss = torch.nn.CrossEntropyLoss(ignore_index=255, reduction="none")(out, labels).mean()
    loss.backward()
    names = ["body.mod1.conv1.weight", "body.mod4.block7.convs.bn2.weight", "head.red_conv.weight", "head.map_bn.bias", "body.mod5.block1.proj_conv.weight"]
    params = dict(model.named_parameters())
    grads = {f"grad_abs::{n}": params[n].grad.double().abs().sum().item() for n in names}
    opt.step()
    upd = {f"after_step::{n}": params[n].detach().flatten()[:16].numpy().copy() for n in names}
    idx = sample_idx(out.numel(), 256)
    save("cfg0_step.npz", cfg=np.array([777, 2, 256]), loss=loss.item(), logits_abs=out.double().abs().sum().item(),
         logits_sample=out.detach().flatten()[idx].numpy(), sample_idx=idx, sem=feat["sem"].detach().numpy()[:, :, ::4, ::4].copy(),
         **grads, **upd)


def _build_pair(models, modules, segm, classes, seed=42, calibrated=False):
    norm = partial(ShimInPlaceABN, activation="leaky_relu", activation_param=0.01)

    def build(cls):
        body = models.net_resnet101(norm_act=norm, output_stride=16)
        head = modules.DeeplabV3(body.out_channels, 256, 256, norm_act=norm, out_stride=16, pooling_size=32)
        return segm.IncrementalSegmentationModule(body, head, 256, classes=cls)

    student, teacher = build(classes), build(classes[:-1])
    sd = synth.fill_state_dict(teacher.state_dict(), seed, calibrated=calibrated)
    teacher.load_state_dict(sd)
    student.load_state_dict(sd, strict=False)
    with torch.no_grad():
        student.init_new_classifier(torch.device("cpu"))
    for p in teacher.parameters():
        p.requires_grad = False
    teacher.eval(); student.train()
    return student, teacher


STEP_NAMES = ["body.mod1.conv1.weight", "body.mod3.block2.convs.bn2.weight", "body.mod5.block3.convs.conv3.weight",
              "head.map_convs.2.weight", "head.red_bn.bias", "head.global_pooling_bn.weight"]


def _ucd_step(student, teacher, img, labels, old_cl, extra_names, max_label=None):
    """train.py:95-151 as intended, through the reference's own classes; returns the golden dict.
    ``max_label``: ADE20K only - the reference's pre_contrastive_pixel zeroes every down-sampled label above the VOC bound 20
    (utils/utils.py:267-268), so with ADE's new ids 101..150 it has no new pixel and stops at ``min`` of an empty tensor; the
    prep then comes from oracle/contrastive.py (pinned to the reference's function on the four pixcon_*.npz cases) with the
    bound generalised, everything else stays the reference's own classes."""
    groups = [{"params": [p for p in m.parameters() if p.requires_grad], "weight_decay": 1e-4}
              for m in (student.body, student.head, student.cls)]
    opt = torch.optim.SGD(groups, lr=1e-3, momentum=0.9, nesterov=True)
    with torch.no_grad():
        out_old, feat_old = teacher(img)
    opt.zero_grad()
    outp, feat = student(img)
    if max_label is None:
        a, c, la, lc, P = ref_loss.pre_contrastive_pixel(feat["pre_logits"], labels.clone(), l_po=feat_old["sem"],
                                                         f_o=feat_old["pre_logits"])
    else:
        from oracle import contrastive as OC
        prep = OC.pre_contrastive_pixel(feat["pre_logits"], labels.clone(), feat_old["sem"], feat_old["pre_logits"],
                                        max_label=max_label)
        a, c, la, lc, P = prep["a"], prep["c"], prep["la"], prep["lc"], prep["P"]
    ce = ref_loss.UnbiasedCrossEntropy(old_cl=old_cl, ignore_index=255, reduction="none")(outp, labels.clone()).mean()
    con = ref_loss.PixelConLossV2(temperature=0.07)(a, c, la, lc, P)
    loss = ce + con / 100
    lkd = 10 * ref_loss.UnbiasedKnowledgeDistillationLoss(alpha=1.0)(outp, out_old)
    (loss + lkd).backward()
    names = STEP_NAMES + extra_names
    params = dict(student.named_parameters())
    grads = {f"grad_abs::{n}": params[n].grad.double().abs().sum().item() for n in names}
    grads.update({f"grad_sum::{n}": params[n].grad.double().sum().item() for n in names})
    opt.step()
    upd = {f"after_step::{n}": params[n].detach().flatten()[:16].numpy().copy() for n in names}
    idx = sample_idx(outp.numel(), 256)
    out = dict(ce=ce.item(), con=con.item(), loss=loss.item(), lkd=lkd.item(), A=a.shape[0], C=c.shape[0],
               logits_sample=outp.detach().flatten()[idx].numpy(), sample_idx=idx,
               teacher_logits_sample=out_old.flatten()[sample_idx(out_old.numel(), 256)].numpy(),
               running_mean_after=student.body.mod1.bn1.running_mean.numpy().copy(), **grads, **upd)
    out.update(compact("teacher_sem", feat_old["sem"].numpy()))
    out.update(compact("student_sem", feat["sem"].detach().numpy()))
    return out


def gold_step513():
    """BASELINE.json configs[1] at its REAL crop: VOC 15-5 step 1 --method UCD on 2 x 513^2 (the per-image shapes of the
    benchmark; at 513 the stride-16 map is 33 x 33 > --pooling 32, so the teacher takes the sliding-window branch of
    modules/deeplab.py:77-88 that smaller crops never reach)."""
    models, modules, segm = import_reference_model()
    student, teacher = _build_pair(models, modules, segm, [16, 5])
    img = synth.images(502, 2, 513)
    labels = synth.seg_labels(502, 2, 513, 513, range(16, 21))
    out = _ucd_step(student, teacher, img, labels, 16, ["cls.1.weight", "cls.1.bias"])
    save("ucd_step_513.npz", cfg=np.array([502, 2, 513]), **out)


def gold_step513_cal():
    """configs[1] at its real crop with the CALIBRATED synthetic checkpoint (synth.fill_state_dict(calibrated=True): teacher
    logits of order 10 instead of 1e5) - the golden the bf16 (--opt_level O1, the benchmarked mode) tests are held against."""
    models, modules, segm = import_reference_model()
    student, teacher = _build_pair(models, modules, segm, [16, 5], calibrated=True)
    img = synth.images(502, 2, 513)
    labels = synth.seg_labels(502, 2, 513, 513, range(16, 21))
    out = _ucd_step(student, teacher, img, labels, 16, ["cls.1.weight", "cls.1.bias"])
    save("ucd_step_513_cal.npz", cfg=np.array([502, 2, 513]), **out)


def gold_cfg3_ade():
    """BASELINE.json configs[3] at its PER-RANK shape: ADE20K 100-50 step 1, 3 x 512^2 (24 images on 8 GPUs), 151-class student,
    K = 101 teacher classes, new ids 101..150; calibrated checkpoint.  See _ucd_step for the one piece the reference cannot run."""
    models, modules, segm = import_reference_model()
    student, teacher = _build_pair(models, modules, segm, [101, 50], calibrated=True)
    img = synth.images(504, 3, 512)
    labels = synth.seg_labels(504, 3, 512, 512, range(101, 151))
    out = _ucd_step(student, teacher, img, labels, 101, ["cls.1.weight", "cls.1.bias"], max_label=150)
    save("ucd_step_ade_512.npz", cfg=np.array([504, 3, 512]), **out)


def gold_cfg4_city():
    """BASELINE.json configs[4] at its PER-RANK shape: Cityscapes 13-6 step 1, 2 x 768^2 (16 images on 8 GPUs), 20-class student,
    K = 14, new ids 14..19 (within the reference's hard-coded label bound: its own pre_contrastive_pixel runs); the 48 x 48 map
    exceeds --pooling 32, so the teacher's image pooling slides; calibrated checkpoint."""
    models, modules, segm = import_reference_model()
    student, teacher = _build_pair(models, modules, segm, [14, 6], calibrated=True)
    img = synth.images(505, 2, 768)
    labels = synth.seg_labels(505, 2, 768, 768, range(14, 20))
    out = _ucd_step(student, teacher, img, labels, 14, ["cls.1.weight", "cls.1.bias"])
    save("ucd_step_city_768.npz", cfg=np.array([505, 2, 768]), **out)


def gold_heads():
    """BASELINE.json configs[2], a later overlapped step: VOC 15-5s step 3 -> student heads [16, 1, 1, 1], teacher
    [16, 1, 1] (tasks.py:16-29), new class id 18, 2 x 129^2."""
    models, modules, segm = import_reference_model()
    student, teacher = _build_pair(models, modules, segm, [16, 1, 1, 1], seed=44)
    img = synth.images(503, 2, 129)
    labels = synth.seg_labels(503, 2, 129, 129, [18])
    out = _ucd_step(student, teacher, img, labels, 18, ["cls.1.weight", "cls.2.bias", "cls.3.weight", "cls.3.bias"])
    save("ucd_step_15_5s_step3.npz", cfg=np.array([503, 2, 129]), **out)


def gold_aspp_eval():
    """DeeplabV3 in evaluation mode on maps LARGER than --pooling 32 (modules/deeplab.py:77-88: avg_pool2d(32, stride 1) +
    replicate pad): 33 x 33 (VOC 513^2) and 48 x 48 (Cityscapes 768^2), plus an even pooling size on an odd map."""
    models, modules, segm = import_reference_model()
    norm = partial(ShimInPlaceABN, activation="leaky_relu", activation_param=0.01)
    out = {}
    for tag, hw, pool in (("33", (33, 33), 32), ("48", (48, 48), 32), ("odd", (21, 35), 8)):
        head = modules.DeeplabV3(64, 32, 16, norm_act=norm, out_stride=16, pooling_size=pool)
        head.load_state_dict(synth.fill_state_dict(head.state_dict(), 21))
        head.eval()
        x = synth.t_normal(410 + len(tag), (2, 64) + hw, stream=1)
        with torch.no_grad():
            y = head(x)
        out.update(compact(f"eval_{tag}", y.numpy()))
        out[f"cfg_{tag}"] = np.array([410 + len(tag), 2, 64, hw[0], hw[1], pool])
    save("aspp_eval.npz", **out)


TRAJ_UPDATE_NAMES = ("body.mod1.conv1.weight", "body.mod2.block1.convs.conv1.weight", "body.mod3.block2.convs.conv2.weight",
                     "body.mod4.block10.convs.conv3.weight", "body.mod5.block3.convs.conv3.weight", "head.map_convs.2.weight",
                     "head.red_conv.weight", "cls.1.weight")
TRAJ_SAMPLES = 512


def traj_sample_index(numel):
    """the TRAJ_SAMPLES evenly spaced flat indices at which a parameter's accumulated update is stored"""
    return np.linspace(0, numel - 1, TRAJ_SAMPLES).astype(np.int64)


def gold_traj513(steps=20, name="ucd_traj_513_cal.npz"):
    """``steps`` consecutive iterations of train.py:95-151 (as intended) on ONE fixed batch - configs[1] at its real crop, 2 x 513^2,
    calibrated checkpoint, SGD-Nesterov lr 1e-3 / wd 1e-4 in the reference's three groups, no scheduler - through the reference's
    own classes in fp32: the per-step losses are the golden the product's multi-step tests are held to (fp32 mode tightly over
    the first steps, the benchmarked bf16 mode along the whole trajectory)."""
    models, modules, segm = import_reference_model()
    student, teacher = _build_pair(models, modules, segm, [16, 5], calibrated=True)
    img = synth.images(502, 2, 513)
    labels = synth.seg_labels(502, 2, 513, 513, range(16, 21))
    groups = [{"params": [p for p in m.parameters() if p.requires_grad], "weight_decay": 1e-4}
              for m in (student.body, student.head, student.cls)]
    opt = torch.optim.SGD(groups, lr=1e-3, momentum=0.9, nesterov=True)
    with torch.no_grad():
        out_old, feat_old = teacher(img)
    rec = {k: [] for k in ("ce", "con", "lkd", "A", "C")}
    # the reference's accumulated UPDATE of eight parameters across the network: sampled elements + the full length,
    # after 2 steps (what the regeneration test re-runs) and after all of them - the product's update DIRECTION is held to these
    params = dict(student.named_parameters())
    before = {n: params[n].detach().double().clone() for n in TRAJ_UPDATE_NAMES}

    def updates():
        d = [(params[n].detach().double() - before[n]).flatten() for n in TRAJ_UPDATE_NAMES]
        return (np.stack([v[torch.from_numpy(traj_sample_index(v.numel()))].numpy() for v in d]),
                np.array([float(v.norm()) for v in d]))
    upd2 = None
    for it in range(steps):
        opt.zero_grad()
        outp, feat = student(img)
        a, c, la, lc, P = ref_loss.pre_contrastive_pixel(feat["pre_logits"], labels.clone(), l_po=feat_old["sem"],
                                                         f_o=feat_old["pre_logits"])
        ce = ref_loss.UnbiasedCrossEntropy(old_cl=16, ignore_index=255, reduction="none")(outp, labels.clone()).mean()
        con = ref_loss.PixelConLossV2(temperature=0.07)(a, c, la, lc, P)
        lkd = 10 * ref_loss.UnbiasedKnowledgeDistillationLoss(alpha=1.0)(outp, out_old)
        (ce + con / 100 + lkd).backward()
        opt.step()
        for k, v in (("ce", ce.item()), ("con", con.item()), ("lkd", lkd.item()), ("A", a.shape[0]), ("C", c.shape[0])):
            rec[k].append(v)
        print(f"traj step {it}: ce {ce.item():.6f} con {con.item():.6f} lkd {lkd.item():.6f} A {a.shape[0]} C {c.shape[0]}", flush=True)
        if it == 1:
            upd2 = updates()
    upd = updates()
    save(name, cfg=np.array([502, 2, 513, steps]), ce=np.array(rec["ce"]), con=np.array(rec["con"]),
         lkd=np.array(rec["lkd"]), A=np.array(rec["A"]), C=np.array(rec["C"]),
         running_mean_after=student.body.mod1.bn1.running_mean.numpy().copy(),
         cls1_bias_after=dict(student.named_parameters())["cls.1.bias"].detach().numpy().copy(),
         upd2=upd2[0], upd2_norm=upd2[1], upd=upd[0], upd_norm=upd[1])


if __name__ == "__main__":
    which = sys.argv[1:] or ["pixcon", "logit", "v1", "model", "cfg0", "step513", "heads", "aspp_eval", "step513_cal", "cfg3_ade",
                             "cfg4_city", "traj513"]
    if "traj513" in which:
        gold_traj513()
    if "traj513_head" in which:          # the first two iterations only (the regeneration test: equal to the committed file's prefix)
        gold_traj513(2, "ucd_traj_513_cal_head.npz")
    if "cfg0" in which:
        gold_cfg0()
    if "pixcon" in which:
        gold_pixcon()
    if "logit" in which:
        gold_logit_losses()
    if "v1" in which:
        gold_v1_losses()
    if "model" in which:
        gold_model()
    if "step513" in which:
        gold_step513()
    if "heads" in which:
        gold_heads()
    if "step513_cal" in which:
        gold_step513_cal()
    if "cfg3_ade" in which:
        gold_cfg3_ade()
    if "cfg4_city" in which:
        gold_cfg4_city()
    if "aspp_eval" in which:
        gold_aspp_eval()
