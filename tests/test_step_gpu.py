"""GPU: the product's full UCD step (HIP ABN + fused contrastive + fused logit losses + MIOpen convs) in
fp32 against the golden captured from the reference's own classes.
Bar: losses and logits within 1e-3 relative (north_star).  Parameter GRADIENTS of the composed 100-layer
network are compared statistically (abs-sum within 10 %, post-step parameters within 1e-3): the
forward activations of two fp32 implementations (CPU oneDNN vs GPU MIOpen + HIP) differ by ~3e-4
relative at the head (measured, tests/diag/layer_diag.py), which flips the sign of ~0.1 % of the leaky-ReLU
pre-activations per layer; each flip changes that element's gradient by 99 %, so activation gradients
differ by sqrt(p) ~ 3 % per layer although every individual kernel matches its reference to 1e-5 on
identical inputs (tests/test_abn_gpu.py, tests/test_pixcon_gpu.py, tests/test_seglosses_gpu.py)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from ucd_amd import argparser, synth, tasks

pytestmark = pytest.mark.gpu


def _opts(extra=()):
    o = argparser.get_argparser().parse_args(["--method", "UCD", "--task", "15-5", "--step", "1", "--lr", "0.001",
                                              "--no_pretrained", "--norm_act", "iabn_sync", *extra])
    return argparser.modify_command_options(o)


def _build(opts, dev):
    from ucd_amd.run import build_models, load_step_checkpoint, make_optimizer
    classes = tasks.get_per_task_classes("voc", "15-5", 1)
    torch.backends.cudnn.allow_tf32 = False
    model, model_old = build_models(opts, dev, classes)
    state = synth.fill_state_dict({k: v.cpu() for k, v in model_old.state_dict().items()}, 42)
    load_step_checkpoint(opts, model, model_old, state, dev)
    return model, model_old, classes


def test_full_step_matches_reference_golden_fp32():
    from ucd_amd.run import make_optimizer
    from ucd_amd.train import Trainer
    g = load_golden("ucd_step.npz")
    dev = torch.device("cuda:0")
    opts = _opts()
    model, model_old, classes = _build(opts, dev)
    trainer = Trainer(model, model_old, device=dev, opts=opts, classes=classes)
    optim = make_optimizer(opts, model)
    img = synth.images(501, 2, 129)
    labels = synth.seg_labels(501, 2, 129, 129, range(16, 21))
    model.train()
    state_before = {k: v.detach().cpu().clone() for k, v in model.named_parameters()}
    r = trainer.train_step(img, labels, optim, None)
    torch.cuda.synchronize()
    assert r["ce"].item() == pytest.approx(float(g["ce"]), rel=1e-3)
    assert r["con"].item() == pytest.approx(float(g["con"]), rel=1e-3)
    assert r["loss"].item() == pytest.approx(float(g["loss"]), rel=1e-3)
    assert r["lkd"].item() == pytest.approx(float(g["lkd"]), rel=1e-3)
    params = dict(model.named_parameters())
    names = [k.split("::")[1] for k in g if k.startswith("grad_abs::")]
    before = {n: v for n, v in state_before.items()}
    for n in names:
        assert params[n].grad.double().abs().sum().item() == pytest.approx(float(g[f"grad_abs::{n}"]), rel=0.1), n
        # the SGD update itself (after - before) against the reference's update, as a direction + length
        p0 = before[n].flatten()[:16].double().numpy()
        up = params[n].detach().flatten()[:16].cpu().double().numpy() - p0
        ur = g[f"after_step::{n}"].astype(np.float64) - p0
        if np.linalg.norm(ur) > 1e-7:
            cos = float(up @ ur / (np.linalg.norm(up) * np.linalg.norm(ur) + 1e-30))
            # 16 weights of one output channel see only B*h*w = 162 pixels here: a couple of sign flips move them a lot
            assert cos > 0.9 and 0.5 < np.linalg.norm(up) / np.linalg.norm(ur) < 2.0, (n, cos)
    np.testing.assert_allclose(model.body.mod1.bn1.running_mean.cpu().numpy(), g["running_mean_after"],
                               rtol=1e-4, atol=1e-6)


def test_logits_and_features_match_reference_golden():
    g = load_golden("model_full.npz")
    dev = torch.device("cuda:0")
    opts = _opts()
    model, model_old, classes = _build(opts, dev)
    img = synth.images(500, 2, 65).to(dev)
    with torch.no_grad():
        lt, ft = model_old(img.clone())
        np.testing.assert_allclose(ft["sem"].cpu().numpy(), g["teacher_sem"], rtol=1e-3, atol=1e-3)
        assert lt.double().abs().sum().item() == pytest.approx(float(g["teacher_logits_abs"]), rel=1e-3)
        assert ft["pre_logits"].double().abs().sum().item() == pytest.approx(float(g["teacher_pl_abs"]), rel=1e-3)
        assert ft["body"].double().abs().sum().item() == pytest.approx(float(g["teacher_body_abs"]), rel=1e-3)
        model.eval()
        ls, fs = model(img.clone())
        np.testing.assert_allclose(fs["sem"].cpu().numpy(), g["student_eval_sem"], rtol=1e-3, atol=1e-3)
    model.train()
    ls, fs = model(img.clone())
    # Train mode on a 65x65 input: the batch statistics of the last stages are taken over 2 x 5 x 5 = 50 values per
    # channel, which amplifies fp32 rounding noise by orders of magnitude.  Measured on MI355X
    # (tools/determinism_probe.py, tools/poison_probe.py): one MIOpen fp32 1x1 convolution (mod2.block2.conv1) is not
    # bit-reproducible run to run (1e-6, atomics in its solver) and that alone moves this output by 3-4e-3 in relative L2
    # between two runs of the SAME code on the SAME input.  The bar here is therefore 1e-2 in L2; the 1e-3 bar of the
    # path is carried by the eval-mode comparisons above (frozen statistics) and by the 129x129 train step
    # (test_full_step_matches_reference_golden_fp32).  The run-to-run spread is asserted too, so a regression that
    # pushes the difference to the reference above our own noise floor is caught.
    got, ref = fs["sem"].detach().cpu().numpy(), g["student_train_sem"]
    with torch.no_grad():
        again = model(img.clone())[1]["sem"].cpu().numpy()
    noise = np.linalg.norm(got - again) / np.linalg.norm(ref)
    err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
    assert err < max(1e-2, 3 * noise), (err, noise)
    np.testing.assert_allclose(got, ref, rtol=2e-2, atol=5e-2)
    got = ls.detach().flatten()[torch.from_numpy(g["sample_idx"]).to(dev)].cpu().numpy()
    ref = g["student_train_logits_sample"]
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 1e-2
    np.testing.assert_allclose(model.cls[1].bias.detach().cpu().numpy(), g["new_head_bias"], rtol=1e-6)


def test_checkpoint_roundtrip_reference_layout(tmp_path):
    """save_ckpt writes the reference's dictionary (run.py:32-43) with module.-prefixed keys."""
    from ucd_amd.ddp import DistributedDataParallel
    from ucd_amd.run import make_optimizer, save_ckpt
    from ucd_amd.scheduler import PolyLR
    from ucd_amd.train import Trainer
    dev = torch.device("cuda:0")
    opts = _opts()
    model, model_old, classes = _build(opts, dev)
    optim = make_optimizer(opts, model)
    sched = PolyLR(optim, max_iters=100)
    ddp = DistributedDataParallel(model)
    trainer = Trainer(ddp, model_old, device=dev, opts=opts, classes=classes)
    path = str(tmp_path / "15-5-voc_test_1.pth")
    save_ckpt(path, ddp, trainer, optim, sched, 3, 0.5)
    ck = torch.load(path, map_location="cpu")
    assert set(ck) == {"epoch", "model_state", "optimizer_state", "scheduler_state", "best_score", "trainer_state"}
    keys = list(ck["model_state"])
    assert all(k.startswith("module.") for k in keys)
    for k in ("module.body.mod1.conv1.weight", "module.body.mod3.block2.convs.bn2.running_var",
              "module.head.map_convs.3.weight", "module.head.red_bn.weight", "module.cls.1.bias"):
        assert k in ck["model_state"], k
    assert ck["trainer_state"] == {"regularizer": None}
    ddp.load_state_dict(ck["model_state"], strict=True)


@pytest.mark.usefixtures("deterministic_stats")
def test_bf16_working_weights_equal_per_call_casts(tmp_path):
    """--opt_level O1 with the flat bf16 working copies (ucd_amd/master.py) is the same arithmetic as autocast's
    per-call weight casts: two steps from the same state give the same losses and the same updated fp32 weights,
    and the checkpoint keeps the reference's keys and values."""
    from ucd_amd.ddp import DistributedDataParallel
    from ucd_amd.run import make_optimizer, save_ckpt
    from ucd_amd.scheduler import PolyLR
    from ucd_amd.train import Trainer
    dev = torch.device("cuda:0")
    img = synth.images(501, 2, 129)
    labels = synth.seg_labels(501, 2, 129, 129, range(16, 21))
    results = []
    # MIOpen's default pick for some narrow 1x1 convolutions is not run-to-run reproducible (tools/determinism_probe.py)
    # and this small random network amplifies that to several percent; with deterministic solvers the two modes are
    # bit-identical, which is the claim
    torch.backends.cudnn.deterministic = True
    for shadows in (False, True):
        opts = _opts(["--opt_level", "O1"])
        opts.bf16_weights = shadows
        opts.graph_teacher = False
        model, model_old, classes = _build(opts, dev)
        optim = make_optimizer(opts, model)
        ddp = DistributedDataParallel(model, bf16_weights=shadows)
        assert (ddp.bf16_weights is not None) == shadows
        trainer = Trainer(ddp, model_old, device=dev, opts=opts, classes=classes)
        ddp.train()
        r1 = {k: v.item() for k, v in trainer.train_step(img, labels, optim, None).items()}
        r2 = {k: v.item() for k, v in trainer.train_step(img, labels, optim, None).items()}
        state = {k: v.detach().float().cpu().clone() for k, v in ddp.state_dict().items()}
        results.append((r1, r2, state))
        if shadows:
            path = str(tmp_path / "ck.pth")
            save_ckpt(path, ddp, trainer, optim, PolyLR(optim, max_iters=10), 0, 0.0)
            ck = torch.load(path, map_location="cpu")["model_state"]
            assert set(ck) == set(state)
            for k in ("module.body.mod1.conv1.weight", "module.head.map_convs.3.weight"):
                assert torch.equal(ck[k].float(), state[k]) and ck[k].dtype == torch.float32
    torch.backends.cudnn.deterministic = False
    (a1, a2, sa), (b1, b2, sb) = results
    for k in ("loss", "ce", "lkd", "con"):
        assert b1[k] == pytest.approx(a1[k], rel=1e-6), k          # identical forward
        assert b2[k] == pytest.approx(a2[k], rel=1e-6), k          # ... and identical update (deterministic convs)
    for k in ("module.body.mod1.conv1.weight", "module.body.mod4.block3.convs.conv2.weight",
              "module.head.red_conv.weight", "module.body.mod5.block1.convs.bn3.weight"):
        d = (sa[k] - sb[k]).norm() / sa[k].norm()
        assert d < 1e-6, (k, d.item())


def test_validation_loop_device_metrics():
    """Trainer.validate (train.py:185-270 of the reference): class loss + confusion matrix accumulated on the device;
    an untrained student predicts some class everywhere, so only the invariants are asserted: every labelled pixel is
    counted once and the scores are consistent with the matrix."""
    from ucd_amd.metrics import StreamSegMetrics
    from ucd_amd.run import SyntheticSegmentation
    from ucd_amd.train import Trainer
    dev = torch.device("cuda:0")
    opts = _opts(["--opt_level", "O1"])
    model, model_old, classes = _build(opts, dev)
    trainer = Trainer(model, model_old, device=dev, opts=opts, classes=classes)
    ds = SyntheticSegmentation(4, 129, list(range(16, 21)), seed=3)
    loader = torch.utils.data.DataLoader(ds, batch_size=2)
    metrics = StreamSegMetrics(21)
    (class_loss, reg_loss), score, _ = trainer.validate(loader, metrics)
    assert torch.isfinite(class_loss).item() and class_loss.item() > 0
    labelled = sum(int(((lab >= 0) & (lab < 21)).sum()) for _, lab in loader)
    assert metrics.confusion_matrix.is_cuda and int(metrics.confusion_matrix.sum().item()) == labelled
    assert score["Total samples"] == 4 and 0.0 <= score["Mean IoU"] <= 1.0 and 0.0 <= score["Overall Acc"] <= 1.0


def test_config0_voc_19_1_step0_ft_matches_reference_golden():
    """BASELINE.json configs[0] on the GPU path: VOC 19-1 step 0, --method FT (no teacher: plain CE through the fused
    up-sampling + loss kernel), 2 synthetic 256x256 images, fp32 - loss and logits within 1e-3 of the reference's CPU
    run (golden captured through the reference's own model classes)."""
    from ucd_amd.run import build_models, make_optimizer
    from ucd_amd.train import Trainer
    g = load_golden("cfg0_step.npz")
    dev = torch.device("cuda:0")
    opts = argparser.modify_command_options(argparser.get_argparser().parse_args(
        ["--method", "FT", "--task", "19-1", "--step", "0", "--lr", "0.01", "--no_pretrained", "--norm_act", "iabn_sync"]))
    classes = tasks.get_per_task_classes("voc", "19-1", 0)
    assert classes == [20]
    torch.backends.cudnn.allow_tf32 = False
    model, model_old = build_models(opts, dev, classes)
    assert model_old is None
    model.load_state_dict(synth.fill_state_dict({k: v.cpu() for k, v in model.state_dict().items()}, 43))
    trainer = Trainer(model, None, device=dev, opts=opts, classes=classes)
    optim = make_optimizer(opts, model)
    img = synth.images(777, 2, 256)
    labels = synth.seg_labels(777, 2, 256, 256, range(1, 20))
    model.train()
    with torch.no_grad():
        pass
    before = {n: p.detach().cpu().clone() for n, p in model.named_parameters()}
    r = trainer.train_step(img, labels, optim, None)
    torch.cuda.synchronize()
    assert r["loss"].item() == pytest.approx(float(g["loss"]), rel=1e-3)
    assert r["ce"].item() == pytest.approx(float(g["loss"]), rel=1e-3)
    params = dict(model.named_parameters())
    for k in g:
        if k.startswith("grad_abs::"):
            n = k.split("::")[1]
            assert params[n].grad.double().abs().sum().item() == pytest.approx(float(g[k]), rel=0.1), n   # see module docstring
            p0 = before[n].flatten()[:16].double().numpy()
            up = params[n].detach().flatten()[:16].cpu().double().numpy() - p0
            ur = g[f"after_step::{n}"].astype(np.float64) - p0
            if np.linalg.norm(ur) > 1e-7:
                cos = float(up @ ur / (np.linalg.norm(up) * np.linalg.norm(ur) + 1e-30))
                assert cos > 0.9, (n, cos)


@pytest.mark.parametrize("dataset,task,new_ids", [("ade", "100-50", range(101, 151)), ("city", "13-6", None),
                                                  ("voc", "15-5s", range(16, 17))])
def test_other_baseline_configs_step_runs(dataset, task, new_ids):
    """BASELINE.json configs[2..4] (VOC 15-5s, ADE 100-50 with its 151-way head and K = 101 teacher classes, Cityscapes
    13-6) as whole steps at a small crop: the losses are finite, the contrastive term sees anchors, and a second step
    runs from the updated weights.  Their arithmetic is pinned piecewise (contrastive / logit-loss goldens for these label
    ranges and class counts); this is the end-to-end plumbing with those head shapes."""
    from ucd_amd.run import build_models, load_step_checkpoint, make_optimizer
    from ucd_amd.train import Trainer
    dev = torch.device("cuda:0")
    try:
        classes = tasks.get_per_task_classes(dataset, task, 1)
        labels_new, labels_old, _ = tasks.get_task_labels(dataset, task, 1)
    except (KeyError, NotImplementedError):
        pytest.skip(f"{dataset} {task} not in the task tables")
    opts = argparser.modify_command_options(argparser.get_argparser().parse_args(
        ["--method", "UCD", "--dataset", dataset, "--task", task, "--step", "1", "--lr", "0.001", "--no_pretrained",
         "--norm_act", "iabn_sync", "--opt_level", "O1"]))
    model, model_old = build_models(opts, dev, classes)
    state = synth.fill_state_dict({k: v.cpu() for k, v in model_old.state_dict().items()}, 42)
    load_step_checkpoint(opts, model, model_old, state, dev)
    trainer = Trainer(model, model_old, device=dev, opts=opts, classes=classes)
    optim = make_optimizer(opts, model)
    ids = [l for l in labels_new if l != 0][:8]
    img = synth.images(601, 2, 129)
    labels = synth.seg_labels(601, 2, 129, 129, ids)
    model.train()
    for _ in range(2):
        r = trainer.train_step(img, labels, optim, None)
        assert all(torch.isfinite(v).item() for v in r.values()), {k: v.item() for k, v in r.items()}
    assert r["con"].item() > 0 and r["ce"].item() > 0


def _capture_features(model):
    """Forward hook that keeps the student's (logits, features) of the next forward (the step itself stays one forward:
    a second one would move the running statistics)."""
    box = {}
    h = model.register_forward_hook(lambda m, args, out: box.__setitem__("out", out))
    return box, h


def _run_golden_step(gname, dataset, task, step, seed_state, crop, new_ids, extra_opts=(), calibrated=False, tol=1e-3,
                     grads_rel=0.1, train_logits_l2=None):
    """One step of the product against a golden of tests/golden/make_goldens.py::_ucd_step.  ``tol``: the bar on the losses and
    on the frozen teacher's logits (1e-3 for fp32 = north_star's; the bf16 tests pass their own).  ``train_logits_l2``: bf16
    only - the relative-L2 bar on the student's TRAIN-mode logits instead of 5 * tol (see BF16_TRAIN_LOGITS_L2)."""
    from conftest import assert_matches_compact
    from ucd_amd.run import build_models, load_step_checkpoint, make_optimizer
    from ucd_amd.train import Trainer
    import torch.nn.functional as F
    g = load_golden(gname)
    seed, B, S = [int(v) for v in g["cfg"]]
    assert S == crop
    dev = torch.device("cuda:0")
    opts = argparser.modify_command_options(argparser.get_argparser().parse_args(
        ["--method", "UCD", "--dataset", dataset, "--task", task, "--step", str(step), "--lr", "0.001", "--no_pretrained",
         "--norm_act", "iabn_sync", *extra_opts]))
    classes = tasks.get_per_task_classes(dataset, task, step)
    torch.backends.cudnn.allow_tf32 = False
    model, model_old = build_models(opts, dev, classes)
    state = synth.fill_state_dict({k: v.cpu() for k, v in model_old.state_dict().items()}, seed_state, calibrated=calibrated)
    optim = make_optimizer(opts, model)
    net = model
    if opts.opt_level != "O0":
        # exactly what bench.py / run.py build for the benchmarked mode: flat fp32 masters + bf16 working weights + cached
        # flipped weights behind the gradient-bucket wrapper, stepped by the one-launch optimiser
        from ucd_amd.ddp import DistributedDataParallel
        model = DistributedDataParallel(model, delay_allreduce=True, bf16_weights=True)
    load_step_checkpoint(opts, model, model_old, state, dev)
    trainer = Trainer(model, model_old, device=dev, opts=opts, classes=classes)
    img = synth.images(seed, B, S)
    labels = synth.seg_labels(seed, B, S, S, new_ids)
    # teacher (eval): low-resolution logits and sampled full-resolution logits
    with torch.no_grad(), trainer._autocast():
        lt, ft = model_old(img.to(dev).contiguous(memory_format=torch.channels_last))
    # "within 1e-3 relative": relative to the magnitude of the logits (the synthetic checkpoint's teacher logits reach 1e5,
    # an element that happens to cancel to ~0 cannot be held to 1e-3 of ITSELF)
    tscale = float(np.abs(g["teacher_sem"]).max()) if "teacher_sem" in g else float(np.abs(g["teacher_sem::samples"]).max())
    from conftest import sample_idx
    tsem = ft["sem"].float().cpu().numpy()
    got = lt.float().flatten()[torch.from_numpy(sample_idx(lt.numel(), 256)).to(dev)].cpu().numpy()
    if tol <= 1e-3:
        assert_matches_compact(g, "teacher_sem", tsem, rtol=tol, atol=tol * tscale)
        np.testing.assert_allclose(got, g["teacher_logits_sample"], rtol=tol, atol=tol * tscale)
    else:
        # bf16: the bar is on the relative L2 of the sampled logits (1.5 * tol) and 3 * tol of the logit scale on every single
        # one (107 layers of 2^-9 roundings: one element in a few hundred reaches 1.5-2 % of the scale)
        ts = g["teacher_sem::samples"]
        mine = tsem.reshape(-1)[sample_idx(tsem.size, 512)]
        l2s, l2f = np.linalg.norm(mine - ts) / np.linalg.norm(ts), np.linalg.norm(got - g["teacher_logits_sample"]) / np.linalg.norm(g["teacher_logits_sample"])
        print(gname, opts.opt_level, "teacher logits rel-L2 (low-res, full-res)", l2s, l2f, "max abs / scale",
              np.abs(mine - ts).max() / tscale)
        assert l2s < 1.5 * tol and l2f < 1.5 * tol, (l2s, l2f)      # measured 1.0-1.2e-2 on the three configurations
        np.testing.assert_allclose(mine, ts, rtol=0, atol=3 * tol * tscale)
        np.testing.assert_allclose(got, g["teacher_logits_sample"], rtol=0, atol=3 * tol * tscale)
    model.train()
    box, hook = _capture_features(net)
    r = trainer.train_step(img, labels, optim, None)
    hook.remove()
    torch.cuda.synchronize()
    print(gname, opts.opt_level, {k: (r[k].item(), float(g[k]), abs(r[k].item() - float(g[k])) / abs(float(g[k])))
                                  for k in ("ce", "con", "loss", "lkd")})
    for k in ("ce", "con", "loss", "lkd"):
        assert r[k].item() == pytest.approx(float(g[k]), rel=tol), (k, r[k].item(), float(g[k]))
    sem = box["out"][1]["sem"].detach().float()
    # train-mode logits: batch statistics over B*33*33 (or 9*9) values amplify fp32 noise; L2 bar like the 65^2 test
    ref_samples = g["student_sem::samples"] if "student_sem::samples" in g else None
    logits = F.interpolate(sem, size=(S, S), mode="bilinear", align_corners=False)
    got = logits.flatten()[torch.from_numpy(g["sample_idx"]).to(dev)].cpu().numpy()
    err = np.linalg.norm(got - g["logits_sample"]) / np.linalg.norm(g["logits_sample"])
    print(gname, opts.opt_level, "student logits rel-L2", err)
    sscale = float(np.abs(g["student_sem"]).max()) if "student_sem" in g else float(np.abs(g["student_sem::samples"]).max())
    if train_logits_l2 is None:
        assert err < tol * 5, err
        np.testing.assert_allclose(got, g["logits_sample"], rtol=5 * tol, atol=5 * tol * max(1.0, float(np.abs(g["logits_sample"]).max())))
        assert_matches_compact(g, "student_sem", sem.cpu().numpy(), rtol=5 * tol, atol=5 * tol * max(1.0, sscale))
    else:
        assert err < train_logits_l2, err
    np.testing.assert_allclose(net.body.mod1.bn1.running_mean.cpu().numpy(), g["running_mean_after"], rtol=max(1e-4, tol),
                               atol=1e-6 if tol <= 1e-3 else 1e-3)
    params = dict(net.named_parameters())
    for k in g:
        if k.startswith("grad_abs::"):
            n = k.split("::")[1]
            got = params[n].grad.double().abs().sum().item()
            if tol > 1e-3 and n.startswith("body."):
                # bf16: the backward through 33 train-mode blocks of a random-weight network multiplies rounding differences
                # (the mirror image of BF16_TRAIN_LOGITS_L2): the SAME step with one kernel switched (UCD_BLOCK_LINK, UCD_OWN_WGRAD,
                # UCD_BWD_LINK, UCD_FUSED_CONV1X1 = 0) or simply run again moves these abs-sums between 0.69 and 1.26 of the fp32
                # golden (tests/diag/bf16_grad_diag.py, profiles/r03_bf16_grad_diag.txt), while the head's stay within 3 % -
                # the body gradients are held to the order of magnitude here and to 1e-2 / 2.5e-2 per block by the bench-shape
                # chain tests (tests/test_conv1x1_fused_gpu.py::test_bench_shape_block_chain..., slope 1)
                assert 0.5 < got / float(g[k]) < 2.0, (n, got, float(g[k]))
            else:
                # bf16 head parameters: 3 % on the convolutions / red_bn / classifier; the image-pooling branch's norm sees B values
                # per channel (two at B = 2: x-hat is +-1 whatever the input), its scale's gradient moved 9 % on the 2 x 768^2 case
                assert got == pytest.approx(float(g[k]), rel=grads_rel if tol <= 1e-3 else 0.15), n   # see module docstring
    return r, g


def test_full_step_at_513_matches_reference_golden_fp32():
    """BASELINE.json configs[1] at its real crop (2 x 513^2, fp32): losses, teacher low-resolution logits and sampled
    full-resolution logits within 1e-3 of the reference's CPU run.  The 33 x 33 map exceeds --pooling 32, so the teacher
    goes through the sliding-window image pooling of modules/deeplab.py:77-88."""
    _run_golden_step("ucd_step_513.npz", "voc", "15-5", 1, 42, 513, range(16, 21))


def test_full_step_at_513_calibrated_checkpoint_fp32():
    """The same 2 x 513^2 step from the CALIBRATED synthetic checkpoint (teacher logits of order 10, like a trained iabn_sync
    checkpoint; synth.fill_state_dict(calibrated=True)): fp32 product within 1e-3 of the reference."""
    _run_golden_step("ucd_step_513_cal.npz", "voc", "15-5", 1, 42, 513, range(16, 21), calibrated=True)


BF16_TOL = 1e-2      # --opt_level O1 (bf16 activations, fp16 contrastive operands, own GEMM kernels) against the reference's fp32 CPU run
# The student's TRAIN-mode logits are the one quantity bf16 STORAGE cannot hold to 1e-2 on this random-weight network: a
# batch-statistics norm removes the per-channel mean of its input, and after a leaky-ReLU that mean is 0.7-1.2 x the standard
# deviation - so every conv -> ABN(train) stage multiplies the RELATIVE size of the roundings already in the map by
# sqrt(1 + (mean/std)^2) ~ 1.2-1.5 (measured per block by tools/bf16_layer_probe.py: 0.3 % after the stem, x 1.5 at each projection
# block, x 3 through the head; the library-kernel bf16 path gives the same numbers to three digits, and so does the CPU oracle
# with every stored map rounded to bf16 - profiles/r03_bf16_layer_probe.txt).  The frozen teacher (running statistics: no mean
# removal) holds 1e-2, the losses (averages over 5e5 pixels) hold 1e-2; the train-mode logits get the measured bound.
BF16_TRAIN_LOGITS_L2 = 0.12      # measured 5.3 % (VOC 2 x 513^2), 3.2 % (ADE 3 x 512^2), 9.8 % (Cityscapes 2 x 768^2)
# Gradient abs-sums in bf16: head parameters within 15 % of the fp32 golden (3 % measured but for the B-sample pooling norm); body parameters within a factor of two (see the
# comment at the assertion in _run_golden_step: chaotic amplification of rounding through the random-weight body's backward).


def test_bf16_step_at_513_calibrated_checkpoint_within_1e2_of_the_reference():
    """The BENCHMARKED path - bf16 activations, the own GEMM / implicit-GEMM kernels with their fused ABN epilogues, the planned
    fp16 contrastive sweeps, the one-launch optimiser - held against the reference's fp32 CPU golden at 1e-2 on every loss, the
    teacher's logits and the student's sampled logits.  What made 1e-1 / 5-12 % necessary before was the
    test network (evaluation logits of 1e5 from a unit-scale random checkpoint), not bf16."""
    _run_golden_step("ucd_step_513_cal.npz", "voc", "15-5", 1, 42, 513, range(16, 21), extra_opts=("--opt_level", "O1"),
                     calibrated=True, tol=BF16_TOL, grads_rel=0.25, train_logits_l2=BF16_TRAIN_LOGITS_L2)


def test_config3_ade_100_50_whole_step_at_per_rank_shape_fp32():
    """BASELINE.json configs[3] as a WHOLE step at its per-rank shape (3 x 512^2, 151-class head, K = 101 teacher classes):
    losses, teacher and student logits within 1e-3 of the golden (reference classes; contrastive prep through the oracle with
    the label bound generalised, which the reference's hard-coded 20 cannot run - tests/golden/make_goldens.py::_ucd_step)."""
    _run_golden_step("ucd_step_ade_512.npz", "ade", "100-50", 1, 42, 512, range(101, 151), calibrated=True)


def test_config4_cityscapes_13_6_whole_step_at_per_rank_shape_fp32():
    """BASELINE.json configs[4] as a WHOLE step at its per-rank shape (2 x 768^2, 48 x 48 maps: sliding teacher pooling)."""
    _run_golden_step("ucd_step_city_768.npz", "city", "13-6", 1, 42, 768, range(14, 20), calibrated=True)


@pytest.mark.parametrize("gname,dataset,task,crop,ids", [("ucd_step_ade_512.npz", "ade", "100-50", 512, range(101, 151)),
                                                         ("ucd_step_city_768.npz", "city", "13-6", 768, range(14, 20))])
def test_config3_and_4_whole_steps_bf16(gname, dataset, task, crop, ids):
    """configs[3] / [4] in the benchmarked precision at their per-rank shapes: K = 101 contrastive path, 151-class fused logit
    losses, 48 x 48 maps - within 1e-2 of the fp32 golden."""
    _run_golden_step(gname, dataset, task, 1, 42, crop, ids, extra_opts=("--opt_level", "O1"), calibrated=True, tol=BF16_TOL,
                     grads_rel=0.25, train_logits_l2=BF16_TRAIN_LOGITS_L2)


def test_multi_head_step_15_5s_step3_matches_reference_golden_fp32():
    """BASELINE.json configs[2], a later overlapped step: four classifier heads [16, 1, 1, 1] in one convolution, a
    three-head teacher, K = 18."""
    _run_golden_step("ucd_step_15_5s_step3.npz", "voc", "15-5s", 3, 44, 129, [18])


def test_config0_logits_and_sem_match_reference_golden():
    """configs[0] again for what the loss test does not look at: sampled full-resolution logits and the low-resolution
    logits of the train-mode forward (256^2: 16 x 16 maps, 512 values per channel in the batch statistics)."""
    import torch.nn.functional as F
    from ucd_amd.run import build_models
    g = load_golden("cfg0_step.npz")
    dev = torch.device("cuda:0")
    opts = argparser.modify_command_options(argparser.get_argparser().parse_args(
        ["--method", "FT", "--task", "19-1", "--step", "0", "--lr", "0.01", "--no_pretrained", "--norm_act", "iabn_sync"]))
    torch.backends.cudnn.allow_tf32 = False
    model, _ = build_models(opts, dev, [20])
    model.load_state_dict(synth.fill_state_dict({k: v.cpu() for k, v in model.state_dict().items()}, 43))
    model.train()
    with torch.no_grad():
        logits, feat = model(synth.images(777, 2, 256).to(dev))
    assert logits.double().abs().sum().item() == pytest.approx(float(g["logits_abs"]), rel=1e-3)
    got = logits.flatten()[torch.from_numpy(g["sample_idx"]).to(dev)].cpu().numpy()
    assert np.linalg.norm(got - g["logits_sample"]) / np.linalg.norm(g["logits_sample"]) < 1e-3
    np.testing.assert_allclose(got, g["logits_sample"], rtol=2e-3, atol=2e-3)
    # fp32 MIOpen convolutions: the solver a fresh box picks (find mode times them) may be Winograd, ~1e-3 of the activation
    # scale per layer; 2 of 640 samples reached 2.8e-3 on one box, hence 4e-3 absolute on values of magnitude ~3
    np.testing.assert_allclose(feat["sem"].cpu().numpy()[:, :, ::4, ::4], g["sem"], rtol=2e-3, atol=4e-3)


@pytest.mark.parametrize("tag", ["33", "48", "odd"])
def test_aspp_eval_pooling_on_the_gpu_matches_reference_golden(tag):
    """DeeplabV3 in evaluation mode on maps larger than the pooling window (modules/deeplab.py:77-88) through the HIP
    layers: 33 x 33 (the teacher at 513^2), 48 x 48 (768^2) and an even window on an odd map; fp32 and bf16."""
    from functools import partial
    from conftest import assert_matches_compact
    from ucd_amd.abn import InPlaceABNSync
    from ucd_amd.blocks import DeeplabV3
    g = load_golden("aspp_eval.npz")
    dev = torch.device("cuda:0")
    seed, B, C, H, W, pool = [int(v) for v in g[f"cfg_{tag}"]]
    norm = partial(InPlaceABNSync, activation="leaky_relu", activation_param=0.01)
    head = DeeplabV3(C, 32, 16, norm_act=norm, out_stride=16, pooling_size=pool)
    head.load_state_dict(synth.fill_state_dict(head.state_dict(), 21))
    head = head.to(dev).to(memory_format=torch.channels_last).eval()
    x = synth.t_normal(seed, (B, C, H, W), stream=1).to(dev).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        y = head(x.clone())
        assert_matches_compact(g, f"eval_{tag}", y.float().cpu().numpy(), rtol=1e-3, atol=1e-4)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            yb = head(x.clone())
    ref = y.float()
    assert ((yb.float() - ref).norm() / ref.norm()).item() < 2e-2


# ---- several consecutive steps: training BEHAVIOUR of the benchmarked mode ---------------------------
TRAJ_UPDATE_NAMES = ("body.mod1.conv1.weight", "body.mod2.block1.convs.conv1.weight", "body.mod3.block2.convs.conv2.weight",
                     "body.mod4.block10.convs.conv3.weight", "body.mod5.block3.convs.conv3.weight", "head.map_convs.2.weight",
                     "head.red_conv.weight", "cls.1.weight")


# bf16 mode against the REFERENCE's updates: cosine floors per part of the network (body: the chaotic random-weight stack, see the
# comment at the bf16-vs-fp32 assertion below; head / classifier: tight)
TRAJ_BF16_REF_COS = {"body": 0.75, "head": 0.97, "cls": 0.97}


def _trajectory(opt_level, steps, step_graph="0"):
    """``steps`` iterations of the product on the fixed 2 x 513^2 batch of tests/golden/make_goldens.py::gold_traj513 (calibrated
    checkpoint, lr 1e-3, no scheduler); returns the per-step losses and the accumulated update of a few parameters."""
    from ucd_amd import switches
    from ucd_amd.ddp import DistributedDataParallel
    from ucd_amd.run import build_models, load_step_checkpoint, make_optimizer
    from ucd_amd.train import Trainer
    dev = torch.device("cuda:0")
    extra = () if opt_level == "O0" else ("--opt_level", opt_level)
    opts = argparser.modify_command_options(argparser.get_argparser().parse_args(
        ["--method", "UCD", "--dataset", "voc", "--task", "15-5", "--step", "1", "--lr", "0.001", "--no_pretrained",
         "--norm_act", "iabn_sync", *extra]))
    classes = tasks.get_per_task_classes("voc", "15-5", 1)
    torch.backends.cudnn.allow_tf32 = False
    model, model_old = build_models(opts, dev, classes)
    state = synth.fill_state_dict({k: v.cpu() for k, v in model_old.state_dict().items()}, 42, calibrated=True)
    optim = make_optimizer(opts, model)
    net = model
    if opt_level != "O0":
        model = DistributedDataParallel(model, delay_allreduce=True, bf16_weights=True)
    load_step_checkpoint(opts, model, model_old, state, dev)
    switches.set("UCD_STEP_GRAPH", step_graph)
    try:
        trainer = Trainer(model, model_old, device=dev, opts=opts, classes=classes)
        img = synth.images(502, 2, 513)
        labels = synth.seg_labels(502, 2, 513, 513, range(16, 21))
        model.train()
        params = dict(net.named_parameters())
        before = {n: params[n].detach().double().cpu().clone() for n in TRAJ_UPDATE_NAMES}
        rec = {k: [] for k in ("ce", "con", "lkd")}
        for _ in range(steps):
            r = trainer.train_step(img, labels, optim, None)
            for k in rec:
                rec[k].append(r[k].item())
        torch.cuda.synchronize()
        upd = {n: params[n].detach().double().cpu() - before[n] for n in TRAJ_UPDATE_NAMES}
        extra_out = {"cls1_bias": params["cls.1.bias"].detach().cpu().numpy().copy(),
                     "running_mean": net.body.mod1.bn1.running_mean.cpu().numpy().copy(),
                     "graph_steps": getattr(trainer, "graph_steps", 0)}
    finally:
        switches.unset("UCD_STEP_GRAPH")
    return {k: np.asarray(v) for k, v in rec.items()}, upd, extra_out


def test_twenty_step_trajectory_fp32_and_bf16_against_the_reference():
    """20 SGD steps on one fixed batch (configs[1] at 2 x 513^2, calibrated checkpoint).  Golden = the reference's own classes run
    for 20 steps in fp32 on the CPU (ucd_traj_513_cal.npz).  The fp32 product is held to it at 1e-3 over the first five steps (and a
    measured bound afterwards: 20 steps of a train-mode network compound the ~3e-4 forward differences of two fp32 implementations),
    the benchmarked bf16 mode to the fp32 product AND to the reference at 2e-2 at EVERY step; both must train (ce and the
    distillation term fall like the reference's).  The accumulated parameter updates of eight layers across the network are compared
    by DIRECTION (cosine) between the two precisions: a gradient kernel with a wrong sign or scale on the composed network cannot
    hide behind abs-sums here."""
    g = load_golden("ucd_traj_513_cal.npz")
    steps = int(g["cfg"][3])
    f32, up32, ex32 = _trajectory("O0", steps)
    b16, up16, ex16 = _trajectory("O1", steps)
    for k in ("ce", "con", "lkd"):
        rel32 = np.abs(f32[k] - g[k]) / np.abs(g[k])
        rel16 = np.abs(b16[k] - g[k]) / np.abs(g[k])
        relab = np.abs(b16[k] - f32[k]) / np.abs(f32[k])
        print(k, "fp32 vs reference: first 5 max %.2e, all max %.2e | bf16 vs reference max %.2e | bf16 vs fp32 max %.2e"
              % (rel32[:5].max(), rel32.max(), rel16.max(), relab.max()))
        assert rel32[:5].max() < 1e-3, (k, rel32)
        assert rel32.max() < 5e-3, (k, rel32)
        assert rel16.max() < 2e-2, (k, rel16)
        assert relab.max() < 2e-2, (k, relab)
    for traj in (f32, b16):
        assert traj["ce"][-1] < 0.7 * traj["ce"][0] and traj["lkd"][-1] < 0.95 * traj["lkd"][0]      # reference: 0.61, 0.90
        assert np.all(np.diff(traj["ce"]) < 0)                                                      # like the reference's
    np.testing.assert_allclose(ex32["cls1_bias"], g["cls1_bias_after"], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(ex16["cls1_bias"], g["cls1_bias_after"], rtol=2e-2, atol=1e-5)
    np.testing.assert_allclose(ex32["running_mean"], g["running_mean_after"], rtol=1e-3, atol=1e-6)
    # the REFERENCE's accumulated updates (512 evenly spaced elements of each of the eight tensors + the full length,
    # tests/golden/make_goldens.py::gold_traj513) hold the fp32 product's update DIRECTION and LENGTH - not only its losses - and the
    # bf16 mode is compared with the reference, not with the product's own fp32 path
    for i, n in enumerate(TRAJ_UPDATE_NAMES):
        ref = torch.from_numpy(g["upd"][i])
        idx = torch.from_numpy(np.linspace(0, up32[n].numel() - 1, ref.numel()).astype(np.int64))
        for tag, up, cos_floor, len_tol in (("fp32", up32, 0.99, 0.03), ("bf16", up16, TRAJ_BF16_REF_COS[n.split(".")[0]], 0.25)):
            mine = up[n].flatten()[idx]
            cos = float(mine @ ref / (mine.norm() * ref.norm() + 1e-300))
            ratio = float(up[n].norm() / float(g["upd_norm"][i]))
            print(f"update over {steps} steps, {tag} vs the reference: {n}: cosine {cos:.4f} length ratio {ratio:.3f}")
            assert cos > cos_floor and abs(ratio - 1.0) < len_tol, (tag, n, cos, ratio)
    for n in TRAJ_UPDATE_NAMES:
        a, b = up32[n].flatten(), up16[n].flatten()
        cos = float(a @ b / (a.norm() * b.norm() + 1e-300))
        ratio = float(b.norm() / (a.norm() + 1e-300))
        print(f"update over {steps} steps, bf16 vs fp32: {n}: cosine {cos:.4f} length ratio {ratio:.3f}")
        # measured on MI355X: head and classifier 0.990 - 0.995, lengths within 2 %.  The body layers depend on WHICH rounding
        # realisation of the random-weight body a build runs (BF16_TRAIN_LOGITS_L2: chaos, not error - a sign or scale error of a
        # gradient kernel would give a cosine <= 0 or a ratio far from 1): builds whose forward is bit-identical sit together
        # (0.89 - 0.91, ratio 1.04 - 1.08, run to run +-0.01 from the library's atomics), a build whose per-tile statistics are summed
        # in another order - same products bit for bit, mean / invstd equal to 1e-6 (test_small_grids_on_64_column_tiles...) - is
        # another realisation: 0.84 - 0.87 and 1.10 - 1.17 over eleven runs with the 64-column tiles of round 4, 0.88 with them on
        # the N = 256 layers only, 0.90 - 0.94 on earlier kernel generations.  The floors leave room for that spread.
        # PINNED: these floors may not be lowered to follow a code change - a build that falls below them needs a new
        # explanation AND the reference comparison above still green; the fp32-vs-reference bound (0.99) is the tight one.
        floor = 0.78 if n.startswith("body.") else 0.98
        assert cos > floor and 0.85 < ratio < 1.25, (n, cos, ratio)


def _scheduled_steps(step_graph, steps=8, batch=3, crop=257, reload_at=None, extra_args=(), probe=None):
    """``steps`` iterations of the benchmarked mode with PolyLR stepping every iteration (train.py:150-151), with or without the
    whole-step graph; returns losses per step, a few parameters afterwards and the number of replayed iterations."""
    from ucd_amd import switches
    from ucd_amd.ddp import DistributedDataParallel
    from ucd_amd.run import build_models, load_step_checkpoint, make_optimizer
    from ucd_amd.scheduler import PolyLR
    from ucd_amd.train import Trainer
    dev = torch.device("cuda:0")
    opts = argparser.modify_command_options(argparser.get_argparser().parse_args(
        ["--method", "UCD", "--dataset", "voc", "--task", "15-5", "--step", "1", "--lr", "0.001", "--no_pretrained",
         "--norm_act", "iabn_sync", "--opt_level", "O1"] + list(extra_args)))
    classes = tasks.get_per_task_classes("voc", "15-5", 1)
    model, model_old = build_models(opts, dev, classes)
    state = synth.fill_state_dict({k: v.cpu() for k, v in model_old.state_dict().items()}, 42, calibrated=True)
    optim = make_optimizer(opts, model)
    sched = PolyLR(optim, max_iters=steps + 2, power=0.9)        # a steep schedule: a frozen learning rate would show at once
    net = model
    model = DistributedDataParallel(model, delay_allreduce=True, bf16_weights=True)
    load_step_checkpoint(opts, model, model_old, state, dev)
    switches.set("UCD_STEP_GRAPH", step_graph)
    torch.backends.cudnn.deterministic = True
    try:
        trainer = Trainer(model, model_old, device=dev, opts=opts, classes=classes)
        model.train()
        rec, lrs = [], []
        for it in range(steps):
            img = synth.images(700 + it % 2, batch, crop)             # two alternating batches: the static inputs must be refreshed
            labels = synth.seg_labels(700 + it % 2, batch, crop, crop, range(16, 21))
            lrs.append(optim.param_groups[0]["lr"])
            if it == reload_at:                                      # what resuming from a checkpoint does to a live optimiser
                optim.load_state_dict(optim.state_dict())
            r = trainer.train_step(img, labels, optim, sched)
            rec.append([r[k].item() for k in ("ce", "con", "lkd", "loss")])
        torch.cuda.synchronize()
        params = dict(net.named_parameters())
        after = {n: params[n].detach().float().cpu().clone() for n in TRAJ_UPDATE_NAMES}
        after["running_var"] = net.body.mod4.block5.convs.bn2.running_var.cpu().clone()
        if probe is not None:
            probe(net, after)
        return np.asarray(rec), after, trainer.graph_steps, lrs, trainer.step_graph_error
    finally:
        torch.backends.cudnn.deterministic = False
        switches.unset("UCD_STEP_GRAPH")


@pytest.mark.usefixtures("deterministic_stats")
def test_fix_bn_freezes_the_norm_parameters_and_nothing_else():
    """``--fix_bn`` (run.py:169-170 -> segmentation_module.py:138-143: every norm layer .eval(), weight / bias requires_grad False) is
    followed by ``model.train()`` at the head of every epoch (train.py:94), which puts the norm layers back into training mode: the
    flag's lasting effect is on the norm PARAMETERS - batch statistics still normalise and still move the running statistics.  Three
    captured-graph-free iterations with and without the flag from the same checkpoint: the first iteration's losses agree (same
    forward), gamma / beta stay bit-identical under the flag and move without it, the running statistics move in both, the
    convolution weights move in both and by the same first update."""
    names = {}

    def probe(net, after):
        bn = net.body.mod3.block2.convs.bn2
        after["bn_weight"], after["bn_bias"] = bn.weight.detach().float().cpu().clone(), bn.bias.detach().float().cpu().clone()
        after["bn_rm"] = bn.running_mean.detach().float().cpu().clone()
        after["frozen"] = sum(1 for m in net.modules() if hasattr(m, "running_mean") and not m.weight.requires_grad)
        after["norms"] = sum(1 for m in net.modules() if hasattr(m, "running_mean"))
        after["conv"] = net.body.mod3.block2.convs.conv2.weight.detach().float().cpu().clone()

    ref0 = {}
    _scheduled_steps("0", steps=0, probe=lambda net, after: (probe(net, after), ref0.update(after)))
    plain_l, plain, _, _, _ = _scheduled_steps("0", steps=3, probe=probe)
    fixed_l, fixed, _, _, _ = _scheduled_steps("0", steps=3, extra_args=("--fix_bn",), probe=probe)
    assert fixed["frozen"] == fixed["norms"] > 100 and plain["frozen"] == 0
    np.testing.assert_allclose(fixed_l[0], plain_l[0], rtol=2e-4)                  # iteration 1: the same forward
    assert torch.equal(fixed["bn_weight"], ref0["bn_weight"]) and torch.equal(fixed["bn_bias"], ref0["bn_bias"])
    assert not torch.equal(plain["bn_weight"], ref0["bn_weight"]) and not torch.equal(plain["bn_bias"], ref0["bn_bias"])
    assert not torch.equal(fixed["bn_rm"], ref0["bn_rm"]) and not torch.equal(plain["bn_rm"], ref0["bn_rm"])
    d_fixed, d_plain = fixed["conv"] - ref0["conv"], plain["conv"] - ref0["conv"]
    assert d_fixed.norm() > 0 and d_plain.norm() > 0
    cos = (d_fixed * d_plain).sum() / (d_fixed.norm() * d_plain.norm())
    assert cos > 0.9, cos            # three updates from the same start; they part ways only through the norm parameters


@pytest.mark.parametrize("stat_atomic", ["0", "1"])
def test_whole_step_graph_replays_the_eager_iteration(stat_atomic):
    """The captured iteration (Trainer._graph_step: teacher + student forward, losses, backward, bucket hand-over, one-launch
    optimiser with its hyper-parameters on the device) is the eager iteration: same losses at every step, same parameters and
    running statistics after 8 steps on two alternating batches under a steep PolyLR - the learning rate reaches the replayed
    optimiser kernel, the inputs reach the static buffers, and nothing that ran on the host during the capture is missing from
    the replay.  stat_atomic 0: the deterministic statistics path (tight bounds: two runs differ by the library's stem weight
    gradient alone); 1: the default since round 5 - fp32-atomic column sums whose order changes from run to run AND the arena's
    one-fill-per-step inside the captured graph - held to the bounds two EAGER runs of that mode keep between each other."""
    from ucd_amd import switches
    switches.set("UCD_STAT_ATOMIC", stat_atomic)
    try:
        eager, pe, n_e, lrs_e, _ = _scheduled_steps("0")
        graph, pg, n_g, lrs_g, err = _scheduled_steps("1")
    finally:
        switches.unset("UCD_STAT_ATOMIC")
    assert err is None, err
    assert n_e == 0 and n_g == 8 - 3, (n_e, n_g)          # three eager warm-up iterations, then replays only
    assert lrs_e == lrs_g and lrs_e[-1] < 0.35 * lrs_e[0]
    print("eager vs graph losses, max rel:", np.abs(eager - graph).max(0) / np.abs(eager).max(0))
    # "0": since the logit gradient is accumulated in fixed point (round 5, csrc/seglogit_loss.hip) two runs of this configuration are
    # bit-identical (tools/nd_matrix.sh: 0 of 48 runs part from the first of their process; 14 of 96 before, two discrete trajectories
    # 5e-4 apart after one update, amplified to 2.4e-3 by the eight updates of this schedule) - the bound is the round-4 one again
    np.testing.assert_allclose(graph, eager, rtol=2e-3 if stat_atomic == "0" else 1e-2)
    for n in pe:
        d = ((pe[n] - pg[n]).norm() / pe[n].norm()).item()
        # deterministic mode: two of three runs are bit-identical (losses equal at every step); in the third a kernel whose float
        # atomics are not ordered (the fused logit-loss kernel's LDS / global atomics, the library's stem weight gradient) moves the
        # result by 1.3e-4 - 1.5e-4 (measured over six runs in round 5: 0, 0, 0, 1.28e-4, 0, 1.54e-4) - hence 3e-4, not 1e-4
        assert d < (3e-4 if stat_atomic == "0" else 5e-3), (n, d)
    # and the update itself took the schedule: against a run whose optimiser never saw the decay the weights differ visibly
    first = eager[0]
    assert np.all(np.isfinite(graph)) and graph[-1][3] < first[3]


@pytest.mark.usefixtures("deterministic_stats")
def test_step_graph_is_dropped_and_rebuilt_when_the_optimiser_state_moves():
    """``optim.load_state_dict`` (resume) replaces the momentum buffers: the captured optimiser launch would update the OLD ones.
    The trainer notices (``SGD.plan_is_current``), drops the graph, runs eagerly and captures again after its warm-up count - the
    run equals the eager run with the same reload."""
    eager, pe, n_e, _, _ = _scheduled_steps("0", steps=12, reload_at=6)
    graph, pg, n_g, _, err = _scheduled_steps("1", steps=12, reload_at=6)
    assert err is None, err
    # replays: steps 3, 4, 5 (captured at step 3), eager again from the reload at step 6 for the warm-up count, replays from step 9
    assert n_e == 0 and n_g == 3 + 3, (n_e, n_g)
    # (the run-to-run spread this bound once had to cover - fp32 atomics in the logit gradient - is gone: see the replay test)
    np.testing.assert_allclose(graph, eager, rtol=2e-3)
    for n in pe:
        d = ((pe[n] - pg[n]).norm() / pe[n].norm()).item()
        # 12 steps: the library's weight-gradient kernel of the stem is not bit-reproducible between two runs (1.7e-4 on its weight
        # here); a graph that kept stepping the OLD momentum buffers would be off by > 1e-2 after six more steps at momentum 0.9
        assert d < 1e-3, (n, d)


@pytest.mark.usefixtures("deterministic_stats")
def test_deferred_weight_gradient_sums_change_no_bit_of_the_step():
    """Round 6: the slab sum of every weight gradient rides in the NEXT weight-gradient launch (csrc/wgrad.hip, SumArgs;
    ``UCD_WGRAD_DEFER``, ucd_amd/ddp.py flushes in front of the bucket copies) instead of a launch of its own (109 per step).  Same
    sums in the same order: four scheduled iterations with and without the deferral end in bit-identical losses and parameters -
    eager and replayed from the step graph."""
    from ucd_amd import switches
    runs = {}
    for defer in ("0", "1"):
        switches.set("UCD_WGRAD_DEFER", defer)
        try:
            for sg in ("0", "1"):
                runs[(defer, sg)] = _scheduled_steps(sg, steps=5 if sg == "1" else 3)
        finally:
            switches.unset("UCD_WGRAD_DEFER")
    for sg in ("0", "1"):
        (la, pa, ga, _, ea), (lb, pb, gb, _, eb) = runs[("0", sg)], runs[("1", sg)]
        assert ea is None and eb is None, (ea, eb)
        assert ga == gb and (sg == "0" or ga >= 1), (ga, gb)
        assert np.array_equal(la, lb), (sg, la, lb)
        for n in pa:
            assert torch.equal(pa[n], pb[n]), (sg, n)


@pytest.mark.usefixtures("deterministic_stats")
def test_weight_gradients_on_the_side_stream_change_no_bit_of_the_step():
    """Round 6: between the wrapper's flushes the nodes' weight gradients run on a side stream of the library
    (``UCD_WGRAD_STREAM``; include/ucd_hip.h ucd_conv_wgrad_ex flags bit 1: a fork point per call, the launch one call later, joined in
    front of the bucket copies) - off the chain of input-gradient products.  Same kernels on the same operands: scheduled iterations
    with and without it end in bit-identical losses and parameters, eager and replayed from the step graph (fork and join are
    captured as the graph's edges; an operand freed too early, a gradient read before the join or a workspace shared across the two
    streams would all show here)."""
    from ucd_amd import hip, switches
    runs = {}
    for side in ("0", "1"):
        switches.set("UCD_WGRAD_STREAM", side)
        try:
            for sg in ("0", "1"):
                runs[(side, sg)] = _scheduled_steps(sg, steps=5 if sg == "1" else 3)
                assert hip.load().ucd_conv_wgrad_mode() == 0                  # finish() switched the mode off again
        finally:
            switches.unset("UCD_WGRAD_STREAM")
    for sg in ("0", "1"):
        (la, pa, ga, _, ea), (lb, pb, gb, _, eb) = runs[("0", sg)], runs[("1", sg)]
        assert ea is None and eb is None, (ea, eb)
        assert ga == gb and (sg == "0" or ga >= 1), (ga, gb)
        assert np.array_equal(la, lb), (sg, la, lb)
        for n in pa:
            assert torch.equal(pa[n], pb[n]), (sg, n)


_SWITCH_COMBOS = [
    # the round-6 stream work off, links off, late copies off: the plainest schedule of the own kernels
    {"UCD_WGRAD_STREAM": "0", "UCD_WGRAD_DEFER": "0", "UCD_BWD_LINK": "0", "UCD_BLOCK_LINK": "0", "UCD_DDP_LATE_COPY": "0",
     "UCD_TEACHER_OVERLAP": "0"},
    # library kernels for the weight gradients, strided layers, heads and stem under the side stream / graph machinery
    {"UCD_OWN_WGRAD": "0", "UCD_OWN_STRIDED": "0", "UCD_OWN_HEADS": "0", "UCD_OWN_STEM": "0", "UCD_STEM_FOLD": "0",
     "UCD_DGRAD_VIA_FWD": "0"},
    # the side stream without deferred sums, Python twins of the C++ nodes, torch's optimiser, no projection alias
    {"UCD_WGRAD_DEFER": "0", "UCD_ABN_NODE": "0", "UCD_SGD": "torch", "UCD_PROJ_ALIAS": "0", "UCD_STEM_EVAL_FUSED": "0"},
]


@pytest.mark.usefixtures("deterministic_stats")
@pytest.mark.parametrize("combo", range(len(_SWITCH_COMBOS)))
def test_switch_combinations_give_the_same_step(combo):
    """The A/B switches are tested one at a time elsewhere; here several are flipped TOGETHER ("only the default
    combination plus single-switch A/Bs are tested").  Each combination runs three scheduled iterations eagerly and - where the
    combination allows the capture - replayed; losses stay within the bf16 bar (1e-2 relative on the total, the contrastive and
    the distillation term follow their own kernels' rounding) of the default combination's, and nothing falls back silently."""
    from ucd_amd import abn as _abn, blocks as _blocks, switches
    ref, _, _, _, e0 = _scheduled_steps("0", steps=3)
    assert e0 is None
    flips = _SWITCH_COMBOS[combo]
    node_cache = (_abn._node_mod, _blocks._node_cache[0])
    for k, v in flips.items():
        switches.set(k, v)
    if "UCD_ABN_NODE" in flips:
        _abn._node_mod, _blocks._node_cache[0] = False, False          # the node modules are looked up once per process
    try:
        for sg in ("0", "1"):
            if sg == "1" and flips.get("UCD_SGD") == "torch":
                continue                                                 # the capture needs the one-launch optimiser (train.py)
            seen = {}

            def probe(net, after):                                       # still inside the run: what the step actually used
                seen["node"] = _abn._abn_node() is not None
                seen["flips"] = {k: switches.get(k) for k in flips}
            got, _, replayed, _, err = _scheduled_steps(sg, steps=5 if sg == "1" else 3, probe=probe)
            assert err is None, err
            assert seen["flips"] == flips and seen["node"] == (flips.get("UCD_ABN_NODE") != "0"), seen
            assert sg == "0" or replayed >= 1
            assert np.isfinite(got).all()
            np.testing.assert_allclose(got[:3, 3], ref[:, 3], rtol=1e-2)                    # total loss
            np.testing.assert_allclose(got[:3, 0], ref[:, 0], rtol=1e-2)                    # cross entropy
            np.testing.assert_allclose(got[:3, 1:3], ref[:, 1:3], rtol=5e-2, atol=1e-3)     # contrastive, distillation
    finally:
        for k in flips:
            switches.unset(k)
        _abn._node_mod, _blocks._node_cache[0] = node_cache
