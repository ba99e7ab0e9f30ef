"""CPU: the oracle (oracle/) reproduces the golden vectors captured from the reference's own Python
(tests/golden/make_goldens.py).  This is what pins the oracle; the GPU tests then compare the HIP
path with the pinned oracle."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_matches_compact, load_golden
from oracle import contrastive as OC
from oracle import losses as OL
from ucd_amd import synth

PIXCON = ["voc_15_5", "city_13_6", "voc_15_5s_step2", "voc_19_1_odd"]


def _case(g):
    cfg = [int(v) for v in g["cfg"]]
    seed, B, N, h, w, K, H, W = cfg[:8]
    return synth.contrastive_case(seed, B, N, h, w, K, H, W, cfg[8:]), (B, N, h, w, K, H, W)


@pytest.mark.parametrize("name", PIXCON)
def test_pixcon_prep_and_loss_match_reference(name):
    g = load_golden(f"pixcon_{name}.npz")
    (f_n, f_o, l_po, labels), (B, N, h, w, K, H, W) = _case(g)
    f_n = f_n.clone().requires_grad_(True)
    prep = OC.pre_contrastive_pixel(f_n, labels, l_po, f_o)
    assert prep["a"].shape[0] == int(g["A"]) and prep["c"].shape[0] == int(g["C"])
    np.testing.assert_array_equal(prep["la"].numpy(), g["la"])
    np.testing.assert_array_equal(prep["lc"].numpy(), g["lc"])
    assert_matches_compact(g, "a", prep["a"].detach().numpy(), rtol=1e-6, atol=1e-7)
    assert_matches_compact(g, "c", prep["c"].numpy(), rtol=1e-6, atol=1e-7)
    assert_matches_compact(g, "P", prep["P"].numpy(), rtol=1e-6, atol=1e-7)
    loss = OC.pixcon_loss(prep["a"], prep["c"], prep["la"], prep["lc"], prep["P"], 0.07)
    assert loss.item() == pytest.approx(float(g["loss"]), rel=1e-6)
    loss.backward()
    assert_matches_compact(g, "grad_f_n", f_n.grad.numpy(), rtol=1e-5, atol=1e-9)
    loss_nop = OC.pixcon_loss(prep["a"].detach(), prep["c"], prep["la"], prep["lc"], None, 0.07)
    assert loss_nop.item() == pytest.approx(float(g["loss_noP"]), rel=1e-6)


@pytest.mark.parametrize("name", PIXCON)
def test_bilinear_label_formula_is_bit_exact(name):
    """The explicit arithmetic the HIP prep kernel implements == F.interpolate, element for element
    (labels are truncated to integers afterwards, so one ulp matters)."""
    g = load_golden(f"pixcon_{name}.npz")
    (_, _, _, labels), (B, N, h, w, K, H, W) = _case(g)
    mine = OC.bilinear_labels_formula(labels, h, w)
    np.testing.assert_array_equal(mine, g["label_interp"])
    ref = F.interpolate(labels.float().unsqueeze(1), size=(h, w), mode="bilinear", align_corners=False)[:, 0]
    np.testing.assert_array_equal(mine, ref.numpy())


@pytest.mark.parametrize("H,h", [(513, 33), (512, 32), (768, 48), (321, 21), (100, 7)])
def test_bilinear_label_formula_sizes(H, h):
    labels = synth.seg_labels(H, 3, H, H + 2, range(1, 21), rects=5)
    ref = F.interpolate(labels.float().unsqueeze(1), size=(h, h + 1), mode="bilinear", align_corners=False)[:, 0]
    np.testing.assert_array_equal(OC.bilinear_labels_formula(labels, h, h + 1), ref.numpy())
    # the reference's int8 cast + clamps == trunc-and-range-check
    ref_i8 = ref.type(torch.int8).clone()
    ref_i8[ref_i8 < 0] = 0
    ref_i8[ref_i8 > 20] = 0
    np.testing.assert_array_equal(OC.downsample_labels(labels, h, h + 1).numpy(), ref_i8.numpy().astype(np.int64))


@pytest.mark.parametrize("name", PIXCON[:2])
def test_closed_form_backward_equals_autograd(name):
    g = load_golden(f"pixcon_{name}.npz")
    (f_n, f_o, l_po, labels), _ = _case(g)
    prep = OC.pre_contrastive_pixel(f_n, labels, l_po, f_o)
    a64 = prep["a"].double().requires_grad_(True)
    loss = OC.pixcon_loss(a64, prep["c"].double(), prep["la"], prep["lc"], prep["P"].double(), 0.07)
    loss.backward()
    l2, da, neg, G, num = OC.pixcon_loss_backward(prep["a"], prep["c"], prep["la"], prep["lc"], prep["P"], 0.07)
    assert l2.item() == pytest.approx(loss.item(), rel=1e-12)
    np.testing.assert_allclose(da.numpy(), a64.grad.numpy(), rtol=1e-9, atol=1e-14)
    # row max of S is the self pair, 1/T
    S = prep["a"] @ prep["c"].T / 0.07
    assert (S.max(dim=1)[0] - 1 / 0.07).abs().max().item() < 1e-4


def test_logit_losses_match_reference():
    g = load_golden("logit_losses.npz")
    for tag in ("voc", "city", "ade"):
        seed, Ctot, K = [int(v) for v in g[f"{tag}_cfg"]]
        x = synth.t_normal(seed, (2, Ctot, 16, 16), stream=1, scale=2.0).requires_grad_(True)
        t = synth.t_normal(seed, (2, K, 16, 16), stream=2, scale=2.0)
        lab = synth.randint(seed, (2, 16, 16), 0, Ctot + 3, stream=3)
        lab = torch.from_numpy(np.where(lab >= Ctot, 255, lab))
        ce = OL.unbiased_cross_entropy(x, lab, K)
        np.testing.assert_allclose(ce.detach().numpy(), g[f"{tag}_ce"], rtol=1e-6, atol=1e-6)
        g_ce, = torch.autograd.grad(ce.mean(), x)
        assert_matches_compact(g, f"{tag}_g_ce", g_ce.numpy(), rtol=1e-5, atol=1e-9)
        kd = OL.unbiased_kd(x, t)
        assert kd.item() == pytest.approx(float(g[f"{tag}_kd"]), rel=1e-6)
        g_kd, = torch.autograd.grad(kd, x)
        assert_matches_compact(g, f"{tag}_g_kd", g_kd.numpy(), rtol=1e-5, atol=1e-9)


def test_v1_losses_match_reference():
    g = load_golden("v1_losses.npz")
    seed, n, d = [int(v) for v in g["cfg"]]
    f = F.normalize(synth.t_normal(seed, (n, d), stream=1), dim=1)
    lab = torch.from_numpy(synth.randint(seed, (n,), 0, 5, stream=2))
    assert OC.pixcon_loss_v1(f[:, None, :], lab, 0.07).item() == pytest.approx(float(g["pixcon_T007"]), rel=1e-6)
    assert OC.pixcon_loss_v1(f[:, None, :], lab).item() == pytest.approx(float(g["pixcon_T1"]), rel=1e-6)
    f2 = F.normalize(synth.t_normal(seed, (n, 2, d), stream=3), dim=2)
    assert OC.supcon_loss(f2, lab, temperature=0.07).item() == pytest.approx(float(g["supcon"]), rel=1e-6)
    assert OC.supcon_loss(f2, lab, temperature=0.1, contrast_mode="one").item() == pytest.approx(
        float(g["supcon_one"]), rel=1e-6)
    assert OC.supcon_loss(f2).item() == pytest.approx(float(g["simclr"]), rel=1e-6)
    # the v1 loss is the V2 form's special case P = 1, c = a, no row-max shift (what the HIP kernel runs for it)
    v2 = OC.pixcon_loss(f.double(), f.double(), lab, lab, None, 0.07, shift=False)
    assert v2.item() == pytest.approx(float(g["pixcon_T007"]), rel=1e-5)


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree exists only in the build container")
def test_golden_recipe_runs_and_reproduces_the_committed_fixtures(tmp_path):
    """VERDICT r3 / ADVICE r3: the committed generator must run as committed.  Its fast entry points (contrastive, logit losses,
    the dead-file losses, the model-level goldens incl. the 2 x 129^2 step) are re-run into a scratch directory and every array must
    equal the committed .npz bit for bit; the slow entry points are at least compiled and checked for undefined names."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    script = os.path.join(here, "golden", "make_goldens.py")
    env = dict(os.environ, UCD_GOLDEN_OUT=str(tmp_path))
    r = subprocess.run([sys.executable, script, "pixcon", "logit", "v1", "model", "traj513_head"], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    made = sorted(f for f in os.listdir(tmp_path) if f.endswith(".npz"))
    assert {"pixcon_voc_15_5.npz", "logit_losses.npz", "v1_losses.npz", "model_full.npz", "model_blocks.npz", "ucd_step.npz",
            "ucd_traj_513_cal_head.npz"} <= set(made), made
    # the 20-step trajectory golden (with the reference's accumulated updates) takes ~4 minutes to regenerate; its first
    # two iterations are re-run here and must equal the committed file's prefix bit for bit: per-step losses, anchor / contrast counts
    # and the sampled two-step updates + their lengths (the keys the update-direction test reads after 20 steps come from the same code)
    head, full = np.load(os.path.join(tmp_path, "ucd_traj_513_cal_head.npz")), np.load(os.path.join(here, "golden", "ucd_traj_513_cal.npz"))
    for k in ("ce", "con", "lkd", "A", "C"):
        assert np.array_equal(head[k], full[k][:2]), k
    for k in ("upd2", "upd2_norm"):
        assert np.array_equal(head[k], full[k]), k
    assert full["upd"].shape == head["upd"].shape == (8, 512) and full["upd_norm"].shape == (8,)
    made.remove("ucd_traj_513_cal_head.npz")
    for f in made:
        new, old = np.load(os.path.join(tmp_path, f)), np.load(os.path.join(here, "golden", f))
        assert sorted(new.files) == sorted(old.files), f
        for k in new.files:
            assert np.array_equal(new[k], old[k]), (f, k)
    # every gold_* entry point: names resolve (the r3 bug was a NameError in a function the default run reaches late)
    import ast
    import builtins
    tree = ast.parse(open(script).read())
    top = {n.name for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef))}
    for n in tree.body:
        if isinstance(n, (ast.Import, ast.ImportFrom)):
            top |= {(a.asname or a.name).split(".")[0] for a in n.names}
        elif isinstance(n, ast.Assign):
            top |= {t.id for t in n.targets if isinstance(t, ast.Name)}
    for fn in (n for n in tree.body if isinstance(n, ast.FunctionDef)):
        bound = {a.arg for a in fn.args.args + fn.args.kwonlyargs} | ({fn.args.vararg.arg} if fn.args.vararg else set()) \
            | ({fn.args.kwarg.arg} if fn.args.kwarg else set())
        for sub in ast.walk(fn):
            if isinstance(sub, ast.Name) and isinstance(sub.ctx, (ast.Store, ast.Del)):
                bound.add(sub.id)
            elif isinstance(sub, (ast.FunctionDef, ast.ClassDef)):
                bound.add(sub.name)
                if isinstance(sub, ast.FunctionDef):
                    bound |= {a.arg for a in sub.args.args + sub.args.kwonlyargs}
            elif isinstance(sub, (ast.Import, ast.ImportFrom)):
                bound |= {(a.asname or a.name).split(".")[0] for a in sub.names}
            elif isinstance(sub, ast.arg):
                bound.add(sub.arg)
            elif isinstance(sub, ast.ExceptHandler) and sub.name:
                bound.add(sub.name)
        for sub in ast.walk(fn):
            if isinstance(sub, ast.Name) and isinstance(sub.ctx, ast.Load):
                assert sub.id in bound or sub.id in top or hasattr(builtins, sub.id), (fn.name, sub.id, sub.lineno)
