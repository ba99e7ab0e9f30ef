"""Validation-path metrics: the device-resident confusion matrix of ucd_amd.metrics against golden vectors
captured from the reference's own StreamSegMetrics (tests/golden/make_metrics_golden.py) and against the numpy oracle.
Integer work: the confusion matrix is bit-exact; the derived ratios are compared at 1e-12."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle.metrics import StreamSegMetrics as OracleMetrics
from ucd_amd.metrics import StreamSegMetrics


def _case(seed, n, B, H, W, present):
    rng = np.random.RandomState(seed)
    lt = rng.choice(present, size=(B, H, W)).astype(np.int64)
    lt[rng.rand(B, H, W) < 0.1] = 255
    lp = np.where(rng.rand(B, H, W) < 0.7, np.where(lt == 255, 0, lt), rng.randint(0, n, size=(B, H, W))).astype(np.int64)
    return lt, lp


def _run(cls, name, g, to=lambda a: a):
    n, present = int(g[f"{name}::n"]), list(g[f"{name}::present"])
    m = cls(n)
    for b, seed in enumerate((11, 12, 13)):
        lt, lp = _case(seed, n, 2 + b, 17, 23, present)
        m.update(to(lt), to(lp))
    return m, m.get_results()


def _check(name, g, m, r):
    cm = m.confusion_matrix.cpu().numpy() if torch.is_tensor(m.confusion_matrix) else m.confusion_matrix
    assert np.array_equal(cm, g[f"{name}::cm"])
    assert r["Total samples"] == int(g[f"{name}::total"])
    for k in ("Overall Acc", "Mean Acc", "FreqW Acc", "Mean IoU"):
        assert r[k] == pytest.approx(float(g[f"{name}::{k}"]), rel=1e-12), k
    iou = np.array([-1.0 if v == "X" else v for v in r["Class IoU"].values()])
    acc = np.array([-1.0 if v == "X" else v for v in r["Class Acc"].values()])
    np.testing.assert_allclose(iou, g[f"{name}::class_iou"], rtol=1e-12)
    np.testing.assert_allclose(acc, g[f"{name}::class_acc"], rtol=1e-12)


@pytest.mark.parametrize("name", ["voc21", "city19"])
def test_oracle_matches_reference_golden(name):
    g = load_golden("metrics.npz")
    _check(name, g, *_run(OracleMetrics, name, g))


@pytest.mark.parametrize("name", ["voc21", "city19"])
def test_host_mirror_matches_reference_golden_cpu(name):
    g = load_golden("metrics.npz")
    m, r = _run(StreamSegMetrics, name, g, to=torch.from_numpy)
    _check(name, g, m, r)
    assert "Class IoU" in m.to_str(r)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["voc21", "city19"])
def test_device_confusion_matrix_matches_reference_golden(name):
    g = load_golden("metrics.npz")
    dev = torch.device("cuda:0")
    m, r = _run(StreamSegMetrics, name, g, to=lambda a: torch.from_numpy(a).to(dev))
    assert m.confusion_matrix.is_cuda
    _check(name, g, m, r)


@pytest.mark.gpu
@pytest.mark.parametrize("Ctot,n,H,W,h,w,B", [(21, 21, 129, 129, 9, 9, 3), (20, 20, 190, 251, 12, 16, 2), (151, 151, 128, 128, 8, 8, 2),
                                            (21, 21, 513, 513, 33, 33, 4)])
def test_fused_upsample_argmax_confusion_matches_oracle(Ctot, n, H, W, h, w, B):
    """ucd_seg_confusion (bilinear up-sampling + arg-max + histogram, full-resolution logits never built) against the oracle
    path the reference takes (train.py:236-246: interpolate -> max(dim=1) -> numpy bincount per image) on the CPU: the
    confusion matrix is integer work -> bit-exact; the arg-max map may differ only where two interpolated logits tie to the
    last bit between the two implementations of the interpolation."""
    import torch.nn.functional as F
    from ucd_amd import hip, synth
    dev = torch.device("cuda:0")
    sem = synth.t_normal(5 + Ctot, (B, Ctot, h, w), stream=1) * 2.0
    labels = torch.from_numpy(synth.randint(7, (B, H, W), 0, n + 3, stream=2))
    labels[labels >= n] = 255                                             # some ignored pixels
    up = F.interpolate(sem, size=(H, W), mode="bilinear", align_corners=False)
    pred_ref = up.max(dim=1)[1]
    # the arg-max map
    s = sem.to(dev).permute(0, 2, 3, 1).reshape(B * h * w, Ctot).contiguous()
    hist = torch.zeros(n, n, dtype=torch.int64, device=dev)
    pred = torch.empty(B, H, W, dtype=torch.int64, device=dev)
    hip._check(hip.load().ucd_seg_confusion(hip.ptr(s), Ctot, hip.ptr(labels.to(dev)), B, H, W, h, w, Ctot, n, hip.ptr(hist),
                                            hip.ptr(pred), hip.stream()), "ucd_seg_confusion")
    pred = pred.cpu()
    diff = pred != pred_ref
    if diff.any():      # only exact ties of two interpolated logits may resolve differently (CPU vs device rounding order)
        top2 = up.topk(2, dim=1)[0]
        assert int(diff.sum()) <= 4 and float((top2[:, 0] - top2[:, 1])[diff].abs().max()) < 1e-5
    # the histogram: integer work, exact for the predictions the kernel made
    om = OracleMetrics(n)
    om.update(labels.numpy(), pred.numpy())
    assert np.array_equal(hist.cpu().numpy(), om.confusion_matrix)
    m = StreamSegMetrics(n)
    m.update_from_logits(labels.to(dev), sem.to(dev))
    m.update_from_logits(labels.to(dev), sem.to(dev))                     # accumulates across batches
    assert np.array_equal(m.confusion_matrix.cpu().numpy(), 2 * om.confusion_matrix)
    assert m.total_samples == 2 * B
    r = m.get_results()
    om.update(labels.numpy(), pred.numpy())
    ro = om.get_results()
    assert r["Mean IoU"] == pytest.approx(ro["Mean IoU"], rel=1e-12)
