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
