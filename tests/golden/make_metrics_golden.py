"""Golden vectors of the reference's StreamSegMetrics (run in the build container only: imports /root/reference).
usage: python tests/golden/make_metrics_golden.py"""
import os, sys
import numpy as np
sys.path.insert(0, "/root/reference")
from metrics.stream_metrics import StreamSegMetrics     # the reference's own class

HERE = os.path.dirname(os.path.abspath(__file__))


def case(seed, n, B, H, W, present):
    rng = np.random.RandomState(seed)
    lt = rng.choice(present, size=(B, H, W)).astype(np.int64)
    lt[rng.rand(B, H, W) < 0.1] = 255                      # ignore band
    lp = np.where(rng.rand(B, H, W) < 0.7, np.where(lt == 255, 0, lt), rng.randint(0, n, size=(B, H, W))).astype(np.int64)
    return lt, lp


out = {}
for name, (n, present) in {"voc21": (21, list(range(0, 21, 2))), "city19": (19, list(range(19)))}.items():
    m = StreamSegMetrics(n)
    for b, seed in enumerate((11, 12, 13)):
        lt, lp = case(seed, n, 2 + b, 17, 23, present)
        m.update(lt, lp)
    r = m.get_results() if False else None
    # get_results() renders a matplotlib figure; take the scalars from the same arithmetic without the figure
    fig = StreamSegMetrics.confusion_matrix_to_fig
    StreamSegMetrics.confusion_matrix_to_fig = lambda self: None
    r = m.get_results()
    StreamSegMetrics.confusion_matrix_to_fig = fig
    out[f"{name}::n"] = n
    out[f"{name}::present"] = np.array(present)
    out[f"{name}::cm"] = m.confusion_matrix
    out[f"{name}::total"] = m.total_samples
    for k in ("Overall Acc", "Mean Acc", "FreqW Acc", "Mean IoU"):
        out[f"{name}::{k}"] = r[k]
    out[f"{name}::class_iou"] = np.array([-1.0 if v == "X" else v for v in r["Class IoU"].values()])
    out[f"{name}::class_acc"] = np.array([-1.0 if v == "X" else v for v in r["Class Acc"].values()])
np.savez_compressed(os.path.join(HERE, "metrics.npz"), **out)
print("wrote metrics.npz", sorted(out)[:6])
