"""TEST INFRASTRUCTURE - numpy restatement of the reference's streaming segmentation metrics
(metrics/stream_metrics.py:34-104: per-image bincount into an [n, n] confusion matrix, labels outside [0, n) ignored;
overall / mean accuracy, frequency-weighted accuracy, mean IoU over the classes that occur)."""
import numpy as np


class StreamSegMetrics:
    def __init__(self, n_classes):
        self.n_classes = n_classes
        self.reset()

    def reset(self):
        self.confusion_matrix = np.zeros((self.n_classes, self.n_classes))
        self.total_samples = 0

    def update(self, label_trues, label_preds):
        n = self.n_classes
        for lt, lp in zip(label_trues, label_preds):            # stream_metrics.py:44-47
            lt, lp = np.asarray(lt).ravel(), np.asarray(lp).ravel()
            mask = (lt >= 0) & (lt < n)                         # :65
            self.confusion_matrix += np.bincount(n * lt[mask].astype(int) + lp[mask], minlength=n * n).reshape(n, n)
        self.total_samples += len(label_trues)

    def get_results(self):
        eps = 1e-6                                              # :80
        hist = self.confusion_matrix
        gt_sum = hist.sum(axis=1)
        mask = gt_sum != 0
        diag = np.diag(hist)
        acc = diag.sum() / hist.sum()
        acc_cls_c = diag / (gt_sum + eps)
        iu = diag / (gt_sum + hist.sum(axis=0) - diag + eps)
        freq = gt_sum / hist.sum()
        return {"Total samples": self.total_samples, "Overall Acc": acc, "Mean Acc": np.mean(acc_cls_c[mask]),
                "FreqW Acc": (freq[freq > 0] * iu[freq > 0]).sum(), "Mean IoU": np.mean(iu[mask]),
                "Class IoU": {i: (iu[i] if m else "X") for i, m in enumerate(mask)},
                "Class Acc": {i: (acc_cls_c[i] if m else "X") for i, m in enumerate(mask)}}
