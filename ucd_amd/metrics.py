"""Streaming segmentation metrics with the confusion matrix kept on the device (SURVEY.md section 8-f3;
reference: metrics/stream_metrics.py:26-122 builds it with numpy bincount on the host per image)."""
from __future__ import annotations

import torch
import torch.distributed as dist


class StreamSegMetrics:
    def __init__(self, n_classes):
        self.n_classes = n_classes
        self.confusion_matrix = None
        self.total_samples = 0

    def reset(self):
        self.confusion_matrix = None
        self.total_samples = 0

    def update(self, label_trues, label_preds):
        lt, lp = torch.as_tensor(label_trues).reshape(-1), torch.as_tensor(label_preds).reshape(-1)
        mask = (lt >= 0) & (lt < self.n_classes)
        idx = self.n_classes * lt[mask].long() + lp[mask].long()
        hist = torch.bincount(idx, minlength=self.n_classes ** 2).reshape(self.n_classes, self.n_classes).double()
        self.confusion_matrix = hist if self.confusion_matrix is None else self.confusion_matrix + hist
        self.total_samples += 1

    def synch(self, device):
        if dist.is_available() and dist.is_initialized() and self.confusion_matrix is not None:
            cm = self.confusion_matrix.to(device)
            dist.reduce(cm, dst=0)
            self.confusion_matrix = cm

    def get_results(self):
        hist = self.confusion_matrix.cpu()
        eps = 1e-6
        gt_sum, diag = hist.sum(dim=1), hist.diag()
        mask = gt_sum != 0
        acc = diag.sum() / hist.sum()
        acc_cls_c = diag / (gt_sum + eps)
        iu = diag / (gt_sum + hist.sum(dim=0) - diag + eps)
        return {"Total samples": self.total_samples, "Overall Acc": acc.item(),
                "Mean Acc": acc_cls_c[mask].mean().item(), "Mean IoU": iu[mask].mean().item(),
                "Class IoU": {i: (iu[i].item() if mask[i] else "X") for i in range(self.n_classes)},
                "Class Acc": {i: (acc_cls_c[i].item() if mask[i] else "X") for i in range(self.n_classes)}}
