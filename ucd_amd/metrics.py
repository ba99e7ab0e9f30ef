"""Streaming segmentation metrics with the confusion matrix kept on the device.

Mirror of the reference's ``metrics/stream_metrics.py:34-122`` (``StreamSegMetrics``: same methods, same result keys and
arithmetic), which builds the matrix with a numpy ``bincount`` per image on the host after copying the full-resolution
predictions over PCIe (``train.py:242-246``).  Here ``update`` takes device tensors (labels and arg-max predictions of the
whole batch) and adds one ``bincount`` of the batch to a device-resident [n, n] matrix; ``update_from_logits`` goes one
step further and takes the model's LOW-resolution logits: up-sampling, arg-max and histogram are one HIP kernel, the
full-resolution logits never exist.  The host sees numbers only in ``get_results``.  ``synch`` reduces matrix and sample count to rank 0 like the reference.  The matplotlib rendering of the
confusion matrix (``confusion_matrix_to_fig``) is out of scope.
"""
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
        lt, lp = torch.as_tensor(label_trues), torch.as_tensor(label_preds)
        n = self.n_classes
        self.total_samples += int(lt.shape[0]) if lt.dim() > 1 else 1          # images of the batch (stream_metrics.py:47)
        lt, lp = lt.reshape(-1), lp.reshape(-1)
        mask = (lt >= 0) & (lt < n)                                            # stream_metrics.py:65
        idx = n * lt[mask].long() + lp[mask].long()
        hist = torch.bincount(idx, minlength=n * n).reshape(n, n).double()
        self.confusion_matrix = hist if self.confusion_matrix is None else self.confusion_matrix + hist

    def update_from_logits(self, label_trues, sem):
        """``update(labels, upsample(sem).argmax(1))`` without the full-resolution logits: bilinear up-sampling, arg-max and
        the histogram in one kernel (``ucd_seg_confusion``; ``sem`` = the model's low-resolution logits [B, Ctot, h, w])."""
        from . import hip
        B, Ctot, h, w = sem.shape
        H, W = label_trues.shape[-2:]
        n = self.n_classes
        s = sem.detach().permute(0, 2, 3, 1).reshape(B * h * w, Ctot).float().contiguous()
        lab = label_trues.contiguous()
        if lab.dtype != torch.int64:
            lab = lab.long()
        hist = torch.zeros(n, n, dtype=torch.int64, device=sem.device)
        hip._check(hip.load().ucd_seg_confusion(hip.ptr(s), Ctot, hip.ptr(lab), B, H, W, h, w, Ctot, n, hip.ptr(hist), None,
                                                hip.stream()), "ucd_seg_confusion")
        self.total_samples += B
        hist = hist.double()
        self.confusion_matrix = hist if self.confusion_matrix is None else self.confusion_matrix + hist

    def synch(self, device):
        """Sum the matrices and sample counts of all ranks onto rank 0 (stream_metrics.py:110-120)."""
        if dist.is_available() and dist.is_initialized() and self.confusion_matrix is not None:
            cm = self.confusion_matrix.to(device)
            samples = torch.tensor(float(self.total_samples), dtype=torch.float64, device=device)
            dist.reduce(cm, dst=0)
            dist.reduce(samples, dst=0)
            if dist.get_rank() == 0:
                self.confusion_matrix = cm
                self.total_samples = int(samples.item())

    def to_str(self, results):
        s = "\n"
        for k, v in results.items():
            if k not in ("Class IoU", "Class Acc", "Confusion Matrix"):
                s += "%s: %f\n" % (k, v)
        s += "Class IoU:\n" + "".join("\tclass %d: %s\n" % (k, str(v)) for k, v in results["Class IoU"].items())
        s += "Class Acc:\n" + "".join("\tclass %d: %s\n" % (k, str(v)) for k, v in results["Class Acc"].items())
        return s

    def get_results(self):
        hist = self.confusion_matrix.cpu()
        eps = 1e-6
        gt_sum, diag = hist.sum(dim=1), hist.diag()
        mask = gt_sum != 0
        acc = diag.sum() / hist.sum()
        acc_cls_c = diag / (gt_sum + eps)
        iu = diag / (gt_sum + hist.sum(dim=0) - diag + eps)
        freq = gt_sum / hist.sum()
        return {"Total samples": self.total_samples, "Overall Acc": acc.item(),
                "Mean Acc": acc_cls_c[mask].mean().item(), "FreqW Acc": (freq[freq > 0] * iu[freq > 0]).sum().item(),
                "Mean IoU": iu[mask].mean().item(),
                "Class IoU": {i: (iu[i].item() if mask[i] else "X") for i in range(self.n_classes)},
                "Class Acc": {i: (acc_cls_c[i].item() if mask[i] else "X") for i in range(self.n_classes)}}
