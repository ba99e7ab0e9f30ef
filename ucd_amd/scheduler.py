"""Learning-rate schedules of the reference (utils/scheduler.py:3-10)."""
from torch.optim.lr_scheduler import LRScheduler, StepLR  # noqa: F401


class PolyLR(LRScheduler):
    """``lr = base_lr * (1 - iteration / max_iters) ** power``, stepped once per training iteration
    (train.py:150-151)."""

    def __init__(self, optimizer, max_iters, power=0.9, last_epoch=-1):
        self.power, self.max_iters = power, max_iters
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        decay = (1 - self.last_epoch / self.max_iters) ** self.power
        return [base_lr * decay for base_lr in self.base_lrs]
