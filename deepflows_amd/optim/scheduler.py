"""Learning-rate schedules with the reference's interface (DeepFlows/optim/scheduler.py:13-61: construct over an
optimizer, call step() once per epoch, the schedule writes `optimizer.lr`).

Each schedule is a closed form `_lr_at(epoch)` of the epoch counter; `step()` advances the counter and stores what
the closed form returns (None = leave the learning rate alone).  The sequences are pinned by values the reference's
own classes produced (oracle/gen_f3_golden.py -> tests/golden/g11_nn_surface.npz, tests/test_frontend_host_logic.py).
"""
import math


def _half_cosine(low, high, position, period):
    """`high` at position 0 falling to `low` at position == period along half a cosine wave."""
    return low + (high - low) * (1 + math.cos(math.pi * position / period)) / 2


class LRScheduler:
    def __init__(self, optimizer) -> None:
        self.optimizer = optimizer
        self.last_epoch = -1

    def _lr_at(self, epoch):
        return None

    def step(self):
        self.last_epoch += 1
        lr = self._lr_at(self.last_epoch)
        if lr is not None and hasattr(self.optimizer, "lr"):
            self.optimizer.lr = lr


class StepLR(LRScheduler):
    """The optimizer's CURRENT rate times `gamma` at every epoch that is a positive multiple of `step_size`
    (multiplicative on whatever the rate is then, as the reference: a rate changed by hand stays changed)."""

    def __init__(self, optimizer, step_size: int, gamma: float = 0.1) -> None:
        super().__init__(optimizer)
        self.step_size = step_size
        self.gamma = gamma

    def _lr_at(self, epoch):
        decays = epoch > 0 and epoch % self.step_size == 0
        return self.optimizer.lr * self.gamma if decays and hasattr(self.optimizer, "lr") else None


class CosineAnnealingLR(LRScheduler):
    """base -> eta_min along half a cosine of `T_max` epochs, restarting at base every T_max epochs."""

    def __init__(self, optimizer, T_max: int, eta_min: float = 0.0) -> None:
        super().__init__(optimizer)
        self.T_max = T_max
        self.eta_min = eta_min
        self.base_lr = getattr(optimizer, "lr", None)

    def _lr_at(self, epoch):
        if self.base_lr is None:
            return None
        return _half_cosine(self.eta_min, self.base_lr, epoch % self.T_max, self.T_max)


class WarmupCosineLR(LRScheduler):
    """Linear warm-up from `warmup_start_lr` to `base_lr` over `warmup_epochs` (epoch == warmup_epochs is the last
    warm-up epoch and sits at base_lr), then half a cosine of `T_max` epochs down to `eta_min` — not periodic: past
    T_max the cosine keeps going, as in the reference."""

    def __init__(self, optimizer, warmup_epochs: int, T_max: int, base_lr: float = None,
                 warmup_start_lr: float = 0.0, eta_min: float = 0.0) -> None:
        super().__init__(optimizer)
        self.warmup_epochs = warmup_epochs
        self.T_max = T_max
        self.eta_min = eta_min
        self.base_lr = getattr(optimizer, "lr", None) if base_lr is None else base_lr
        self.warmup_start_lr = warmup_start_lr

    def _lr_at(self, epoch):
        if self.base_lr is None:
            return None
        if 0 < self.warmup_epochs and epoch <= self.warmup_epochs:
            return self.warmup_start_lr + (self.base_lr - self.warmup_start_lr) * (epoch / self.warmup_epochs)
        return _half_cosine(self.eta_min, self.base_lr, max(0, epoch - self.warmup_epochs), max(1, self.T_max))
