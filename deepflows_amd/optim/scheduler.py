"""LR schedulers — mirror of DeepFlows/optim/scheduler.py:13-61 (host scalar math that
mutates `optimizer.lr`)."""
import math


class LRScheduler:
    def __init__(self, optimizer) -> None:
        self.optimizer = optimizer
        self.last_epoch = -1

    def step(self):
        self.last_epoch += 1


class StepLR(LRScheduler):
    def __init__(self, optimizer, step_size: int, gamma: float = 0.1) -> None:
        super().__init__(optimizer)
        self.step_size, self.gamma = step_size, gamma

    def step(self):
        super().step()
        if self.last_epoch != 0 and self.last_epoch % self.step_size == 0 and hasattr(self.optimizer, 'lr'):
            self.optimizer.lr = self.optimizer.lr * self.gamma


class CosineAnnealingLR(LRScheduler):
    def __init__(self, optimizer, T_max: int, eta_min: float = 0.0) -> None:
        super().__init__(optimizer)
        self.T_max, self.eta_min = T_max, eta_min
        self.base_lr = getattr(optimizer, 'lr', None)

    def step(self):
        super().step()
        if self.base_lr is None:
            return
        t = self.last_epoch % self.T_max
        self.optimizer.lr = self.eta_min + (self.base_lr - self.eta_min) * (1 + math.cos(math.pi * t / self.T_max)) / 2


class WarmupCosineLR(LRScheduler):
    def __init__(self, optimizer, warmup_epochs: int, T_max: int, base_lr: float = None,
                 warmup_start_lr: float = 0.0, eta_min: float = 0.0) -> None:
        super().__init__(optimizer)
        self.warmup_epochs, self.T_max, self.eta_min = warmup_epochs, T_max, eta_min
        self.base_lr = base_lr if base_lr is not None else getattr(optimizer, 'lr', None)
        self.warmup_start_lr = warmup_start_lr

    def step(self):
        super().step()
        if self.base_lr is None:
            return
        if self.last_epoch <= self.warmup_epochs and self.warmup_epochs > 0:
            lr = self.warmup_start_lr + (self.base_lr - self.warmup_start_lr) * (self.last_epoch / max(1, self.warmup_epochs))
        else:
            t = max(0, self.last_epoch - self.warmup_epochs)
            lr = self.eta_min + (self.base_lr - self.eta_min) * (1 + math.cos(math.pi * t / max(1, self.T_max))) / 2
        self.optimizer.lr = lr
