"""Loss modules — mirror of DeepFlows/nn/modules/loss.py."""
from .. import functional as F
from .module import Module

__all__ = ['NLLLoss', 'MSELoss', 'CrossEntropyLoss']


class _Loss(Module):
    def __init__(self, reduction: str = 'mean') -> None:
        super().__init__()
        assert reduction in {'mean', 'sum'}
        self.reduction = reduction


class NLLLoss(_Loss):
    def forward(self, input, target):
        return F.nll_loss(input, target, reduction=self.reduction)


class MSELoss(_Loss):
    def forward(self, input, target):
        return F.mse_loss(input, target, reduction=self.reduction)


class CrossEntropyLoss(_Loss):
    def forward(self, input, target):
        return F.cross_entropy(input, target, reduction=self.reduction)
