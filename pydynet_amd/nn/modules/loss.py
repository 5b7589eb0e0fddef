"""Loss layers (surface of pydynet/nn/modules/loss.py)."""
from .module import Module
from .. import functional as F


class Loss(Module):
    def __init__(self, reduction='mean') -> None:
        super().__init__()
        assert reduction in {'mean', 'sum'}
        self.reduction = reduction


class MSELoss(Loss):
    def forward(self, y_pred, y_true): return F.mse_loss(y_pred, y_true, self.reduction)


class NLLLoss(Loss):
    def forward(self, y_pred, y_true): return F.nll_loss(y_pred, y_true, self.reduction)


class CrossEntropyLoss(Loss):
    def forward(self, y_pred, y_true): return F.cross_entropy_loss(y_pred, y_true, self.reduction)
