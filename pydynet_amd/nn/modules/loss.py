"""Loss layers (surface of pydynet/nn/modules/loss.py): thin modules over nn/functional.py."""
from .module import Module
from .. import functional as F


class Loss(Module):
    _criterion = None                                   # the functional form a subclass applies

    def __init__(self, reduction='mean') -> None:
        super().__init__()
        assert reduction in {'mean', 'sum'}
        self.reduction = reduction

    def forward(self, y_pred, y_true):
        return type(self)._criterion(y_pred, y_true, self.reduction)


class MSELoss(Loss):
    _criterion = staticmethod(F.mse_loss)


class NLLLoss(Loss):
    _criterion = staticmethod(F.nll_loss)


class CrossEntropyLoss(Loss):
    _criterion = staticmethod(F.cross_entropy_loss)
