from .optimizer import Optimizer, SGD, Adagrad, Adadelta, Adam
from .lr_scheduler import ExponentialLR, StepLR, MultiStepLR, CosineAnnealingLR
