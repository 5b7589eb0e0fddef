"""Learning-rate schedules (surface of pydynet/optim/lr_scheduler.py): host-side scalar logic
that wraps `optimizer.step` with a step counter."""
import math
import weakref
from functools import wraps


class _LRScheduler:
    def __init__(self, optimizer, last_epoch: int = -1) -> None:
        self.optimizer = optimizer
        self.base_lr = optimizer.lr
        self.last_epoch = last_epoch
        if not getattr(optimizer.step, "_with_counter", False):
            method = optimizer.step
            ref = weakref.ref(optimizer)
            func, cls = method.__func__, method.__self__.__class__

            @wraps(func)
            def wrapper(*a, **kw):
                inst = ref()
                inst._step_count += 1
                return func.__get__(inst, cls)(*a, **kw)

            wrapper._with_counter = True
            optimizer.step = wrapper
        optimizer._step_count = 0
        self._step_count = 0
        self.step()

    def get_lr(self):
        raise NotImplementedError

    def step(self):
        self._step_count += 1
        self.last_epoch += 1
        self.optimizer.lr = self.get_lr()


class ExponentialLR(_LRScheduler):
    def __init__(self, optimizer, gamma=0.1, last_epoch=-1):
        self.gamma = gamma
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        return self.optimizer.lr if self.last_epoch == 0 else self.optimizer.lr * self.gamma


class StepLR(_LRScheduler):
    def __init__(self, optimizer, step_size, gamma=0.1, last_epoch=-1):
        self.step_size, self.gamma = step_size, gamma
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        if self.last_epoch == 0 or self.last_epoch % self.step_size != 0:
            return self.optimizer.lr
        return self.optimizer.lr * self.gamma


class MultiStepLR(_LRScheduler):
    def __init__(self, optimizer, milestones, gamma=0.1, last_epoch=-1):
        self.milestones, self.gamma = set(milestones), gamma
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        return self.optimizer.lr * self.gamma if self.last_epoch in self.milestones else self.optimizer.lr


class CosineAnnealingLR(_LRScheduler):
    def __init__(self, optimizer, T_max, eta_min=0., last_epoch=-1):
        self.T_max, self.eta_min = T_max, eta_min
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        return self.eta_min + (self.base_lr - self.eta_min) * (1 + math.cos(math.pi * self.last_epoch / self.T_max)) / 2
