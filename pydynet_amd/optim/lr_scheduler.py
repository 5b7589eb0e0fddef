"""Learning-rate schedules with the reference's semantics (pydynet/optim/lr_scheduler.py): host-side
scalar logic that wraps `optimizer.step` with a step counter; pinned by tests/test_misc_layers.py."""
import math
import weakref
from collections import Counter
from functools import wraps


class _LRScheduler:
    """Reference protocol (optim/lr_scheduler.py:16-88): `optimizer.initial_lr` is recorded, `optimizer.step`
    is wrapped with a counter, the constructor performs the first `step()`; every `step()` advances
    `last_epoch`, remembers the optimizer's current rate as `_last_lr` and installs `get_lr()`.
    The update rules below are the reference's own -- each is applied to the CURRENT rate, so the
    exponential and step schedules compound (lr_t = lr_{t-1} * gamma**t): parity keeps that."""

    def __init__(self, optimizer, last_epoch: int = -1) -> None:
        self.optimizer = optimizer
        self.last_epoch = last_epoch
        if last_epoch == -1:
            optimizer.initial_lr = optimizer.lr
        else:
            assert hasattr(optimizer, "initial_lr"), "last_epoch=1 but no 'initial_lr' attribute in optimizer!"
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

    def get_last_lr(self):
        return self._last_lr

    def step(self):
        self._step_count += 1
        self.last_epoch += 1
        lr = self.get_lr()
        self._last_lr = self.optimizer.lr
        self.optimizer.lr = lr


class ExponentialLR(_LRScheduler):
    def __init__(self, optimizer, gamma=0.1, last_epoch=-1):
        self.gamma = gamma
        super().__init__(optimizer, last_epoch)

    def get_lr(self):                                           # lr_scheduler.py:100-101
        return self.optimizer.lr * self.gamma ** self.last_epoch


class StepLR(_LRScheduler):
    def __init__(self, optimizer, step_size, gamma=0.1, last_epoch=-1):
        self.step_size, self.gamma = step_size, gamma
        super().__init__(optimizer, last_epoch)

    def get_lr(self):                                           # :116-118
        return self.optimizer.lr * self.gamma ** (self.last_epoch // self.step_size)


class MultiStepLR(_LRScheduler):
    def __init__(self, optimizer, milestones, gamma=0.1, last_epoch=-1):
        self.milestones, self.gamma = Counter(milestones), gamma   # a repeated milestone counts twice
        super().__init__(optimizer, last_epoch)

    def get_lr(self):                                           # :133-136
        if self.last_epoch not in self.milestones:
            return self.optimizer.lr
        return self.optimizer.lr * self.gamma ** self.milestones[self.last_epoch]


class CosineAnnealingLR(_LRScheduler):
    def __init__(self, optimizer, T_max, eta_min=0., last_epoch=-1):
        self.T_max, self.eta_min = T_max, eta_min
        super().__init__(optimizer, last_epoch)

    def get_lr(self):                                           # recursive form of :150-160
        base, t, T = self.optimizer.initial_lr, self.last_epoch, self.T_max
        if t == 0:
            return base
        if (t - 1 - T) % (2 * T) == 0:
            return self.get_last_lr() + (base - self.eta_min) * (1 - math.cos(math.pi / T)) / 2
        return ((1 + math.cos(math.pi * t / T)) / (1 + math.cos(math.pi * (t - 1) / T))
                * (self.get_last_lr() - self.eta_min) + self.eta_min)
