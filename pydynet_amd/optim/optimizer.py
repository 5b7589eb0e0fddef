"""Optimizers (surface of pydynet/optim/optimizer.py).  They update the raw `param.data`
arrays in place, as the reference does.  `Adam` on HIP parameters is ONE multi-tensor kernel
launch per step (the reference runs 8 array expressions per parameter, optimizer.py:185-196);
SGD / Adagrad / Adadelta are plain array expressions through `xp`."""
from math import sqrt

import numpy as np

from ..core import Tensor


class Optimizer:
    def __init__(self, params) -> None:
        self.params = list(params)
        self._flat_grad = None
        # data parallel: gradients arrive as the SUM over ranks and every optimizer applies
        # `grad_scale` = 1 / world_size (Adam folds it into its kernel)
        self.grad_scale = 1.0

    def _grad(self, p, weight_decay):
        g = p.grad * self.grad_scale if self.grad_scale != 1.0 else p.grad
        return g + weight_decay * p.data

    def step(self):
        raise NotImplementedError

    def zero_grad(self):
        if self._flat_grad is not None and self._flat_ok():
            with self.params[0].device:
                self._flat_grad[...] = 0.          # one fill instead of one per parameter
            return
        for p in self.params:
            p.zero_grad()

    def flatten_grads(self):
        """Move all gradients into one flat buffer (see optim/flat.py); optional, HIP or CPU."""
        from .flat import flatten_gradients
        self._flat_grad, self._flat_offsets = flatten_gradients(self.params)
        self._flat_views = [p.grad for p in self.params]
        return self._flat_grad

    def _flat_ok(self):
        # user code may have re-assigned p.grad; fall back to per-parameter zeroing then
        for p, v in zip(self.params, self._flat_views):
            if p.grad is not v:
                return False
        return True

    def _state(self):
        out = []
        for p in self.params:
            with p.device:
                out.append(p.xp.zeros(p.shape, dtype=p.dtype))
        return out


class SGD(Optimizer):
    """Momentum SGD with optional Nesterov look-ahead (optimizer.py:32-74)."""

    def __init__(self, params, lr, momentum=.5, weight_decay=0., nesterov=True) -> None:
        super().__init__(params)
        self.lr, self.momentum, self.weight_decay, self.nesterov = lr, momentum, weight_decay, nesterov
        self.v = self._state()

    def step(self):
        for p, v in zip(self.params, self.v):
            with p.device:
                grad = self._grad(p, self.weight_decay)
                v *= self.momentum
                v += self.lr * grad
                p.data -= v
                if self.nesterov:
                    p.data -= self.lr * grad


class Adagrad(Optimizer):
    """optimizer.py:77-112 (eps inside the square root)."""

    def __init__(self, params, lr=1e-2, weight_decay=0, eps=1e-10) -> None:
        super().__init__(params)
        self.lr, self.weight_decay, self.eps = lr, weight_decay, eps
        self.G = self._state()

    def step(self):
        for p, G in zip(self.params, self.G):
            with p.device:
                grad = self._grad(p, self.weight_decay)
                G += grad ** 2
                p.data -= self.lr * grad / (self.eps + G) ** 0.5


class Adadelta(Optimizer):
    """optimizer.py:115-157 (as written there: an RMSprop-style update)."""

    def __init__(self, params, lr=1.0, rho=0.9, weight_decay=0, eps=1e-6) -> None:
        super().__init__(params)
        self.lr, self.rho, self.eps, self.weight_decay = lr, rho, eps, weight_decay
        self.G = self._state()

    def step(self):
        for i, p in enumerate(self.params):
            with p.device:
                grad = self._grad(p, self.weight_decay)
                self.G[i] = self.rho * self.G[i] + (1 - self.rho) * grad ** 2
                p.data -= self.lr * grad / (self.G[i] + self.eps) ** 0.5


class Adam(Optimizer):
    """Adam with the reference's exact arithmetic (optimizer.py:160-196): step counter starts at
    1, a_t = sqrt(1-b2^t)/(1-b1^t) is a host scalar, and eps is added to sqrt(v) WITHOUT the
    bias-correction divisor (this differs from PyTorch and is kept)."""

    CHUNK = 16384

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0) -> None:
        super().__init__(params)
        self.lr = lr
        self.beta1, self.beta2 = betas
        self.eps, self.weight_decay = eps, weight_decay
        self.m, self.v = self._state(), self._state()
        self.t = 1
        self._table = None
        self._table_key = None

    # -- HIP multi-tensor path ------------------------------------------------------------
    def _hip_params(self):
        return [i for i, p in enumerate(self.params)
                if p.device.is_hip and p.dtype == np.float32 and p.grad is not None
                and p.grad.dtype == np.float32 and p.data.is_contiguous() and p.grad.is_contiguous()]

    def _chunk_table(self, idx):
        key = tuple((self.params[i].data._ptr, self.params[i].grad._ptr) for i in idx)
        if key != self._table_key:
            from .. import hipnp
            rows = []
            for i in idx:
                p = self.params[i]
                for off in range(0, p.size, self.CHUNK):
                    n = min(self.CHUNK, p.size - off)
                    rows.append((p.data._ptr + 4 * off, p.grad._ptr + 4 * off,
                                 self.m[i]._ptr + 4 * off, self.v[i]._ptr + 4 * off, n))
            self._table = hipnp.from_numpy(np.asarray(rows, dtype=np.int64).reshape(-1, 5))
            self._table_key = key
        return self._table

    def step(self):
        a_t = sqrt(1 - self.beta2 ** self.t) / (1 - self.beta1 ** self.t)
        fast = self._hip_params() if self.params and self.params[0].device.is_hip else []
        if fast:
            from .. import hipnp, _lib
            with self.params[fast[0]].device:
                table = self._chunk_table(fast)
                graph = hipnp.capturing()
                if graph is not None:
                    graph.pin(table)                     # the captured launch holds the table's address
                    # replayed steps cannot take a host scalar: {t, lr} live on the device, a one-thread
                    # kernel forms lr * a_t there and advances t (the host counter follows at every replay)
                    if len(fast) != len(self.params):
                        raise RuntimeError("Adam inside a hipnp.Graph needs all parameters on the fused HIP path")
                    if graph.warming:                    # the eager run inside the graph's pool: seed the device state
                        self._tick = (hipnp.from_numpy(np.array([float(self.t), float(self.lr)])),
                                      hipnp.empty((1,), np.float32))
                    else:
                        def advance(opt=self):
                            opt.t += 1
                        graph.on_replay(advance)
                        self.t -= 1                      # the captured run executes nothing; replay() counts it
                    _lib.lib().call("pdn_adam_multi_tick_f32", table._ptr, table.shape[0], self._tick[0]._ptr,
                                    self._tick[1]._ptr, self.beta1, self.beta2, self.eps, self.weight_decay,
                                    self.grad_scale, hipnp.stream())
                else:
                    _lib.lib().call("pdn_adam_multi_f32", table._ptr, table.shape[0], self.lr * a_t,
                                    self.beta1, self.beta2, 1 - self.beta1, 1 - self.beta2, self.eps,
                                    self.weight_decay, self.grad_scale, hipnp.stream())
        done = set(fast)
        for i, p in enumerate(self.params):
            if i in done:
                continue
            with p.device:
                grad = self._grad(p, self.weight_decay)
                self.m[i] *= self.beta1
                self.m[i] += (1 - self.beta1) * grad
                self.v[i] *= self.beta2
                self.v[i] += (1 - self.beta2) * grad ** 2
                p.data -= self.lr * a_t * self.m[i] / (self.v[i] ** 0.5 + self.eps)
        self.t += 1
