"""Process-global autograd switch with `no_grad` / `enable_grad` (surface of pydynet/autograd.py:3-50)."""
import functools

_enabled = [True]


def is_grad_enable() -> bool:
    return _enabled[0]


def set_grad_enabled(mode: bool):
    _enabled[0] = bool(mode)


class _GradMode:
    _target = True

    def __enter__(self):
        self._prev = _enabled[0]
        _enabled[0] = self._target

    def __exit__(self, *exc):
        _enabled[0] = self._prev

    def __call__(self, func):
        cls = type(self)

        @functools.wraps(func)
        def wrapped(*a, **kw):
            with cls():
                return func(*a, **kw)
        return wrapped


class no_grad(_GradMode):
    _target = False


class enable_grad(_GradMode):
    _target = True
