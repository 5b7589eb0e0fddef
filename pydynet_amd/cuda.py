"""Device seam of the framework (replaces pydynet/cuda.py:16-99).

`Device.xp` is the array module every operator computes with: `numpy` for "cpu" (the
reference's own CPU device, kept for host-side initialisation and the CPU-only configs) and
`pydynet_amd.hipnp` -- HBM arrays driven by the hand-written HIP kernels -- for an MI355X.
"cuda", "cuda:N", "hip", "hip:N" and a bare int all name GPU N, so reference scripts that say
`.to('cuda')` run on the HIP backend unchanged.  There is no fallback between the two: a GPU
device can only be constructed when libpdnhip.so loads and a GPU is visible.
"""
import numpy as np

from . import _lib


def is_available() -> bool:
    """True when the HIP library is built AND at least one GPU is visible."""
    if not _lib.is_built():
        return False
    try:
        return _lib.lib().query("pdn_device_count") > 0
    except Exception:
        return False


def device_count() -> int:
    return _lib.lib().query("pdn_device_count") if _lib.is_built() else 0


def current_device() -> int:
    from . import hipnp
    return hipnp.current_device()


def set_device(device: int) -> None:
    from . import hipnp
    hipnp.set_device(int(device))


_previous = []          # stack of GPUs to return to when `with device:` blocks exit (one thread per process)


class Device:
    __slots__ = ("device", "device_id")

    def __init__(self, device=None) -> None:
        self.device_id = None
        if isinstance(device, Device):
            self.device, self.device_id = device.device, device.device_id
            return
        if device is None or device == "cpu":
            self.device = "cpu"
            return
        if isinstance(device, str):
            kind, _, idx = device.partition(":")
            if kind not in ("cuda", "hip"):
                raise ValueError(f'Unknown device "{device}"!')
            if idx == "":
                idx = "0"
            if not idx.isdigit():
                raise ValueError(f'Wrong cuda id "{idx}"!')
            self.device_id = int(idx)
        elif isinstance(device, (int, np.integer)) and not isinstance(device, bool):
            self.device_id = int(device)
        else:
            raise ValueError(f'Unknown device "{device}"!')
        n = device_count()
        if n == 0:
            raise RuntimeError("HIP device is not supported on this system "
                               "(libpdnhip.so missing or no GPU visible).")
        if self.device_id >= n:
            raise ValueError(f"device id {self.device_id} out of range ({n} GPU(s) visible)")
        self.device = "hip"

    @property
    def is_hip(self) -> bool:
        return self.device == "hip"

    @property
    def xp(self):
        if self.device == "cpu":
            return np
        from . import hipnp
        return hipnp

    def __repr__(self) -> str:
        if self.device == "cpu":
            return "Device(type='cpu')"
        return f"Device(type='hip', index={self.device_id})"

    def __str__(self) -> str:
        return "cpu" if self.device == "cpu" else f"hip:{self.device_id}"

    def __eq__(self, other) -> bool:
        if not isinstance(other, Device):
            try:
                other = Device(other)
            except (ValueError, RuntimeError):
                return False
        return self.device == other.device and self.device_id == other.device_id

    def __hash__(self):
        return hash((self.device, self.device_id))

    # `with device:` makes it the current GPU for allocations (cuda.py:93-99)
    def __enter__(self):
        prev = None
        if self.device == "hip":
            from . import hipnp
            cur = hipnp.current_device()
            if cur != self.device_id:
                prev = cur
                hipnp.set_device(self.device_id)
        _previous.append(prev)
        return self

    def __exit__(self, *exc):
        prev = _previous.pop() if _previous else None
        if prev is not None:
            from . import hipnp
            hipnp.set_device(prev)
        return False


def __getattr__(name):
    # the reference exposes a module-level flag set at import (pydynet/cuda.py:8-12); here it is
    # evaluated on first use so that importing the package never touches the driver
    if name == "cuda_available":
        return is_available()
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
