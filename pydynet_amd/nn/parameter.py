"""Trainable leaf tensor (surface of pydynet/nn/parameter.py:4-15): shares its array with
the tensor it wraps (copy=False) and pins the dtype so its gradient has the same dtype."""
from ..core import Tensor


class Parameter(Tensor):
    def __init__(self, data: Tensor, requires_grad: bool = True) -> None:
        super().__init__(data=data.data, dtype=data.dtype, device=data.device, copy=False,
                         requires_grad=requires_grad)

    def __repr__(self) -> str:
        dev = "" if self.device.device == "cpu" else f",\ndevice={self.device}"
        return f"Parameter : \n{self.numpy()}{dev}"
