"""Module registry (surface of pydynet/nn/modules/module.py:9-126): parameters are recorded at
attribute-assignment time and flattened under "child.name"; `train(mode)` also flips the
process-global autograd switch, exactly as the reference does."""
from collections import OrderedDict

from ..parameter import Parameter
from ...core import Tensor  # noqa: F401  (re-exported for `from .module import Tensor` users)
from ...autograd import set_grad_enabled
from ...cuda import Device, current_device


class Module:
    def __init__(self) -> None:
        self._train, self.device, self._parameters = True, Device("cpu"), OrderedDict()

    # -- registration ---------------------------------------------------------------------------------------------
    def __setattr__(self, name, value) -> None:
        object.__setattr__(self, name, value)
        if isinstance(value, Parameter):
            self._parameters[name] = value
        elif isinstance(value, Module):                 # a child's table is flattened into this one AT ASSIGNMENT
            self._parameters.update((f"{name}.{k}", p) for k, p in value._parameters.items())

    def _children(self):
        """(attribute name, sub-module) in assignment order."""
        return [(n, v) for n, v in vars(self).items() if isinstance(v, Module)]

    def named_parameters(self):
        return ((n, p) for n, p in self._parameters.items() if p.requires_grad)

    def parameters(self):
        return (p for _, p in self.named_parameters())

    # -- calling --------------------------------------------------------------------------------------------------
    def __call__(self, *x, **kw):
        return self.forward(*x, **kw)

    def forward(self, x):
        raise NotImplementedError

    def __repr__(self) -> str:
        rows = ["{:>10} : {}".format(n, m) for n, m in self._children()]
        return "{}(\n{}\n)".format(type(self).__name__, "\n".join(rows))

    # -- mode -----------------------------------------------------------------------------------------------------
    def set_module_state(self, mode: bool):
        self._train = mode
        for _, child in self._children():
            child.set_module_state(mode)

    def train(self, mode: bool = True):
        set_grad_enabled(mode)                           # (the reference couples the two: module.py:60-62)
        self.set_module_state(mode)

    def eval(self):
        return self.train(False)

    # -- placement ------------------------------------------------------------------------------------------------
    def move(self, device):
        self.device = device
        for v in vars(self).values():
            if isinstance(v, Module):
                v.move(device)
            if isinstance(v, Parameter):
                v.to(device)

    def to(self, device):
        target = device if isinstance(device, Device) else Device(device)
        if target != self.device:
            self.move(target)
        return self

    def cpu(self):
        return self.to('cpu')

    def cuda(self):
        return self.to(current_device())

    def hip(self, index=0):
        return self.to(f"hip:{index}")


class Sequential(Module):
    def __init__(self, *args) -> None:
        super().__init__()
        named = list(args[0].items()) if len(args) == 1 and isinstance(args[0], OrderedDict) else \
            [(str(i), m) for i, m in enumerate(args)]
        self.module_list = [m for _, m in named]
        for key, m in named:
            setattr(self, key, m)

    def __len__(self):
        return len(self.module_list)

    def forward(self, x):
        out = x
        for layer in self.module_list:
            out = layer(out)
        return out


class ModuleList(Module):
    def __init__(self, module_list: list) -> None:
        super().__init__()
        self.module_list = module_list
        for position, m in enumerate(module_list):
            setattr(self, str(position), m)

    def __iter__(self): return iter(self.module_list)
    def __len__(self): return len(self.module_list)
    def __getitem__(self, index): return self.module_list[index]

    def index(self, module):
        return self.module_list.index(module)

    def append(self, module):
        setattr(self, str(len(self.module_list)), module)
        self.module_list.append(module)
