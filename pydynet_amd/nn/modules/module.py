"""Module registry (surface of pydynet/nn/modules/module.py:9-126): parameters are recorded at
attribute-assignment time and flattened under "child.name"; `train(mode)` also flips the
process-global autograd switch, exactly as the reference does."""
from collections import OrderedDict

from ..parameter import Parameter
from ...core import Tensor
from ...autograd import set_grad_enabled
from ...cuda import Device, current_device


class Module:
    def __init__(self) -> None:
        self._train = True
        self.device = Device("cpu")
        self._parameters = OrderedDict()

    def __call__(self, *x, **kw):
        return self.forward(*x, **kw)

    def __setattr__(self, name, value) -> None:
        self.__dict__[name] = value
        if isinstance(value, Parameter):
            self._parameters[name] = value
        if isinstance(value, Module):
            for key, p in value._parameters.items():
                self._parameters[f"{name}.{key}"] = p

    def __repr__(self) -> str:
        kids = [(n, m) for n, m in self.__dict__.items() if isinstance(m, Module)]
        body = "\n".join("{:>10} : {}".format(n, m) for n, m in kids)
        return f"{self.__class__.__name__}(\n{body}\n)"

    def parameters(self):
        for p in self._parameters.values():
            if p.requires_grad:
                yield p

    def named_parameters(self):
        for n, p in self._parameters.items():
            if p.requires_grad:
                yield n, p

    def train(self, mode: bool = True):
        set_grad_enabled(mode)
        self.set_module_state(mode)

    def eval(self):
        return self.train(False)

    def set_module_state(self, mode: bool):
        self._train = mode
        for m in self.__dict__.values():
            if isinstance(m, Module):
                m.set_module_state(mode)

    def forward(self, x):
        raise NotImplementedError

    def to(self, device):
        if not isinstance(device, Device):
            device = Device(device)
        if self.device != device:
            self.move(device)
        return self

    def move(self, device):
        self.device = device
        for v in self.__dict__.values():
            if isinstance(v, Module):
                v.move(device)
            if isinstance(v, Parameter):
                v.to(device)

    def cuda(self):
        return self.to(current_device())

    def hip(self, index=0):
        return self.to(f"hip:{index}")

    def cpu(self):
        return self.to('cpu')


class Sequential(Module):
    def __init__(self, *args) -> None:
        super().__init__()
        self.module_list = []
        items = args[0].items() if len(args) == 1 and isinstance(args[0], OrderedDict) else \
            ((str(i), m) for i, m in enumerate(args))
        for name, module in items:
            setattr(self, name, module)
            self.module_list.append(module)

    def forward(self, x):
        for module in self.module_list:
            x = module(x)
        return x

    def __len__(self):
        return len(self.module_list)


class ModuleList(Module):
    def __init__(self, module_list: list) -> None:
        super().__init__()
        self.module_list = module_list
        for idx, module in enumerate(module_list):
            setattr(self, str(idx), module)

    def __getitem__(self, index): return self.module_list[index]
    def __len__(self): return len(self.module_list)
    def __iter__(self): return iter(self.module_list)

    def append(self, module):
        self.module_list.append(module)
        setattr(self, str(len(self.module_list) - 1), module)

    def index(self, module):
        return self.module_list.index(module)
