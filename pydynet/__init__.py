"""Drop-in alias: `import pydynet` resolves to the MI355X backend (pydynet_amd).

Put this repository on PYTHONPATH ahead of the reference and its programs (examples/,
llm/llama) run unchanged: `pydynet.Tensor`, `pydynet.nn`, `pydynet.nn.functional`,
`pydynet.optim`, `pydynet.core.tensor`, `pydynet.cuda` ... are the pydynet_amd modules.
"""
import importlib
import sys

import pydynet_amd as _impl
from pydynet_amd import *  # noqa: F401,F403
from pydynet_amd import __all__  # noqa: F401

for _name in ("core", "core.tensor", "core.function", "core.fused", "nn", "nn.functional", "nn.init",
              "nn.parameter", "nn.modules", "nn.modules.module", "nn.modules.linear", "nn.modules.conv",
              "nn.modules.norm", "nn.modules.rnn", "nn.modules.activation", "nn.modules.dropout",
              "nn.modules.loss", "nn.modules.pool", "optim", "optim.optimizer", "optim.lr_scheduler",
              "cuda", "autograd", "special", "distributed", "hipnp"):
    sys.modules[f"pydynet.{_name}"] = importlib.import_module(f"pydynet_amd.{_name}")

core, nn, optim, cuda, autograd, special = _impl.core, _impl.nn, _impl.optim, _impl.cuda, _impl.autograd, _impl.special
