"""CPU dry run of the GPU test bodies: the HIP library is replaced by a no-op stub and device
memory by host memory, so every Python-level code path of `hipnp` (argument marshalling,
broadcasting, views, index parsing) executes here; numerical assertions are expected to fail
and are ignored.  This catches host-side bugs before GPU minutes are spent."""
import inspect

import numpy as np
import pytest


class _StubLib:
    def __init__(self, real_protos):
        self.protos = real_protos
        self.calls = []

    def call(self, name, *args):
        assert name in self.protos, name
        assert len(args) == len(self.protos[name][1]), (name, len(args), len(self.protos[name][1]))
        self.calls.append(name)

    def query(self, name, *args):
        assert name in self.protos, name
        assert len(args) == len(self.protos[name][1]), (name, len(args))
        return 1 << 20


@pytest.fixture()
def fake_hip(monkeypatch):
    from pydynet_amd import hipnp, _lib
    stub = _StubLib(_lib.parse_header())
    monkeypatch.setattr(_lib, "_LIB", stub)
    monkeypatch.setattr(hipnp, "_dev", lambda: "cpu")
    monkeypatch.setattr(hipnp, "_ws", {"buf": None, "bytes": 0})
    monkeypatch.setattr(hipnp, "_err", {"buf": None})
    return hipnp


def _run(fn, hip):
    sig = inspect.signature(fn)
    marks = [m for m in getattr(fn, "pytestmark", []) if m.name == "parametrize"]
    cases = [{}]
    for m in marks:
        names = [n.strip() for n in m.args[0].split(",")]
        cases = [dict(zip(names, v if isinstance(v, tuple) and len(names) > 1 else (v,))) for v in m.args[1]]
    for kw in cases:
        try:
            fn(hip, **kw) if "hip" in sig.parameters else fn(**kw)
        except AssertionError:
            pass  # numerics are meaningless with the stub


def test_gpu_kernel_tests_run_host_side(fake_hip):
    import tests.test_kernels_gpu as mod
    for name, fn in sorted(vars(mod).items()):
        if name.startswith("test_") and callable(fn):
            _run(fn, fake_hip)
    assert "pdn_gemm_f32" in fake_hip._lib.lib().calls
