"""Host logic of `hipnp` (broadcast/stride marshalling, views, index parsing, workspace and
argument plumbing) checked in the GPU-less container: the bodies of the GPU kernel tests run
against the NumPy emulation of the C ABI, with their numerical assertions enforced."""
import inspect

import pytest

import tests.test_kernels_gpu as kernel_tests


def _cases(fn):
    marks = [m for m in getattr(fn, "pytestmark", []) if m.name == "parametrize"]
    if not marks:
        return [{}]
    m = marks[0]
    names = [n.strip() for n in m.args[0].split(",")]
    return [dict(zip(names, v if isinstance(v, tuple) and len(names) > 1 else (v,))) for v in m.args[1]]


_ALL = [(name, kw) for name, fn in sorted(vars(kernel_tests).items())
        if name.startswith("test_") and callable(fn) for kw in _cases(fn)]


@pytest.mark.parametrize("name,kw", _ALL, ids=[f"{n}-{i}" for i, (n, _) in enumerate(_ALL)])
def test_kernel_test_body_on_emulator(emulated_hip, name, kw):
    fn = getattr(kernel_tests, name)
    if "hip" not in inspect.signature(fn).parameters:
        pytest.skip("drives the real library in a subprocess; nothing to emulate")
    fn(emulated_hip, **kw)
