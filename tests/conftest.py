import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The float64 references of the GPU tests run on the box's host cores, which a GPU box shares with other tenants: with one
# BLAS thread per core (256) a loaded host turned a 60 s run of `pytest -m gpu` into 9-12 minutes (seen three times in round
# 6, every test uniformly slower).  32 threads are plenty for the largest reference product here.
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, str(min(32, os.cpu_count() or 32)))
try:
    import threadpoolctl
    threadpoolctl.threadpool_limits(limits=min(32, os.cpu_count() or 32))
except Exception:                                    # (already-imported NumPy keeps its pool size without threadpoolctl)
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_present():
    """Asked of the product's own library (pdn_device_count): the test suite needs no PyTorch."""
    try:
        from pydynet_amd import cuda
        return cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def hip():
    """hipnp module bound to GPU 0; fails loudly if the HIP library is not built."""
    from pydynet_amd import hipnp, _lib
    _lib.lib()
    hipnp.set_device(0)
    return hipnp


@pytest.fixture()
def emulated_hip(monkeypatch):
    """hipnp running on host memory against the NumPy emulation of the C ABI (tests/abi_emulator/)."""
    from tests import abi_emulator
    from pydynet_amd import hipnp
    from pydynet_amd.core.tensor import Graph
    abi_emulator.install(monkeypatch)
    Graph.clear()
    yield hipnp
    Graph.clear()


def device_variants(namespace, fn):
    """Register `fn(device)` twice: on the real MI355X (-m gpu) and on the emulated ABI (CPU)."""
    name = fn.__name__.replace("check_", "")

    @pytest.mark.gpu
    def on_gpu(hip):
        from pydynet_amd.core.tensor import Graph
        Graph.clear()
        fn("hip:0")

    def on_emulator(emulated_hip):
        fn("hip:0")

    namespace[f"test_{name}_gpu"] = on_gpu
    namespace[f"test_{name}_emulated"] = on_emulator
