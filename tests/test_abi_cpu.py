"""C-ABI boundary checks that need no GPU: the shared object loads, exports every symbol the
header declares (and nothing pdn_* beyond it), follows the error convention, and the product
fails LOUDLY -- never falls back to the CPU -- when a HIP device is requested without a GPU."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from pydynet_amd import _lib

NO_GPU = _lib.lib().query("pdn_device_count") == 0


def test_library_exports_exactly_the_declared_abi():
    protos = _lib.parse_header()
    assert len(protos) >= 36
    cdll = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(cdll, name), f"{name} declared in include/pdn_hip.h but not exported"
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in nm.splitlines() if " T " in ln and ln.split()[-1].startswith("pdn_")}
    assert exported == set(protos), exported ^ set(protos)


def test_header_cites_reference_for_every_entry_point_group():
    text = open(_lib.HEADER_PATH).read()
    for cite in ["tensor.py:659", "tensor.py:701", "nn/functional.py:43-49", "norm.py:245-248",
                 "llm/llama/model.py:23-44", "tensor.py:937-940", "nn/functional.py:364-381",
                 "optim/optimizer.py:185-196", "nn/functional.py:194-339", "pydynet/cuda.py:89-91"]:
        assert cite in text, cite


def test_error_convention_without_launching():
    L = _lib.lib()
    assert L.query("pdn_abi_version") == 1
    rc = L.fn["pdn_gemm_f32"](-1, 4, 4, 1.0, None, 1, 1, None, 1, 1, 0.0, None, 4, None, 1, 1,
                              0, 0, 0, 0, 0, 0, None, None, 0, None, 0, None)
    assert rc == -1 and b"negative" in L.fn["pdn_last_error"]()
    with pytest.raises(_lib.HipLibraryError, match="pdn_gemm_f32"):
        L.call("pdn_gemm_f32", 4, 4, 4, 1.0, None, 1, 1, None, 1, 1, 0.0, None, 4, None, 1, 1,
               0, 0, 0, 0, 0, 0, None, None, 0, None, 0, None)       # null operands
    shape = (ctypes.c_int64 * 1)(4)
    rc = L.fn["pdn_ew_binary"](0, 99, 0, 1, shape, 8, shape, 8, shape, 0.0, 8, shape, None)
    assert rc == -1                                    # bad op code rejected before any launch
    assert L.fn["pdn_reduce"](0, 77, 0, None, None, None, None, None, None, 0, None) == -1


@pytest.mark.skipif(not NO_GPU, reason="needs a GPU-less host")
def test_no_silent_cpu_fallback_for_hip_devices():
    import pydynet_amd as pdn
    assert _lib.lib().query("pdn_device_count") == 0
    assert not pdn.cuda.is_available()
    for name in ("cuda", "cuda:0", "hip:0", 0):
        with pytest.raises(RuntimeError):
            pdn.Device(name)
    with pytest.raises(RuntimeError):
        pdn.Tensor(np.ones(3, np.float32), device="cuda")
    with pytest.raises(RuntimeError):
        pdn.nn.Linear(2, 2).to("hip:0")
    with pytest.raises(ValueError):
        pdn.Device("tpu")
    with pytest.raises(ValueError):
        pdn.Device("cuda:x")
    assert pdn.Device("cpu") == "cpu" and pdn.Device(None).xp is np


def test_product_never_imports_the_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for base, _, files in os.walk(os.path.join(root, "pydynet_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(base, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "abi_emulator" not in src, f


def test_integration_snippet_is_the_documented_one_and_needs_no_torch():
    """tools/integration_snippet.py is INTEGRATION.md section 2 made runnable: same binding lines, no torch."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    src = open(os.path.join(root, "tools", "integration_snippet.py")).read()
    assert "import torch" not in src and "import torch" not in doc.split("## 3.")[0]
    for line in ("lib.pdn_malloc.argtypes, lib.pdn_free.argtypes = [ctypes.POINTER(vp), i64], [vp]",
                 "check(lib.pdn_memcpy_h2d(p, a.ctypes.data, a.nbytes, stream)); return p",
                 "0, 0, 0, 0, 0, 0, None, None, 0, None, 0, stream))"):
        assert line in doc and line in src, line
