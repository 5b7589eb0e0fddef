"""ORACLE -- CPU (NumPy) restatement of the reference's algorithm for the Tensor-op hot path.

TEST INFRASTRUCTURE ONLY.  Importers allowed: `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg.  The product package `pydynet_amd` never imports it and has
no CPU fallback for HIP tensors.

Pinned against the real reference (imported from /root/reference in the build container) by
`tools/gen_golden.py` -> `tests/golden/*.npz`, checked in `tests/test_oracle_cpu.py`.
"""
from . import tape, nn, llama  # noqa: F401
