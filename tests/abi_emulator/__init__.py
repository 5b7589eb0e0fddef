"""TEST-ONLY host emulation of the pdnhip C ABI.

Lets the whole Python front end (hipnp striding / broadcasting / views / indexing, the tape
engine, nn, optim, data-parallel wrapper) run in the GPU-less container: `hipnp` is pointed
at host memory and every `pdn_*` entry point is answered by a NumPy statement of the same
contract (include/pdn_hip.h).  It is NOT a product path: `pydynet_amd` never imports it, and
on the GPU box the `-m gpu` tests exercise the real library.  Numerics here follow NumPy, so
CPU tests compare against the oracle at fp32 round-off.
"""
from pydynet_amd import _lib
from ._base import _NP, _ints, view, flat  # noqa: F401
from ._runtime import RuntimeMixin
from ._gemm import GemmMixin
from ._pointwise import PointwiseMixin
from ._attention import AttentionMixin
from ._decode import DecodeMixin
from ._recurrent_norm import RecurrentNormMixin
from ._conv import ConvMixin


class EmulatedLib(RuntimeMixin, GemmMixin, PointwiseMixin, AttentionMixin, DecodeMixin, RecurrentNormMixin, ConvMixin):
    """Every `pdn_*` entry point of include/pdn_hip.h, one mixin per kernel family (round 6: split out of one 1670-line file)."""


def install(monkeypatch):
    """Point hipnp at the emulated library (host memory); returns the emulator."""
    from pydynet_amd import hipnp
    emu = EmulatedLib()
    monkeypatch.setattr(_lib, "_LIB", emu)
    monkeypatch.setattr(_lib, "is_built", lambda: True)
    monkeypatch.setattr(hipnp, "_ws", {})
    monkeypatch.setattr(hipnp, "_err", {})
    monkeypatch.setattr(hipnp, "_state", {"device": 0, "stream": 0, "streams": {}})
    # recycled handles of a PREVIOUS emulator instance (pinned read-back slots and their events, the side stream with its
    # fork / join events) mean nothing to this one: a loss read through a stale slot returns whatever that host block
    # held last (seen as a wrong first loss in tests/test_dropin_reference_programs.py after the bench.main() tests)
    from pydynet_amd import _hipnp_host, _hipnp_streams
    monkeypatch.setattr(_hipnp_host.read_later, "_free", {})
    monkeypatch.setattr(_hipnp_streams, "_side", {})
    from pydynet_amd.core import fused
    monkeypatch.setattr(fused.qkv_attention, "_rope_tables", {})
    return emu

