"""ctypes binding of ``libpdnhip.so`` -- the only way the Python front end reaches the GPU.

The prototypes are parsed from ``include/pdn_hip.h`` so the header stays the single source
of truth for the C ABI (a CPU test checks that the shared object exports every declared
symbol).  There is deliberately NO fallback: if the library is missing or fails to load,
:func:`lib` raises, and every HIP-device operation in ``pydynet_amd`` fails loudly.
"""
from __future__ import annotations

import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PDN_LIB") or os.path.join(_HERE, "libpdnhip.so")     # PDN_LIB: another build, for same-box A/B runs
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "pdn_hip.h")

_CTYPE = {
    "int": ctypes.c_int,
    "int64_t": ctypes.c_int64,
    "float": ctypes.c_float,
    "double": ctypes.c_double,
}


def parse_header(path: str = HEADER_PATH):
    """Return {name: (restype, [argtypes])} for every function declared in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    protos = {}
    for m in re.finditer(r"(const\s+char\s*\*|int64_t|int)\s+(pdn_\w+)\s*\(([^)]*)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        restype = ctypes.c_char_p if "char" in ret else _CTYPE[ret]
        argtypes = []
        args = args.strip()
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    base = a.replace("const", "").split()[0]
                    argtypes.append(_CTYPE[base])
        protos[name] = (restype, argtypes)
    return protos


class HipLibraryError(RuntimeError):
    """A `pdn_*` entry point returned non-zero; `code` is that status (negative PDN_E*, positive hipError_t)."""

    def __init__(self, message, code=None):
        super().__init__(message)
        self.code = code


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C pydynet_amd/csrc`). The HIP backend has no CPU fallback.")
        # The library owns its device runtime (allocator, copies, streams, RCCL): nothing but the
        # ROCm HIP runtime it is linked against is needed.  RTLD_GLOBAL so that a HIP runtime loaded
        # later by another package in the same process resolves to the same libamdhip64.
        self.cdll = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        self.protos = parse_header()
        self.fn = {}
        for name, (restype, argtypes) in self.protos.items():
            f = getattr(self.cdll, name)  # AttributeError if the .so lacks a declared symbol
            f.restype = restype
            f.argtypes = argtypes
            self.fn[name] = f
        self._last_error = self.fn["pdn_last_error"]
        self.free = self.fn["pdn_free"]              # hot: called from _Buffer.__del__

    def call(self, name, *args):
        rc = self.fn[name](*args)
        if rc != 0:
            msg = self._last_error()
            raise HipLibraryError(f"{name} failed (code {rc}): {msg.decode() if msg else ''}", rc)

    def query(self, name, *args):
        """For functions that return a value (workspace sizes, counts) rather than a status."""
        return self.fn[name](*args)


_LIB = None


def lib() -> _Lib:
    global _LIB
    if _LIB is None:
        _LIB = _Lib()
    return _LIB


def is_built() -> bool:
    return os.path.exists(LIB_PATH)
