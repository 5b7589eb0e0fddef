"""The binding of INTEGRATION.md section 2, runnable: ctypes + NumPy only (no PyTorch, no CuPy, not even this
package) drive libpdnhip.so -- select the GPU, allocate from the caching allocator, copy in, one MFMA GEMM,
copy out.  `python tools/integration_snippet.py` prints the max error against NumPy."""
import ctypes, os, sys
import numpy as np

lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pydynet_amd", "libpdnhip.so"))
vp, i64, c_int = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
lib.pdn_last_error.restype = ctypes.c_char_p
lib.pdn_malloc.argtypes, lib.pdn_free.argtypes = [ctypes.POINTER(vp), i64], [vp]
lib.pdn_memcpy_h2d.argtypes = lib.pdn_memcpy_d2h.argtypes = [vp, vp, i64, vp]
lib.pdn_compute_stream.argtypes = [ctypes.POINTER(vp)]
lib.pdn_gemm_f32.argtypes = [c_int] * 3 + [ctypes.c_float, vp, i64, i64, vp, i64, i64, ctypes.c_float, vp,
    i64, vp, c_int, c_int] + [i64] * 6 + [vp, vp, c_int, vp, i64, vp]


def check(rc):
    if rc: raise RuntimeError(lib.pdn_last_error().decode())


check(lib.pdn_set_device(0))                                  # cp.cuda.Device(0).use(), cuda.py:93-99
stream = vp(); check(lib.pdn_compute_stream(ctypes.byref(stream)))


def to_device(a):                                             # xp.asarray(host), tensor.py:395-403
    p = vp(); check(lib.pdn_malloc(ctypes.byref(p), a.nbytes))
    check(lib.pdn_memcpy_h2d(p, a.ctypes.data, a.nbytes, stream)); return p


def matmul(a_dev, b_dev, M, K, N):                            # x.data @ y.data, tensor.py:659
    c = vp(); check(lib.pdn_malloc(ctypes.byref(c), 4 * M * N))
    check(lib.pdn_gemm_f32(M, N, K, 1.0, a_dev, K, 1, b_dev, N, 1, 0.0, c, N, None, 1, 1,
                           0, 0, 0, 0, 0, 0, None, None, 0, None, 0, stream))
    return c


def to_host(p, shape):                                        # `.get()`, tensor.py:385-390
    out = np.empty(shape, np.float32)
    check(lib.pdn_memcpy_d2h(out.ctypes.data, p, out.nbytes, stream)); return out


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    a, b = rng.standard_normal((300, 288), dtype=np.float32), rng.standard_normal((288, 520), dtype=np.float32)
    pa, pb = to_device(a), to_device(b)
    c = to_host(matmul(pa, pb, 300, 288, 520), (300, 520))
    err = float(np.abs(c - a.astype(np.float64) @ b.astype(np.float64)).max())
    assert "torch" not in sys.modules and "pydynet_amd" not in sys.modules
    print(f"max |err| vs float64 NumPy: {err:.3e}  (torch loaded: {'torch' in sys.modules})")
    assert err < 1e-3
