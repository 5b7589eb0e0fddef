// Strided / broadcasting elementwise kernels for the `xp` facade (gfx950).
//
// Serves every ndarray operator the reference applies to `Tensor.data`:
//   + - * / ** maximum minimum      pydynet/core/tensor.py:548,564,591,612,634,811,820
//   exp log abs sign, unary -       tensor.py:689,786,802,829
//   sigmoid / tanh (overflow-safe piecewise forms)   tensor.py:999-1003,1012-1016
//   comparisons (eq ne lt le gt ge)                  tensor.py:289-316
//   astype / copy / strided assignment / fill        tensor.py:168-177,279,380-383
// All are HBM-bound streams: contiguous same-shape operands take a 16 B/lane vector
// path; anything else (broadcast, transposed views, in-place into a slice) goes through a
// dimension-collapsed strided index walk.
#include "common.h"
#include <stdlib.h>

enum { PDN_F32 = 0, PDN_F64 = 1, PDN_I64 = 2, PDN_BOOL = 3, PDN_I32 = 4, PDN_F16 = 5 };

// float16 STORAGE (the reference's tests draw float16 operands, tests/test_tensor_basic.py:16,80-81):
// + - * / are IEEE half operations (= NumPy's "compute in float32, round once" for a single operation),
// transcendental functions are evaluated in float32 and rounded to half.
typedef _Float16 half_t;

enum {
  BOP_ADD = 0, BOP_SUB, BOP_MUL, BOP_DIV, BOP_POW, BOP_MAX, BOP_MIN,
  BOP_EQ = 16, BOP_NE, BOP_LT, BOP_LE, BOP_GT, BOP_GE,
};
enum {
  UOP_COPY = 0, UOP_NEG, UOP_EXP, UOP_LOG, UOP_ABS, UOP_SIGN, UOP_SQRT, UOP_SQUARE, UOP_RECIP,
  UOP_SIGMOID, UOP_TANH,
};

struct EwDims {
  int ndim;
  int fast;                        // fewer than 2^31 elements: the index walk divides by multiply-high (mg, sh), see ew_offsets
  int64_t shape[PDN_MAX_DIMS];
  int64_t sa[PDN_MAX_DIMS];
  int64_t sb[PDN_MAX_DIMS];
  int64_t so[PDN_MAX_DIMS];
  uint32_t mg[PDN_MAX_DIMS];
  uint32_t sh[PDN_MAX_DIMS];
};

template <typename T> __device__ __forceinline__ T t_exp(T x);
template <> __device__ __forceinline__ float t_exp<float>(float x) { return expf(x); }
template <> __device__ __forceinline__ double t_exp<double>(double x) { return exp(x); }
template <> __device__ __forceinline__ half_t t_exp<half_t>(half_t x) { return (half_t)expf((float)x); }
template <typename T> __device__ __forceinline__ T t_log(T x);
template <> __device__ __forceinline__ half_t t_log<half_t>(half_t x) { return (half_t)logf((float)x); }
template <> __device__ __forceinline__ float t_log<float>(float x) { return logf(x); }
template <> __device__ __forceinline__ double t_log<double>(double x) { return log(x); }
template <typename T> __device__ __forceinline__ T t_pow(T x, T y);
template <> __device__ __forceinline__ float t_pow<float>(float x, float y) {
  if (y == 0.5f) return sqrtf(x);  // numpy's pow(x, 0.5) fast path is correctly rounded too
  if (y == 2.0f) return x * x;
  return powf(x, y);
}
template <> __device__ __forceinline__ double t_pow<double>(double x, double y) {
  if (y == 0.5) return sqrt(x);
  if (y == 2.0) return x * x;
  return pow(x, y);
}
template <> __device__ __forceinline__ half_t t_pow<half_t>(half_t x, half_t y) {
  return (half_t)t_pow<float>((float)x, (float)y);
}
template <typename T> __device__ __forceinline__ T t_sqrt(T x);
template <> __device__ __forceinline__ half_t t_sqrt<half_t>(half_t x) { return (half_t)sqrtf((float)x); }
template <> __device__ __forceinline__ float t_sqrt<float>(float x) { return sqrtf(x); }
template <> __device__ __forceinline__ double t_sqrt<double>(double x) { return sqrt(x); }

template <typename T, int OP>
__device__ __forceinline__ T bin_apply(T a, T b) {
  if constexpr (OP == BOP_ADD) return a + b;
  if constexpr (OP == BOP_SUB) return a - b;
  if constexpr (OP == BOP_MUL) return a * b;
  if constexpr (OP == BOP_DIV) return a / b;
  if constexpr (OP == BOP_POW) return t_pow<T>(a, b);
  // numpy maximum/minimum propagate NaN
  if constexpr (OP == BOP_MAX) return (a != a) ? a : ((b != b) ? b : (a > b ? a : b));
  if constexpr (OP == BOP_MIN) return (a != a) ? a : ((b != b) ? b : (a < b ? a : b));
  return a;
}
template <typename T, int OP>
__device__ __forceinline__ bool cmp_apply(T a, T b) {
  if constexpr (OP == BOP_EQ) return a == b;
  if constexpr (OP == BOP_NE) return a != b;
  if constexpr (OP == BOP_LT) return a < b;
  if constexpr (OP == BOP_LE) return a <= b;
  if constexpr (OP == BOP_GT) return a > b;
  return a >= b;
}
template <typename T, int OP>
__device__ __forceinline__ T un_apply(T x) {
  if constexpr (OP == UOP_COPY) return x;
  if constexpr (OP == UOP_NEG) return -x;
  if constexpr (OP == UOP_EXP) return t_exp<T>(x);
  if constexpr (OP == UOP_LOG) return t_log<T>(x);
  if constexpr (OP == UOP_ABS) return x < (T)0 ? -x : x;
  if constexpr (OP == UOP_SIGN) return (x != x) ? x : (T)((x > (T)0) - (x < (T)0));
  if constexpr (OP == UOP_SQRT) return t_sqrt<T>(x);
  if constexpr (OP == UOP_SQUARE) return x * x;
  if constexpr (OP == UOP_RECIP) return (T)1 / x;
  // reference piecewise forms, tensor.py:1000-1002 / 1013-1015
  if constexpr (OP == UOP_SIGMOID)
    return x > (T)0 ? (T)1 / ((T)1 + t_exp<T>(-x)) : (T)1 - (T)1 / ((T)1 + t_exp<T>(x));
  if constexpr (OP == UOP_TANH)
    return x > (T)0 ? (T)2 / ((T)1 + t_exp<T>((T)-2 * x)) - (T)1
                    : (T)1 - (T)2 / ((T)1 + t_exp<T>((T)2 * x));
  return x;
}

__device__ __forceinline__ void ew_offsets(const EwDims& d, int64_t i, int64_t& oa, int64_t& ob,
                                           int64_t& oo) {
  oa = ob = oo = 0;
  if (d.fast) {
    // gfx950 has no integer divide: a 64-bit i / s is an ~80-instruction routine per dimension and element, which bounded
    // these kernels at 1.4-2.4 TB/s (a RoPE product of the plain-operator Llama: 57 us for 113 MB; the transposed copy in
    // front of its score product 97 us for 151 MB).  For n < 2^31 and s >= 2 (size-1 dimensions are dropped):
    // n / s = umulhi(n, ceil(2^(31 + l) / s)) >> (l - 1), l = ceil(log2 s) -- exact, the error term n e / (s 2^p) < 1 / s.
    // (Tried and dropped, tools/ew_strided_probe.py under rocprofv3: a walk of compile-time depth over right-aligned slots
    // -- slower, 33 -> 44 us on a two-dimensional add -- and 24-bit multiply-adds for the offsets -- no change: past the
    // divisions these kernels wait for memory.)
    uint32_t n = (uint32_t)i;
#pragma unroll 1
    for (int k = d.ndim - 1; k > 0; --k) {
      const uint32_t q = __umulhi(n, d.mg[k]) >> d.sh[k];
      const uint32_t r = n - q * (uint32_t)d.shape[k];
      oa += (int64_t)r * d.sa[k]; ob += (int64_t)r * d.sb[k]; oo += (int64_t)r * d.so[k];
      n = q;
    }
    if (d.ndim > 0) { oa += (int64_t)n * d.sa[0]; ob += (int64_t)n * d.sb[0]; oo += (int64_t)n * d.so[0]; }
    return;
  }
#pragma unroll 1
  for (int k = d.ndim - 1; k >= 0; --k) {
    const int64_t s = d.shape[k];
    const int64_t q = i / s, r = i - q * s;
    oa += r * d.sa[k]; ob += r * d.sb[k]; oo += r * d.so[k];
    i = q;
  }
}

// ---- binary ---------------------------------------------------------------------------
// mode: 0 = a[i] op b[i], 1 = a[i] op scalar, 2 = scalar op a[i]
template <typename T, typename O, int OP, bool CMP>
__global__ void ew_binary_strided(const T* __restrict__ a, const T* __restrict__ b, O* out,
                                  EwDims d, int64_t total, int mode, T scalar) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    int64_t oa, ob, oo;
    ew_offsets(d, i, oa, ob, oo);
    const T x = (mode == 2) ? scalar : a[oa];
    const T y = (mode == 0) ? b[ob] : (mode == 1 ? scalar : a[oa]);
    if constexpr (CMP) out[oo] = (O)cmp_apply<T, OP>(x, y);
    else out[oo] = (O)bin_apply<T, OP>(x, y);
  }
}

template <int OP>
__global__ void ew_binary_contig_f32(const float* __restrict__ a, const float* __restrict__ b,
                                     float* __restrict__ out, int64_t n4, int64_t total, int mode,
                                     float scalar) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 x = reinterpret_cast<const float4*>(a)[i];
    float4 y = make_float4(scalar, scalar, scalar, scalar);
    if (mode == 0) y = reinterpret_cast<const float4*>(b)[i];
    float4 r;
    if (mode == 2) {
      r.x = bin_apply<float, OP>(y.x, x.x); r.y = bin_apply<float, OP>(y.y, x.y);
      r.z = bin_apply<float, OP>(y.z, x.z); r.w = bin_apply<float, OP>(y.w, x.w);
    } else {
      r.x = bin_apply<float, OP>(x.x, y.x); r.y = bin_apply<float, OP>(x.y, y.y);
      r.z = bin_apply<float, OP>(x.z, y.z); r.w = bin_apply<float, OP>(x.w, y.w);
    }
    reinterpret_cast<float4*>(out)[i] = r;
  }
  // tail
  for (int64_t i = n4 * 4 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += stride) {
    const float x = a[i];
    const float y = mode == 0 ? b[i] : scalar;
    out[i] = mode == 2 ? bin_apply<float, OP>(y, x) : bin_apply<float, OP>(x, y);
  }
}

__device__ __forceinline__ float4 ew_load_quad(const float* __restrict__ p, int64_t step, int vec) {
  if (vec) return *reinterpret_cast<const float4*>(p);
  if (step == 0) { const float v = p[0]; return make_float4(v, v, v, v); }
  return make_float4(p[0], p[step], p[2 * step], p[3 * step]);
}

template <int OP>
__global__ void ew_unary_quad_f32(const float* __restrict__ a, float* out, EwDims d, int64_t total4, int64_t ia, int avec) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t oa, ob, oo;
    ew_offsets(d, i, oa, ob, oo);
    const float4 v = ew_load_quad(a + oa, ia, avec);
    *reinterpret_cast<float4*>(out + oo) = make_float4(un_apply<float, OP>(v.x), un_apply<float, OP>(v.y),
                                                       un_apply<float, OP>(v.z), un_apply<float, OP>(v.w));
  }
}

// ---- unary / cast ---------------------------------------------------------------------
template <typename T, typename O, int OP>
__global__ void ew_unary_strided(const T* __restrict__ a, O* out, EwDims d, int64_t total) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    int64_t oa, ob, oo;
    ew_offsets(d, i, oa, ob, oo);
    if constexpr (OP == UOP_COPY) out[oo] = (O)a[oa];
    else out[oo] = (O)un_apply<T, OP>(a[oa]);
  }
}

template <int OP>
__global__ void ew_unary_contig_f32(const float* __restrict__ a, float* __restrict__ out,
                                    int64_t n4, int64_t total) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 x = reinterpret_cast<const float4*>(a)[i];
    float4 r;
    r.x = un_apply<float, OP>(x.x); r.y = un_apply<float, OP>(x.y);
    r.z = un_apply<float, OP>(x.z); r.w = un_apply<float, OP>(x.w);
    reinterpret_cast<float4*>(out)[i] = r;
  }
  for (int64_t i = n4 * 4 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += stride)
    out[i] = un_apply<float, OP>(a[i]);
}

template <typename O>
__global__ void ew_fill_strided(O* out, EwDims d, int64_t total, O value) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    int64_t oa, ob, oo;
    ew_offsets(d, i, oa, ob, oo);
    out[oo] = value;
  }
}

// out[mask != 0] = value  (reference: Tensor.__setitem__ with a boolean key, tensor.py:279)
template <typename O>
__global__ void ew_masked_fill_strided(O* out, const uint8_t* __restrict__ mask, EwDims d,
                                       int64_t total, O value) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    int64_t oa, ob, oo;
    ew_offsets(d, i, oa, ob, oo);
    if (mask[oa]) out[oo] = value;
  }
}

// ---- host helpers ---------------------------------------------------------------------
static size_t dtype_size(int dt) {
  switch (dt) {
    case PDN_F32: return 4; case PDN_F64: return 8; case PDN_I64: return 8;
    case PDN_BOOL: return 1; case PDN_I32: return 4; case PDN_F16: return 2;
  }
  return 0;
}

// n / s for n < 2^31 by multiply-high (see ew_offsets): mg = ceil(2^(31 + l) / s), sh = l - 1, l = ceil(log2 s), s >= 2
static void ew_set_magic(EwDims& d, int k) {
  const uint64_t s = (uint64_t)d.shape[k];
  int l = 0;
  while (((uint64_t)1 << l) < s) ++l;
  d.mg[k] = (uint32_t)((((uint64_t)1 << (31 + l)) + s - 1) / s);
  d.sh[k] = (uint32_t)(l > 0 ? l - 1 : 0);
}

// Drop size-1 dims, merge adjacent dims that are jointly contiguous in every operand.
static int64_t collapse(EwDims& d, int ndim, const int64_t* shape, const int64_t* sa,
                        const int64_t* sb, const int64_t* so) {
  int64_t total = 1;
  int n = 0;
  for (int k = 0; k < ndim; ++k) {
    total *= shape[k];
    if (shape[k] == 1) continue;
    const int64_t a = sa ? sa[k] : 0, b = sb ? sb[k] : 0, o = so ? so[k] : 0;
    if (n > 0 && d.sa[n - 1] == a * shape[k] && d.sb[n - 1] == b * shape[k] &&
        d.so[n - 1] == o * shape[k]) {
      d.shape[n - 1] *= shape[k];
      d.sa[n - 1] = a; d.sb[n - 1] = b; d.so[n - 1] = o;
    } else {
      d.shape[n] = shape[k]; d.sa[n] = a; d.sb[n] = b; d.so[n] = o;
      ++n;
    }
  }
  d.ndim = n;
  static const bool no_fast = getenv("PDN_EW_NO_FASTDIV") != nullptr;       // A/B switch (tools/ew_strided_probe.py)
  d.fast = !no_fast && total > 0 && total < ((int64_t)1 << 31);
  for (int k = 0; k < n && d.fast; ++k) ew_set_magic(d, k);
  return total;
}

// Four consecutive outputs per lane for the UNARY kernels and the strided float copy: the view of `d` whose innermost
// dimension counts QUADS (one index walk and one 16-byte store per four elements; the operand is read as 16 bytes where its
// innermost stride is 1 and everything is 16-byte aligned, as one value where it is 0, as four dwords otherwise).  Needs a
// unit-stride, 16-byte aligned output whose innermost extent is a multiple of four.  Measured (rocprofv3, 75 MB operands):
// transposed copy (B, L, H, hd) -> (B, H, L, hd) 39 -> 31 us; the BINARY kernels measured no better this way (a stride-2
// view times a broadcast table 29 us either way, a row-broadcast add 33 -> 39 us) and keep one element per lane.
struct EwQuad {
  EwDims d;
  int64_t total4, ia;
  int avec;
  bool ok;
};
static EwQuad ew_quad_view(const EwDims& d0, int64_t total, const void* a, const void* out) {
  EwQuad q;
  q.ok = false;
  static const bool off = getenv("PDN_EW_NO_QUAD") != nullptr;                 // A/B switch (tools/ew_strided_probe.py)
  const int last = d0.ndim - 1;
  if (off || !d0.fast || last < 0 || d0.shape[last] % 4 || d0.so[last] != 1 || ((uintptr_t)out & 15)) return q;
  if (d0.shape[last] < 8) return q;                        // (the quad count of a divided dimension must stay >= 2)
  bool a4 = ((uintptr_t)a & 15) == 0;
  for (int k = 0; k < last; ++k) {
    if (d0.so[k] % 4) return q;
    a4 = a4 && d0.sa[k] % 4 == 0;
  }
  q.d = d0;
  q.ia = d0.sa[last];
  q.avec = a4 && q.ia == 1;
  q.d.shape[last] /= 4;
  q.d.sa[last] *= 4; q.d.so[last] *= 4;
  ew_set_magic(q.d, last);
  q.total4 = total / 4;
  q.ok = true;
  return q;
}

static inline int grid_for(int64_t n, int per_thread = 1) {
  int64_t b = (n + 256 * (int64_t)per_thread - 1) / (256 * (int64_t)per_thread);
  if (b < 1) b = 1;
  if (b > 256 * 8) b = 256 * 8;
  return (int)b;
}

static inline bool is_contig1(const EwDims& d, const int64_t* s) {
  return d.ndim == 0 || (d.ndim == 1 && s[0] == 1);
}

#define BIN_CASES(T, O, CMP, LAUNCH)                   \
  switch (op) {                                        \
    case BOP_ADD: LAUNCH(T, O, BOP_ADD, CMP); break;   \
    case BOP_SUB: LAUNCH(T, O, BOP_SUB, CMP); break;   \
    case BOP_MUL: LAUNCH(T, O, BOP_MUL, CMP); break;   \
    case BOP_DIV: LAUNCH(T, O, BOP_DIV, CMP); break;   \
    case BOP_POW: LAUNCH(T, O, BOP_POW, CMP); break;   \
    case BOP_MAX: LAUNCH(T, O, BOP_MAX, CMP); break;   \
    case BOP_MIN: LAUNCH(T, O, BOP_MIN, CMP); break;   \
    default: pdn_set_error("pdn_ew_binary: bad op %d", op); return PDN_EINVAL; \
  }
#define CMP_CASES(T, LAUNCH)                               \
  switch (op) {                                            \
    case BOP_EQ: LAUNCH(T, uint8_t, BOP_EQ, true); break;  \
    case BOP_NE: LAUNCH(T, uint8_t, BOP_NE, true); break;  \
    case BOP_LT: LAUNCH(T, uint8_t, BOP_LT, true); break;  \
    case BOP_LE: LAUNCH(T, uint8_t, BOP_LE, true); break;  \
    case BOP_GT: LAUNCH(T, uint8_t, BOP_GT, true); break;  \
    case BOP_GE: LAUNCH(T, uint8_t, BOP_GE, true); break;  \
    default: pdn_set_error("pdn_ew_binary: bad op %d", op); return PDN_EINVAL; \
  }

// out = a (op) b with NumPy broadcasting expressed through 0 strides.
// mode 0: both arrays; mode 1: a op scalar; mode 2: scalar op a.  Comparison ops (>=16)
// write uint8 0/1.  `shape` is the broadcast (output) shape; strides are in elements.
extern "C" int pdn_ew_binary(int dtype, int op, int mode, int ndim, const int64_t* shape,
                             const void* a, const int64_t* sa, const void* b, const int64_t* sb,
                             double scalar, void* out, const int64_t* so, void* stream) {
  PDN_CHECK_ARG(ndim >= 0 && ndim <= PDN_MAX_DIMS, "pdn_ew_binary: ndim %d", ndim);
  PDN_CHECK_ARG(mode >= 0 && mode <= 2, "pdn_ew_binary: mode %d", mode);
  hipStream_t st = (hipStream_t)stream;
  EwDims d;
  const int64_t total = collapse(d, ndim, shape, sa, mode == 0 ? sb : nullptr, so);
  if (total == 0) return PDN_OK;
  PDN_CHECK_ARG(a && out && (mode != 0 || b), "pdn_ew_binary: null operand");
  const bool cmp = op >= BOP_EQ;
  const int grid = grid_for(total, 4);

#define L_STRIDED(T, O, OPC, CMP)                                                          \
  hipLaunchKernelGGL((ew_binary_strided<T, O, OPC, CMP>), dim3(grid), dim3(256), 0, st,     \
                     (const T*)a, (const T*)b, (O*)out, d, total, mode, (T)scalar)
  if (dtype == PDN_F32) {
    const bool contig = !cmp && is_contig1(d, d.sa) && is_contig1(d, d.so) &&
                        (mode != 0 || is_contig1(d, d.sb)) && (((uintptr_t)a | (uintptr_t)out |
                        (mode == 0 ? (uintptr_t)b : 0)) & 15) == 0;
    if (contig) {
      const int64_t n4 = total / 4;
      const int g = grid_for(n4 > 0 ? n4 : 1, 2);
#define L_CONTIG(T, O, OPC, CMP)                                                            \
  hipLaunchKernelGGL((ew_binary_contig_f32<OPC>), dim3(g), dim3(256), 0, st, (const float*)a, \
                     (const float*)b, (float*)out, n4, total, mode, (float)scalar)
      BIN_CASES(float, float, false, L_CONTIG)
    } else if (cmp) {
      CMP_CASES(float, L_STRIDED)
    } else {
      BIN_CASES(float, float, false, L_STRIDED)
    }
  } else if (dtype == PDN_F64) {
    if (cmp) { CMP_CASES(double, L_STRIDED) } else { BIN_CASES(double, double, false, L_STRIDED) }
  } else if (dtype == PDN_F16) {
    if (cmp) { CMP_CASES(half_t, L_STRIDED) } else { BIN_CASES(half_t, half_t, false, L_STRIDED) }
  } else if (dtype == PDN_I64) {
    if (cmp) { CMP_CASES(int64_t, L_STRIDED) }
    else {
      switch (op) {
        case BOP_ADD: L_STRIDED(int64_t, int64_t, BOP_ADD, false); break;
        case BOP_SUB: L_STRIDED(int64_t, int64_t, BOP_SUB, false); break;
        case BOP_MUL: L_STRIDED(int64_t, int64_t, BOP_MUL, false); break;
        default: pdn_set_error("pdn_ew_binary: op %d unsupported for int64", op); return PDN_EUNSUPPORTED;
      }
    }
  } else if (dtype == PDN_BOOL) {
    if (cmp) { CMP_CASES(uint8_t, L_STRIDED) }
    else { pdn_set_error("pdn_ew_binary: arithmetic on bool"); return PDN_EUNSUPPORTED; }
  } else {
    pdn_set_error("pdn_ew_binary: dtype %d unsupported", dtype);
    return PDN_EUNSUPPORTED;
  }
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

#define UN_CASES(T, LAUNCH)                            \
  switch (op) {                                        \
    case UOP_COPY: LAUNCH(T, UOP_COPY); break;         \
    case UOP_NEG: LAUNCH(T, UOP_NEG); break;           \
    case UOP_EXP: LAUNCH(T, UOP_EXP); break;           \
    case UOP_LOG: LAUNCH(T, UOP_LOG); break;           \
    case UOP_ABS: LAUNCH(T, UOP_ABS); break;           \
    case UOP_SIGN: LAUNCH(T, UOP_SIGN); break;         \
    case UOP_SQRT: LAUNCH(T, UOP_SQRT); break;         \
    case UOP_SQUARE: LAUNCH(T, UOP_SQUARE); break;     \
    case UOP_RECIP: LAUNCH(T, UOP_RECIP); break;       \
    case UOP_SIGMOID: LAUNCH(T, UOP_SIGMOID); break;   \
    case UOP_TANH: LAUNCH(T, UOP_TANH); break;         \
    default: pdn_set_error("pdn_ew_unary: bad op %d", op); return PDN_EINVAL; \
  }

extern "C" int pdn_ew_unary(int dtype, int op, int ndim, const int64_t* shape, const void* a,
                            const int64_t* sa, void* out, const int64_t* so, void* stream) {
  PDN_CHECK_ARG(ndim >= 0 && ndim <= PDN_MAX_DIMS, "pdn_ew_unary: ndim %d", ndim);
  hipStream_t st = (hipStream_t)stream;
  EwDims d;
  const int64_t total = collapse(d, ndim, shape, sa, nullptr, so);
  if (total == 0) return PDN_OK;
  PDN_CHECK_ARG(a && out, "pdn_ew_unary: null operand");
  const int grid = grid_for(total, 4);
#define LU_STRIDED(T, OPC)                                                                  \
  hipLaunchKernelGGL((ew_unary_strided<T, T, OPC>), dim3(grid), dim3(256), 0, st, (const T*)a, \
                     (T*)out, d, total)
  if (dtype == PDN_F32) {
    const bool contig = is_contig1(d, d.sa) && is_contig1(d, d.so) &&
                        (((uintptr_t)a | (uintptr_t)out) & 15) == 0;
    if (contig) {
      const int64_t n4 = total / 4;
      const int g = grid_for(n4 > 0 ? n4 : 1, 2);
#define LU_CONTIG(T, OPC)                                                                    \
  hipLaunchKernelGGL((ew_unary_contig_f32<OPC>), dim3(g), dim3(256), 0, st, (const float*)a, \
                     (float*)out, n4, total)
      UN_CASES(float, LU_CONTIG)
    } else {
      const EwQuad q = ew_quad_view(d, total, a, out);
      if (q.ok) {
        const int gq = grid_for(q.total4, 2);
#define LU_QUAD(T, OPC)                                                                                        \
  hipLaunchKernelGGL((ew_unary_quad_f32<OPC>), dim3(gq), dim3(256), 0, st, (const float*)a, (float*)out, q.d, \
                     q.total4, q.ia, q.avec)
        UN_CASES(float, LU_QUAD)
      } else {
        UN_CASES(float, LU_STRIDED)
      }
    }
  } else if (dtype == PDN_F64) {
    UN_CASES(double, LU_STRIDED)
  } else if (dtype == PDN_F16) {
    UN_CASES(half_t, LU_STRIDED)
  } else {
    pdn_set_error("pdn_ew_unary: dtype %d unsupported", dtype);
    return PDN_EUNSUPPORTED;
  }
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

// Strided copy with dtype conversion (astype / contiguous() / slice assignment).
template <typename S>
static int cast_from(int dst_dtype, const void* a, void* out, const EwDims& d, int64_t total,
                     hipStream_t st) {
  const int grid = grid_for(total, 4);
#define LC(O)                                                                               \
  hipLaunchKernelGGL((ew_unary_strided<S, O, UOP_COPY>), dim3(grid), dim3(256), 0, st,      \
                     (const S*)a, (O*)out, d, total)
  switch (dst_dtype) {
    case PDN_F32: LC(float); break;
    case PDN_F64: LC(double); break;
    case PDN_I64: LC(int64_t); break;
    case PDN_I32: LC(int32_t); break;
    case PDN_F16: LC(half_t); break;
    case PDN_BOOL: {
      // numpy astype(bool): nonzero -> True
      hipLaunchKernelGGL((ew_binary_strided<S, uint8_t, BOP_NE, true>), dim3(grid), dim3(256), 0,
                         st, (const S*)a, (const S*)nullptr, (uint8_t*)out, d, total, 1, (S)0);
      break;
    }
    default: pdn_set_error("pdn_cast: bad dst dtype %d", dst_dtype); return PDN_EINVAL;
  }
  return PDN_OK;
}

extern "C" int pdn_cast(int src_dtype, int dst_dtype, int ndim, const int64_t* shape,
                        const void* a, const int64_t* sa, void* out, const int64_t* so,
                        void* stream) {
  PDN_CHECK_ARG(ndim >= 0 && ndim <= PDN_MAX_DIMS, "pdn_cast: ndim %d", ndim);
  hipStream_t st = (hipStream_t)stream;
  EwDims d;
  const int64_t total = collapse(d, ndim, shape, sa, nullptr, so);
  if (total == 0) return PDN_OK;
  PDN_CHECK_ARG(a && out, "pdn_cast: null operand");
  if (src_dtype == dst_dtype && is_contig1(d, d.sa) && is_contig1(d, d.so)) {
    PDN_HIP(hipMemcpyAsync(out, a, total * dtype_size(src_dtype), hipMemcpyDeviceToDevice, st));
    return PDN_OK;
  }
  if (src_dtype == PDN_F32 && dst_dtype == PDN_F32) {       // transposed / sliced views made contiguous: 16 bytes per lane
    const EwQuad q = ew_quad_view(d, total, a, out);
    if (q.ok) {
      hipLaunchKernelGGL((ew_unary_quad_f32<UOP_COPY>), dim3(grid_for(q.total4, 2)), dim3(256), 0, st, (const float*)a,
                         (float*)out, q.d, q.total4, q.ia, q.avec);
      PDN_LAUNCH_CHECK();
      return PDN_OK;
    }
  }
  int rc;
  switch (src_dtype) {
    case PDN_F32: rc = cast_from<float>(dst_dtype, a, out, d, total, st); break;
    case PDN_F64: rc = cast_from<double>(dst_dtype, a, out, d, total, st); break;
    case PDN_I64: rc = cast_from<int64_t>(dst_dtype, a, out, d, total, st); break;
    case PDN_I32: rc = cast_from<int32_t>(dst_dtype, a, out, d, total, st); break;
    case PDN_BOOL: rc = cast_from<uint8_t>(dst_dtype, a, out, d, total, st); break;
    case PDN_F16: rc = cast_from<half_t>(dst_dtype, a, out, d, total, st); break;
    default: pdn_set_error("pdn_cast: bad src dtype %d", src_dtype); return PDN_EINVAL;
  }
  if (rc) return rc;
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

extern "C" int pdn_fill(int dtype, double value, int ndim, const int64_t* shape, void* out,
                        const int64_t* so, void* stream) {
  PDN_CHECK_ARG(ndim >= 0 && ndim <= PDN_MAX_DIMS, "pdn_fill: ndim %d", ndim);
  hipStream_t st = (hipStream_t)stream;
  EwDims d;
  const int64_t total = collapse(d, ndim, shape, nullptr, nullptr, so);
  if (total == 0) return PDN_OK;
  PDN_CHECK_ARG(out, "pdn_fill: null output");
  if (value == 0.0 && is_contig1(d, d.so)) {
    PDN_HIP(hipMemsetAsync(out, 0, total * dtype_size(dtype), st));
    return PDN_OK;
  }
  const int grid = grid_for(total, 4);
#define LF(O) hipLaunchKernelGGL((ew_fill_strided<O>), dim3(grid), dim3(256), 0, st, (O*)out, d, total, (O)value)
  switch (dtype) {
    case PDN_F32: LF(float); break;
    case PDN_F64: LF(double); break;
    case PDN_I64: LF(int64_t); break;
    case PDN_I32: LF(int32_t); break;
    case PDN_BOOL: LF(uint8_t); break;
    case PDN_F16: LF(half_t); break;
    default: pdn_set_error("pdn_fill: bad dtype %d", dtype); return PDN_EINVAL;
  }
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

extern "C" int pdn_masked_fill(int dtype, double value, int ndim, const int64_t* shape,
                               const void* mask, const int64_t* smask, void* out,
                               const int64_t* so, void* stream) {
  PDN_CHECK_ARG(ndim >= 0 && ndim <= PDN_MAX_DIMS, "pdn_masked_fill: ndim %d", ndim);
  hipStream_t st = (hipStream_t)stream;
  EwDims d;
  const int64_t total = collapse(d, ndim, shape, smask, nullptr, so);
  if (total == 0) return PDN_OK;
  PDN_CHECK_ARG(out && mask, "pdn_masked_fill: null operand");
  const int grid = grid_for(total, 4);
#define LMF(O)                                                                              \
  hipLaunchKernelGGL((ew_masked_fill_strided<O>), dim3(grid), dim3(256), 0, st, (O*)out,    \
                     (const uint8_t*)mask, d, total, (O)value)
  switch (dtype) {
    case PDN_F32: LMF(float); break;
    case PDN_F64: LMF(double); break;
    case PDN_I64: LMF(int64_t); break;
    case PDN_BOOL: LMF(uint8_t); break;
    case PDN_F16: LMF(half_t); break;
    default: pdn_set_error("pdn_masked_fill: bad dtype %d", dtype); return PDN_EINVAL;
  }
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

