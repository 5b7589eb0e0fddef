// Axis reductions for the `xp` facade (gfx950):  sum / mean / max / min / argmax / argmin
// with NumPy axis semantics, replacing `getattr(xp, name)(x, axis, keepdims)`
// (pydynet/core/tensor.py:701,705) and the engine's un-broadcast sums (tensor.py:360-370).
//
// The host permutes the view into (kept dims..., reduced dims...).  Two device strategies:
//   * ROW: the reduced run is walked by a whole workgroup (or its k-chunk of it) with
//     wave64 shuffle reductions -- used when the innermost (unit-stride) dim is reduced;
//   * COL: the innermost dim is kept -- 64 consecutive outputs per workgroup, 4 row lanes,
//     loads coalesced along the kept dim.
// Long reductions are split over workgroups into a workspace and finished by a second
// pass in fixed order (deterministic; no float atomics).
#include <type_traits>
#include <algorithm>
#include "common.h"

enum { PDN_F32 = 0, PDN_F64 = 1, PDN_I64 = 2, PDN_BOOL = 3 };
enum { ROP_SUM = 0, ROP_MEAN, ROP_MAX, ROP_MIN, ROP_ARGMAX, ROP_ARGMIN };

struct RedDims {
  int nk, nr;                       // number of kept / reduced dims
  int64_t kshape[PDN_MAX_DIMS], kstride[PDN_MAX_DIMS];
  int64_t rshape[PDN_MAX_DIMS], rstride[PDN_MAX_DIMS];
};

__device__ __forceinline__ int64_t off_of(int64_t i, int n, const int64_t* shape,
                                          const int64_t* stride) {
  int64_t o = 0;
#pragma unroll 1
  for (int k = n - 1; k >= 0; --k) {
    const int64_t q = i / shape[k];
    o += (i - q * shape[k]) * stride[k];
    i = q;
  }
  return o;
}

template <typename T, int OP> struct Acc {
  T v; int64_t idx;
  __device__ __forceinline__ void init() {
    idx = -1;
    if constexpr (OP == ROP_SUM || OP == ROP_MEAN) v = (T)0;
    else if constexpr (OP == ROP_MAX || OP == ROP_ARGMAX) v = (T)(-INFINITY);
    else v = (T)INFINITY;
  }
  __device__ __forceinline__ void push(T x, int64_t i) {
    if (OP == ROP_SUM || OP == ROP_MEAN) v += x;
    else if (OP == ROP_MAX) v = (x > v || x != x) ? x : v;
    else if (OP == ROP_MIN) v = (x < v || x != x) ? x : v;
    else if (OP == ROP_ARGMAX) { if (idx < 0 || x > v || (x == v && i < idx)) { v = x; idx = i; } }
    else { if (idx < 0 || x < v || (x == v && i < idx)) { v = x; idx = i; } }
  }
  __device__ __forceinline__ void merge(T ov, int64_t oi) {
    if (OP >= ROP_ARGMAX) { if (oi >= 0) push(ov, oi); }
    else push(ov, 0);
  }
};

template <typename T> __device__ __forceinline__ T shfl_xor_t(T v, int o) { return __shfl_xor(v, o, 64); }
template <> __device__ __forceinline__ int64_t shfl_xor_t<int64_t>(int64_t v, int o) {
  int lo = (int)(v & 0xffffffffll), hi = (int)(v >> 32);
  lo = __shfl_xor(lo, o, 64); hi = __shfl_xor(hi, o, 64);
  return ((int64_t)hi << 32) | (uint32_t)lo;
}

template <typename T, int OP>
__device__ __forceinline__ void block_merge(Acc<T, OP>& a, T* sv, int64_t* si) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const T ov = shfl_xor_t<T>(a.v, o);
    const int64_t oi = OP >= ROP_ARGMAX ? shfl_xor_t<int64_t>(a.idx, o) : 0;
    a.merge(ov, oi);
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (nw > 1) {
    __syncthreads();
    if (lane == 0) { sv[wid] = a.v; si[wid] = a.idx; }
    __syncthreads();
    a.v = sv[0]; a.idx = si[0];
    for (int w = 1; w < nw; ++w) a.merge(sv[w], si[w]);
  }
}

// ROW strategy: grid = (outN, chunks).  Partial (or final) result for output o, chunk c.
template <typename T, int OP>
__global__ void reduce_row_kernel(const T* __restrict__ x, RedDims d, int64_t R, int64_t rc,
                                  T* pv, int64_t* pi, int64_t outN) {
  __shared__ T sv[16];
  __shared__ int64_t si[16];
  const int64_t o = blockIdx.x;
  const int c = blockIdx.y;
  const int64_t base = off_of(o, d.nk, d.kshape, d.kstride);
  const int64_t r0 = c * rc, r1 = min(R, r0 + rc);
  Acc<T, OP> a; a.init();
  if (d.nr == 1) {
    const int64_t s = d.rstride[0];
    for (int64_t r = r0 + threadIdx.x; r < r1; r += blockDim.x) a.push(x[base + r * s], r);
  } else {
    for (int64_t r = r0 + threadIdx.x; r < r1; r += blockDim.x)
      a.push(x[base + off_of(r, d.nr, d.rshape, d.rstride)], r);
  }
  block_merge<T, OP>(a, sv, si);
  if (threadIdx.x == 0) {
    pv[(int64_t)c * outN + o] = a.v;
    if (OP >= ROP_ARGMAX) pi[(int64_t)c * outN + o] = a.idx;
  }
}

// COL strategy: block = 64 outputs x 4 row lanes; grid = (ceil(outN/64), chunks).
template <typename T, int OP>
__global__ void reduce_col_kernel(const T* __restrict__ x, RedDims d, int64_t R, int64_t rc,
                                  T* pv, int64_t* pi, int64_t outN) {
  __shared__ T sv[4][64];
  __shared__ int64_t si[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t o = blockIdx.x * 64ll + tx;
  const int c = blockIdx.y;
  const int64_t r0 = c * rc, r1 = min(R, r0 + rc);
  Acc<T, OP> a; a.init();
  if (o < outN) {
    const int64_t base = off_of(o, d.nk, d.kshape, d.kstride);
    if (d.nr == 1) {
      const int64_t s = d.rstride[0];
      for (int64_t r = r0 + ty; r < r1; r += 4) a.push(x[base + r * s], r);
    } else {
      for (int64_t r = r0 + ty; r < r1; r += 4)
        a.push(x[base + off_of(r, d.nr, d.rshape, d.rstride)], r);
    }
  }
  sv[ty][tx] = a.v; si[ty][tx] = a.idx;
  __syncthreads();
  if (ty == 0 && o < outN) {
    for (int w = 1; w < 4; ++w) a.merge(sv[w][tx], si[w][tx]);
    pv[(int64_t)c * outN + o] = a.v;
    if (OP >= ROP_ARGMAX) pi[(int64_t)c * outN + o] = a.idx;
  }
}

// Column sums of a row-major float matrix whose rows are 16-byte aligned (the bias gradient of a wide layer: 65536 x 1024
// at 2.3 TB/s through the dword kernel above): a thread owns FOUR columns, a workgroup 256 columns x 4 row lanes, eight
// independent 16-byte loads in flight per thread.
__global__ __launch_bounds__(256) void reduce_colsum4_kernel(const float* __restrict__ x, int64_t ld, int64_t R, int64_t rc,
                                                             float* __restrict__ pv, int64_t outN) {
  __shared__ float4 sv[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t o = (blockIdx.x * 64ll + tx) * 4;
  const int64_t r0 = blockIdx.y * rc, r1 = min(R, r0 + rc);
  float4 a[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (o < outN) {
    const float* col = x + o;
    int64_t r = r0 + ty;
    for (; r + 28 < r1; r += 32) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(col + (r + 4 * u) * ld);
#pragma unroll
      for (int u = 0; u < 8; ++u) { a[u & 3].x += v[u].x; a[u & 3].y += v[u].y; a[u & 3].z += v[u].z; a[u & 3].w += v[u].w; }
    }
    for (; r < r1; r += 4) {
      const float4 v = *reinterpret_cast<const float4*>(col + r * ld);
      a[0].x += v.x; a[0].y += v.y; a[0].z += v.z; a[0].w += v.w;
    }
  }
  float4 t;
  t.x = (a[0].x + a[1].x) + (a[2].x + a[3].x); t.y = (a[0].y + a[1].y) + (a[2].y + a[3].y);
  t.z = (a[0].z + a[1].z) + (a[2].z + a[3].z); t.w = (a[0].w + a[1].w) + (a[2].w + a[3].w);
  sv[ty][tx] = t;
  __syncthreads();
  if (ty == 0 && o < outN) {
#pragma unroll
    for (int w = 1; w < 4; ++w) { t.x += sv[w][tx].x; t.y += sv[w][tx].y; t.z += sv[w][tx].z; t.w += sv[w][tx].w; }
    *reinterpret_cast<float4*>(pv + (int64_t)blockIdx.y * outN + o) = t;
  }
}

// Second pass: combine `chunks` partials per output in a fixed order, apply mean scale.  A workgroup owns
// 64 outputs; its 4 row lanes each merge every fourth partial (independent loads in flight), then the four
// lane results are merged in lane order -- deterministic, and 4x shorter than one serial walk per output.
template <typename T, int OP>
__global__ void reduce_finish_kernel(const T* __restrict__ pv, const int64_t* __restrict__ pi,
                                     int chunks, int64_t outN, T scale, T* out, int64_t* outi) {
  __shared__ T sv[4][64];
  __shared__ int64_t si[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t o = blockIdx.x * 64ll + tx;
  Acc<T, OP> a; a.init();
  if (o < outN)
    for (int c = ty; c < chunks; c += 4)
      a.merge(pv[(int64_t)c * outN + o], OP >= ROP_ARGMAX ? pi[(int64_t)c * outN + o] : 0);
  sv[ty][tx] = a.v; si[ty][tx] = a.idx;
  __syncthreads();
  if (ty == 0 && o < outN) {
    for (int w = 1; w < 4; ++w) a.merge(sv[w][tx], si[w][tx]);
    if (OP >= ROP_ARGMAX) outi[o] = a.idx;
    else out[o] = (OP == ROP_MEAN) ? a.v / scale : a.v;
  }
}

template <typename T, int OP>
static int run_reduce(const T* x, const RedDims& d, int64_t outN, int64_t R, bool col,
                      void* out, void* ws, int64_t ws_bytes, hipStream_t st) {
  // choose chunking of the reduced run
  int chunks = 1;
  const int64_t par = col ? cdiv64(outN, 64) : outN;      // workgroups without chunking
  const int64_t per_chunk_min = col ? 64 : 2048;
  if (par < 1024 && R >= 2 * per_chunk_min) {
    int64_t want = cdiv64(1024, par);
    int64_t maxc = R / per_chunk_min;
    chunks = (int)(want < maxc ? want : maxc);
    if (chunks > 256) chunks = 256;
  }
  constexpr bool ARG = OP >= ROP_ARGMAX;
  const size_t rec = sizeof(T) + (ARG ? sizeof(int64_t) : 0);
  // wide float column sums: four columns per thread (reduce_colsum4_kernel), ~2048 workgroups
  const bool colsum4 = std::is_same<T, float>::value && (OP == ROP_SUM || OP == ROP_MEAN) && col && d.nk == 1 &&
                       d.kstride[0] == 1 && d.nr == 1 && outN % 4 == 0 && outN >= 256 && R >= 4096 &&
                       d.rstride[0] % 4 == 0 && (((uintptr_t)x | (uintptr_t)out | (uintptr_t)ws) & 15) == 0;
  if (colsum4) {
    const int64_t want = cdiv64(1024, cdiv64(outN, 256));      // (the second pass walks `chunks` partials per output: keep them few)
    chunks = (int)std::min<int64_t>(std::min<int64_t>(want, R / 128), 256);
    if (chunks < 1) chunks = 1;
  }
  if (chunks > 1 && (int64_t)(rec * chunks * outN) > ws_bytes) chunks = 1;
  const int64_t rc = cdiv64(R, chunks);
  chunks = (int)cdiv64(R, rc);
  const T scale = (T)(R > 0 ? R : 1);  // mean divides by the count

  // Single-chunk sum/max/min write straight to `out`; everything else stages partial
  // (value[, index]) records in the workspace and is finished by a second pass.
  const bool direct = (chunks == 1) && !ARG && OP != ROP_MEAN;
  T* pv; int64_t* pi;
  if (direct) {
    pv = (T*)out; pi = nullptr;
  } else {
    if ((int64_t)(rec * chunks * outN) > ws_bytes) {
      pdn_set_error("pdn_reduce: workspace too small (%lld bytes needed)",
                    (long long)(rec * chunks * outN));
      return PDN_EWORKSPACE;
    }
    pi = (int64_t*)ws;                                   // 8-byte aligned region first
    pv = ARG ? (T*)(pi + (size_t)chunks * outN) : (T*)ws;
  }

  if (colsum4) {
    dim3 grid((unsigned)cdiv64(outN, 256), chunks);
    hipLaunchKernelGGL(reduce_colsum4_kernel, grid, dim3(256), 0, st, (const float*)x, d.rstride[0], R, rc, (float*)pv, outN);
  } else if (col) {
    dim3 grid((unsigned)cdiv64(outN, 64), chunks);
    hipLaunchKernelGGL((reduce_col_kernel<T, OP>), grid, dim3(256), 0, st, x, d, R, rc, pv, pi, outN);
  } else {
    PDN_CHECK_ARG(outN <= 2147483647ll, "pdn_reduce: too many outputs");
    dim3 grid((unsigned)outN, chunks);
    const int threads = (rc >= 1024) ? 256 : (rc >= 128 ? 128 : 64);
    hipLaunchKernelGGL((reduce_row_kernel<T, OP>), grid, dim3(threads), 0, st, x, d, R, rc, pv, pi, outN);
  }
  PDN_LAUNCH_CHECK();
  if (!direct) {
    const unsigned g = (unsigned)cdiv64(outN, 64);
    hipLaunchKernelGGL((reduce_finish_kernel<T, OP>), dim3(g), dim3(256), 0, st, (const T*)pv,
                       (const int64_t*)pi, chunks, outN, scale, (T*)out, (int64_t*)out);
    PDN_LAUNCH_CHECK();
  }
  return PDN_OK;
}

// Reduce `x` (ndim, shape, element strides) over the axes flagged in `reduce_axis[ndim]`.
// `out` is contiguous over the kept dims in their original order (int64 for arg ops).
// For arg ops the index is the C-order position within the reduced dims (NumPy semantics
// for axis=int and axis=None).
extern "C" int pdn_reduce(int dtype, int op, int ndim, const int64_t* shape, const int64_t* strides,
                          const uint8_t* reduce_axis, const void* x, void* out, void* workspace,
                          int64_t workspace_bytes, void* stream) {
  PDN_CHECK_ARG(ndim >= 0 && ndim <= PDN_MAX_DIMS, "pdn_reduce: ndim %d", ndim);
  PDN_CHECK_ARG(op >= ROP_SUM && op <= ROP_ARGMIN, "pdn_reduce: op %d", op);
  hipStream_t st = (hipStream_t)stream;
  RedDims d; d.nk = d.nr = 0;
  int64_t outN = 1, R = 1;
  int64_t min_kept_stride = INT64_MAX, min_red_stride = INT64_MAX;
  for (int k = 0; k < ndim; ++k) {
    if (reduce_axis[k]) {
      R *= shape[k];
      if (shape[k] == 1) continue;
      // merge with previous reduced dim when jointly contiguous
      if (d.nr > 0 && d.rstride[d.nr - 1] == strides[k] * shape[k]) {
        d.rshape[d.nr - 1] *= shape[k]; d.rstride[d.nr - 1] = strides[k];
      } else { d.rshape[d.nr] = shape[k]; d.rstride[d.nr] = strides[k]; ++d.nr; }
      if (llabs(strides[k]) < min_red_stride) min_red_stride = llabs(strides[k]);
    } else {
      outN *= shape[k];
      if (shape[k] == 1) continue;
      if (d.nk > 0 && d.kstride[d.nk - 1] == strides[k] * shape[k]) {
        d.kshape[d.nk - 1] *= shape[k]; d.kstride[d.nk - 1] = strides[k];
      } else { d.kshape[d.nk] = shape[k]; d.kstride[d.nk] = strides[k]; ++d.nk; }
      if (llabs(strides[k]) < min_kept_stride) min_kept_stride = llabs(strides[k]);
    }
  }
  if (outN == 0) return PDN_OK;
  PDN_CHECK_ARG(R > 0 || op == ROP_SUM, "pdn_reduce: zero-size reduction without identity");
  PDN_CHECK_ARG(x && out, "pdn_reduce: null operand");
  if (d.nr == 0) { d.rshape[0] = 1; d.rstride[0] = 0; d.nr = 1; }
  const bool col = d.nk > 0 && min_kept_stride < min_red_stride && outN >= 16;

#define RUN(T, OPC) return run_reduce<T, OPC>((const T*)x, d, outN, R, col, out, workspace, workspace_bytes, st)
#define OPS(T)                                   \
  switch (op) {                                  \
    case ROP_SUM: RUN(T, ROP_SUM);               \
    case ROP_MEAN: RUN(T, ROP_MEAN);             \
    case ROP_MAX: RUN(T, ROP_MAX);               \
    case ROP_MIN: RUN(T, ROP_MIN);               \
    case ROP_ARGMAX: RUN(T, ROP_ARGMAX);         \
    default: RUN(T, ROP_ARGMIN);                 \
  }
  if (dtype == PDN_F32) { OPS(float) }
  else if (dtype == PDN_F64) { OPS(double) }
  else if (dtype == PDN_I64) {
    switch (op) {
      case ROP_SUM: RUN(int64_t, ROP_SUM);
      default: pdn_set_error("pdn_reduce: int64 supports sum only"); return PDN_EUNSUPPORTED;
    }
  }
  pdn_set_error("pdn_reduce: dtype %d unsupported", dtype);
  return PDN_EUNSUPPORTED;
}
