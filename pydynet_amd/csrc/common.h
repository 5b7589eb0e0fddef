// Shared helpers for the pdnhip C-ABI library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define PDN_MAX_DIMS 8

// ---- error convention (include/pdn_hip.h): every entry point returns 0 on success,
// a negative PDN_E* code for argument errors, or a positive hipError_t.
enum {
  PDN_OK = 0,
  PDN_EINVAL = -1,      // bad argument (shape / stride / alignment / enum)
  PDN_EUNSUPPORTED = -2,  // valid request the library does not implement
  PDN_EWORKSPACE = -3,  // caller-provided workspace too small
};

void pdn_set_error(const char* fmt, ...);

#define PDN_CHECK_ARG(cond, ...)          \
  do {                                    \
    if (!(cond)) {                        \
      pdn_set_error(__VA_ARGS__);         \
      return PDN_EINVAL;                  \
    }                                     \
  } while (0)

#define PDN_LAUNCH_CHECK()                                            \
  do {                                                                \
    hipError_t _e = hipGetLastError();                                \
    if (_e != hipSuccess) {                                           \
      pdn_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,    \
                    hipGetErrorString(_e));                           \
      return (int)_e;                                                 \
    }                                                                 \
  } while (0)

#define PDN_HIP(call)                                                 \
  do {                                                                \
    hipError_t _e = (call);                                           \
    if (_e != hipSuccess) {                                           \
      pdn_set_error("%s:%d %s: %s", __FILE__, __LINE__, #call,        \
                    hipGetErrorString(_e));                           \
      return (int)_e;                                                 \
    }                                                                 \
  } while (0)

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Strided N-d view descriptor passed by value to generic kernels.
struct PdnView {
  int ndim;
  int64_t shape[PDN_MAX_DIMS];
  int64_t stride[PDN_MAX_DIMS];  // in elements
};

// wave64 reductions -----------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide reductions for blockDim.x a multiple of 64 (<= 1024). `smem` holds >= 16 T.
template <typename T>
__device__ __forceinline__ T block_sum(T v, T* smem) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if (nw == 1) return v;
  __syncthreads();
  if (lane == 0) smem[wid] = v;
  __syncthreads();
  T r = (T)0;
  for (int i = 0; i < nw; ++i) r += smem[i];
  return r;
}
// Barrier for LDS traffic only.  __syncthreads() carries a fence that drains vmcnt to 0: it waits for EVERY global
// load in flight, i.e. it ends any overlap of a prefetch with the LDS work in front of it.  Only where no thread
// reads global memory another thread of the workgroup wrote in this kernel.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// block_sum / block_max below with that barrier
__device__ __forceinline__ float block_sum_lds(float v, float* smem) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if (nw == 1) return v;
  lds_barrier();
  if (lane == 0) smem[wid] = v;
  lds_barrier();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += smem[i];
  return r;
}
__device__ __forceinline__ float block_max_lds(float v, float* smem) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if (nw == 1) return v;
  lds_barrier();
  if (lane == 0) smem[wid] = v;
  lds_barrier();
  float r = smem[0];
  for (int i = 1; i < nw; ++i) r = fmaxf(r, smem[i]);
  return r;
}
__device__ __forceinline__ float block_max(float v, float* smem) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if (nw == 1) return v;
  __syncthreads();
  if (lane == 0) smem[wid] = v;
  __syncthreads();
  float r = smem[0];
  for (int i = 1; i < nw; ++i) r = fmaxf(r, smem[i]);
  return r;
}
