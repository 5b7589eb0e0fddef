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

// ---- kernel launch counters (include/pdn_hip.h: pdn_kernel_counters): which kernel an entry point really launched.
// bench.py's parity gates and the tests read them -- a dispatch regression to a slower kernel must not stay green.
enum {
  PDN_CNT_ROWRES_CHUNK = 0,       // gemm_rowres_kernel (chunk kernel), any epilogue
  PDN_CNT_ROWTILE_PLAIN = 1,      // gemm_rowtile_kernel<EPI 0>
  PDN_CNT_ROWTILE_SWIGLU_FWD = 2, // EPI 1: gate | up + SwiGLU
  PDN_CNT_ROWTILE_SWIGLU_BWD = 3, // EPI 2: dh + SwiGLU backward
  PDN_CNT_ROWTILE_ROPE = 4,       // EPI 3: q | k | v + RoPE
  PDN_CNT_ROWTILE_ROWMAX = 5,     // EPI 5: vocabulary projection + row maxima
  PDN_CNT_ROWRES_CHUNK_EPI = 6,   // gemm_rowres_kernel with a fused epilogue (EPI 1 / 2 / 3 / 4 / 5)
  PDN_CNT_ATT_P_FWD = 7,          // attention_p_fwd_kernel (persistent, DMA-staged)
  PDN_CNT_ATT_P_BWD = 8,          // attention_p_bwd_dq / dkv kernels
  PDN_CNT_ATT_RES_FWD = 9,        // attention_fwd_kernel (resident, chunked)
  PDN_CNT_ATT_RES_BWD = 10,
  PDN_CNT_ATT_STREAM = 11,        // attn_*_stream_kernel, forward or backward
  PDN_CNT_CE_DX_DEFERRED = 12,    // gemm_outres_kernel<.., CE 2>: lm_head input gradient + sum of exponentials
  PDN_CNT_CE_DW = 13,             // gemm_outres_tn_kernel with the cross-entropy gradient formed inside
  PDN_CNT_OUTRES = 14,            // gemm_outres_kernel, plain
  PDN_CNT_OUTRES_TN = 15,         // gemm_outres_tn_kernel, plain
  PDN_CNT_LINEAR_RELU_FWD = 16,   // tiled kernel with the relu + bit-mask store (pdn_linear_relu_fwd_f32)
  PDN_CNT_LINEAR_DX_MASKED = 17,  // tiled kernel with the bit mask applied in the store (pdn_linear_dx_masked_f32)
  PDN_CNT_CE_SMALL = 18,          // ce_small_kernel: cross entropy over <= 32 classes, one thread per row
  PDN_CNT_TILED_SWIGLU_FWD = 19,  // tiled kernel with SwiGLU in the store (pdn_gateup_swiglu_tiled_fwd_f32)
  PDN_CNT_TILED_SWIGLU_BWD = 20,  // ... with the SwiGLU backward in the store (pdn_swiglu_bwd_tiled_f32)
  PDN_CNT_CONV_QUAD_FWD = 21,     // conv_quad_fwd_kernel (conv + relu + max_pool of the LeNet shapes, csrc/conv_quad.hip)
  PDN_CNT_CONV_QUAD_DGRAD = 22,   // conv_quad_dgrad_kernel (col2im-style data gradient)
  PDN_CNT_CONV_QUAD_WGRAD = 23,   // conv_quad_wgrad_kernel (shifted image copies)
  PDN_CNT_SLOTS = 24
};
void pdn_count(int slot);

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
