// Products with a handful of output columns (gfx950): the classifier head of examples/pydynet/mnist.py:70-78,
// Linear(1024, 10) at batch 65536 -- `x @ W` forward (tensor.py:659) and `x^T @ grad` for dW (tensor.py:675).
//
// Ten columns fill a third of one 32-wide MFMA tile and the tiled kernel still stages both operands through LDS with a
// barrier per 32 contraction steps: 148 us forward / 128 us weight gradient for 268 MB of activations that HBM
// delivers in ~45 us.  These are bandwidth kernels instead (10 multiply-adds per loaded float):
//
//  * gemm_narrow_nn_kernel: C (M x N) = A (M x K) B (K x N) + bias, N <= 16: 16 x 16 x 4 MFMA tiles (the output is one tile
//    wide), A staged per wave through LDS strips with coalesced loads, the four waves of a workgroup split the contraction.
//  (The third product of such a head, the input gradient with a contraction of 10, stays on the tiled kernel: it is
//   write-bound there (76 us for 268 MB; 92 us with the relu bits applied in the store); a vector-ALU kernel with the
//   weights in registers measured 115-140 us in three variants.)
//  * gemm_narrow_tn_kernel: C (Mc x N) = A^T B with A (K x Mc) and B (K x N) both token-major, N <= 16, K = tokens: the
//    same tiles, both fragments plain 4-byte loads (A's: 64 contiguous bytes per token and row block); a wave owns 256
//    rows of the output and walks its share of the tokens.  Waves of a workgroup are combined through LDS, workgroups
//    through the split-K slabs of csrc/gemm.hip (fixed order: deterministic).
#include "common.h"
#include <stdint.h>
#include <algorithm>

// 64 rows per workgroup, the contraction split over its four waves.  A wave stages 64 rows x 32 k of A into its own LDS
// strip with coalesced 16-byte loads (no workgroup barrier) and multiplies it with v_mfma_f32_16x16x4_f32: the output's <= 16
// columns are exactly one such tile wide, so nothing is padded to 32.  B's fragment for a step (k = 4 s + lane / 16, column
// lane % 16) is one 4-byte load per lane from L2, requested a chunk ahead like A.  Per chunk and wave: 8 + 8 global loads,
// 32 LDS reads (conflict-free: row stride 36 floats), 32 MFMAs.
// (Tried first on the vector ALUs: lanes splitting the contraction + a shuffle butterfly, 97 us; one row per lane with B
//  through scalar loads, 89-95 us -- the scalar loads miss their cache (B is 40 KB) and share a counter with the LDS reads,
//  so they cannot be requested ahead; 159 us with the prefetch written out.)
typedef float f32x4n __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void gemm_narrow_nn_kernel(const float* __restrict__ A, int64_t lda,
                                                             const float* __restrict__ B, int64_t b_rs, int64_t b_cs,
                                                             const float* __restrict__ bias, float* __restrict__ C,
                                                             int64_t ldc, int M, int N, int K) {
  constexpr int LD = 36;                                               // floats per staged row
  __shared__ __attribute__((aligned(16))) float strip[4][64 * LD];
  __shared__ float red[3][64 * 16];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int li = lane & 15, lq = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * 64;
  const int kq = ((K + 3) / 4 + 31) / 32 * 32;                         // contraction range of a wave
  const int k0 = min(K, wave * kq), k1 = min(K, k0 + kq);
  float* mystrip = strip[wave];
  // staging map: piece p = lane + 64 j (j < 8): row p / 8, float4 p % 8
  const float* src[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int p = lane + 64 * j;
    const int64_t r = row0 + (p >> 3);
    src[j] = A + (r < M ? r : (int64_t)M - 1) * lda + 4 * (p & 7);     // (rows past the end: loaded, never stored)
  }
  const float* bsrc = B + (int64_t)lq * b_rs + (int64_t)li * b_cs;     // B(k = lq, n = li)
  const bool bcol = li < N;
  f32x4n acc[4];
#pragma unroll
  for (int rb = 0; rb < 4; ++rb) acc[rb] = f32x4n{0.f, 0.f, 0.f, 0.f};
  float4 na[8];
  float nb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
    na[j] = (k0 + 4 * ((lane + 64 * j) & 7) < k1) ? *reinterpret_cast<const float4*>(src[j] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int s2 = 0; s2 < 8; ++s2) nb[s2] = (bcol && k0 + 4 * s2 + lq < k1) ? bsrc[(int64_t)(k0 + 4 * s2) * b_rs] : 0.f;
  for (int kc = k0; kc < k1; kc += 32) {
    float wb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int p = lane + 64 * j;
      *reinterpret_cast<float4*>(mystrip + (p >> 3) * LD + 4 * (p & 7)) = na[j];
    }
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) wb[s2] = nb[s2];
    if (kc + 32 < k1) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        na[j] = (kc + 32 + 4 * ((lane + 64 * j) & 7) < k1) ? *reinterpret_cast<const float4*>(src[j] + kc + 32)
                                                           : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int s2 = 0; s2 < 8; ++s2)
        nb[s2] = (bcol && kc + 32 + 4 * s2 + lq < k1) ? bsrc[(int64_t)(kc + 32 + 4 * s2) * b_rs] : 0.f;
    }
    // A fragment of a step: row 16 rb + li, k = 4 s + lq (contraction steps past k1 hold zeros in both operands)
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) {
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        const float a = mystrip[(16 * rb + li) * LD + 4 * s2 + lq];
        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wb[s2], acc[rb], 0, 0, 0);
      }
    }
  }
  // accumulator map of the 16 x 16 tile: column li, rows 4 lq + j
  if (wave > 0) {
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[wave - 1][(16 * rb + 4 * lq + j) * 16 + li] = acc[rb][j];
  }
  __syncthreads();
  if (wave == 0 && bcol) {
    const float bv = bias ? bias[li] : 0.f;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 16 * rb + 4 * lq + j;
        if (row0 + r < M)
          C[(row0 + r) * ldc + li] = ((acc[rb][j] + red[0][r * 16 + li]) + (red[1][r * 16 + li] + red[2][r * 16 + li])) + bv;
      }
  }
}

// slab[blockIdx.x] (Mc x N) = sum over this workgroup's tokens of A[t][m] * B[t][n], on 16 x 16 x 4 MFMA tiles: four tokens
// per step; the A fragment of a 16-row block of the output is A[t0 + lane / 16][m + lane % 16] -- a plain 4-byte load,
// 64 contiguous bytes per token -- and the B fragment B[t0 + lane / 16][lane % 16], shared by the wave's 16 row blocks
// (256 rows of the output per wave).  Eight tokens' fragments are requested before the previous eight are multiplied.
// (On the vector ALUs, B's row through scalar loads: 75-85 us for 268 MB.)
__global__ __launch_bounds__(512) void gemm_narrow_tn_kernel(const float* __restrict__ A, int64_t a_cs,
                                                             const float* __restrict__ B, int64_t b_rs,
                                                             float* __restrict__ slabs, int Mc, int N, int K, int kps) {
  __shared__ float red[7][16 * 64 * 4];                               // (112 KB: one workgroup per CU, as planned)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int li = lane & 15, lq = lane >> 4;
  const int m0 = blockIdx.y * 256;
  const int t0 = blockIdx.x * kps, t1 = min(K, t0 + kps);
  const int per = ((t1 - t0 + 7) / 8 + 7) / 8 * 8;                     // tokens of a wave: a multiple of 8
  const int w0 = min(t1, t0 + wave * per), w1 = min(t1, w0 + per);
  f32x4n acc[16];
#pragma unroll
  for (int mb = 0; mb < 16; ++mb) acc[mb] = f32x4n{0.f, 0.f, 0.f, 0.f};
  // columns of A past Mc: clamped (their accumulators are never stored)
  int acol[16];
#pragma unroll
  for (int mb = 0; mb < 16; ++mb) acol[mb] = min(m0 + 16 * mb + li, Mc - 1);
  const bool bcol = li < N;
  float ca[2][16], cb[2], na[2][16], nb[2];
  auto fetch = [&](int t, float (&fa)[2][16], float (&fb)[2]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int tt = t + 4 * h + lq;
      const bool ok = tt < w1;
      const float* ar = A + (int64_t)(ok ? tt : w1 - 1) * a_cs;
#pragma unroll
      for (int mb = 0; mb < 16; ++mb) { const float v = ar[acol[mb]]; fa[h][mb] = ok ? v : 0.f; }
      fb[h] = (ok && bcol) ? B[(int64_t)tt * b_rs + li] : 0.f;
    }
  };
  if (w0 < w1) {
    fetch(w0, ca, cb);
    for (int t = w0; t < w1; t += 8) {
      if (t + 8 < w1) fetch(t + 8, na, nb);
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int mb = 0; mb < 16; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[h][mb], cb[h], acc[mb], 0, 0, 0);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        cb[h] = nb[h];
#pragma unroll
        for (int mb = 0; mb < 16; ++mb) ca[h][mb] = na[h][mb];
      }
    }
  }
  // accumulator map of a 16 x 16 tile: column li, rows 4 lq + j
  if (wave > 0) {
#pragma unroll
    for (int mb = 0; mb < 16; ++mb)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[wave - 1][(mb * 4 + j) * 64 + lane] = acc[mb][j];
  }
  __syncthreads();
  if (wave == 0 && bcol) {
    float* out = slabs + (int64_t)blockIdx.x * Mc * N;
#pragma unroll
    for (int mb = 0; mb < 16; ++mb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = m0 + 16 * mb + 4 * lq + j;
        float v = acc[mb][j];
#pragma unroll
        for (int w = 0; w < 7; ++w) v += red[w][(mb * 4 + j) * 64 + lane];
        if (m < Mc) out[(int64_t)m * N + li] = v;
      }
  }
}

// ---- launchers (called from pdn_gemm_f32's routing, csrc/gemm.hip) ----------------------------------------------------
bool pdn_gemm_narrow_nn_ok(int M, int N, int K, int64_t a_rs, int64_t a_cs, const void* A) {
  return N >= 1 && N <= 16 && M >= 4096 && K >= 64 && K % 4 == 0 && a_cs == 1 && a_rs % 4 == 0 &&
         ((uintptr_t)A & 15) == 0;
}

int pdn_gemm_narrow_nn_launch(const float* A, int64_t lda, const float* B, int64_t b_rs, int64_t b_cs, const float* bias,
                              float* C, int64_t ldc, int M, int N, int K, void* stream) {
  const dim3 grid((unsigned)(((int64_t)M + 63) / 64));
  hipLaunchKernelGGL(gemm_narrow_nn_kernel, grid, dim3(256), 0, (hipStream_t)stream, A, lda, B, b_rs, b_cs, bias, C, ldc, M, N, K);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

// splits (= slabs of Mc x N floats the caller reduces) for C (Mc x N) = A^T B over K tokens; 0: shape not taken
int pdn_gemm_narrow_tn_plan(int Mc, int N, int K, int64_t a_rs, int64_t a_cs, int64_t b_cs, const void* A, int64_t ws_cap_floats,
                            int* kps) {
  if (!(N >= 1 && N <= 16 && Mc >= 64 && Mc % 4 == 0 && K >= 4096 && a_rs == 1 && a_cs % 4 == 0 && b_cs == 1 &&
        ((uintptr_t)A & 15) == 0))
    return 0;
  const int chunks = (Mc + 255) / 256;
  int want = (256 + chunks - 1) / chunks;                                      // one 8-wave workgroup per CU
  const int64_t room = ws_cap_floats / ((int64_t)Mc * N);
  if (want > room) want = (int)room;
  if (want > K / 256) want = K / 256;
  if (want < 1) return 0;
  *kps = ((K + want - 1) / want + 7) / 8 * 8;
  return (K + *kps - 1) / *kps;
}

int pdn_gemm_narrow_tn_launch(const float* A, int64_t a_cs, const float* B, int64_t b_rs, float* slabs, int Mc, int N, int K,
                              int kps, int splits, void* stream) {
  const dim3 grid(splits, (Mc + 255) / 256);
  hipLaunchKernelGGL(gemm_narrow_tn_kernel, grid, dim3(512), 0, (hipStream_t)stream, A, a_cs, B, b_rs, slabs, Mc, N, K, kps);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}
