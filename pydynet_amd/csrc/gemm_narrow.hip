// Products with a handful of output columns (gfx950): the classifier head of examples/pydynet/mnist.py:70-78,
// Linear(1024, 10) at batch 65536 -- `x @ W` forward (tensor.py:659) and `x^T @ grad` for dW (tensor.py:675).
//
// Ten columns fill a third of one 32-wide MFMA tile and the tiled kernel still stages both operands through LDS with a
// barrier per 32 contraction steps: 148 us forward / 128 us weight gradient for 268 MB of activations that HBM
// delivers in ~45 us.  These are bandwidth kernels on the vector ALUs instead (10 multiply-adds per loaded float):
//
//  * gemm_narrow_nn_kernel: C (M x N) = A (M x K) B (K x N) + bias, N <= 16.  One row per lane: a wave stages 64 rows x 32 k
//    of A into its own LDS strip (coalesced 16-byte loads, no barrier) and every lane walks its row; B arrives through scalar
//    loads; the four waves of a workgroup split the contraction.  (First version: lanes split the contraction and a
//    shuffle butterfly combined them -- 240 cross-lane operations per four rows: 97 us.)
//  * gemm_narrow_k_kernel: C (M x N) = [bits o] (A (M x K) B (K x N) + existing), K <= 16 (the input gradient below such
//    a head): a thread owns four columns and their weights, walks the rows, 16-byte stores.
//  * gemm_narrow_tn_kernel: C (Mc x N) = A^T B with A (K x Mc) and B (K x N) both token-major, N <= 16, K = tokens.
//    A lane owns four columns of A (= rows of C) and N accumulators for each; the wave walks its share of the tokens,
//    B's row for a token comes through scalar loads (wave-uniform address).  Waves of a workgroup are combined through
//    LDS, workgroups through the split-K slabs of csrc/gemm.hip (fixed order: deterministic).
#include "common.h"
#include <stdint.h>
#include <algorithm>

// 64 rows per workgroup (one per lane), the contraction split over its four waves.  A wave stages 64 rows x 32 k of A into
// its own LDS strip with coalesced 16-byte loads (no workgroup barrier), then every lane walks its row with b128 reads
// while B's rows arrive through scalar loads (the address is wave-uniform).
template <int NMAX>
__global__ __launch_bounds__(256) void gemm_narrow_nn_kernel(const float* __restrict__ A, int64_t lda,
                                                             const float* __restrict__ B, int64_t b_rs, int64_t b_cs,
                                                             const float* __restrict__ bias, float* __restrict__ C,
                                                             int64_t ldc, int M, int N, int K) {
  constexpr int LD = 36;                                               // floats per staged row (16-byte aligned, conflict-free b128)
  __shared__ __attribute__((aligned(16))) float strip[4][64 * LD];
  __shared__ float red[3][64][NMAX];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int64_t row0 = (int64_t)blockIdx.x * 64;
  const int kq = ((K + 3) / 4 + 31) / 32 * 32;                         // contraction range of a wave
  const int k0 = wave * kq, k1 = min(K, k0 + kq);
  float* mystrip = strip[wave];
  // staging map: piece p = lane + 64 j (j < 8): row p / 8, float4 p % 8
  const float* src[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int p = lane + 64 * j;
    const int64_t r = row0 + (p >> 3);
    src[j] = A + (r < M ? r : (int64_t)M - 1) * lda + 4 * (p & 7);     // (rows past the end: loaded, never stored)
  }
  float acc[NMAX];
#pragma unroll
  for (int n = 0; n < NMAX; ++n) acc[n] = 0.f;
  float4 nxt[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
    nxt[j] = (k0 + 4 * ((lane + 64 * j) & 7) < k1) ? *reinterpret_cast<const float4*>(src[j] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
  for (int kc = k0; kc < k1; kc += 32) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int p = lane + 64 * j;
      *reinterpret_cast<float4*>(mystrip + (p >> 3) * LD + 4 * (p & 7)) = nxt[j];
    }
    if (kc + 32 < k1) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        nxt[j] = (kc + 32 + 4 * ((lane + 64 * j) & 7) < k1) ? *reinterpret_cast<const float4*>(src[j] + kc + 32)
                                                            : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float* mine = mystrip + lane * LD;
    const int kn = min(32, k1 - kc);                                   // (a multiple of 4: K is)
    if (b_cs == 1) {                                                   // row-major B: a row's values in wide scalar loads
      // Rows up to `safe` may be read NMAX wide (the excess lands in the following rows of B and feeds accumulators that
      // are never stored); the last row or two take the guarded loop below.  (Requesting round i + 1's rows of B before
      // round i's multiplies does not help: scalar and LDS loads share one counter, so the wait for this round's LDS
      // read is a wait for those as well -- measured 159 us against 89.)
      const int safe = K - 1 - (NMAX + (int)b_rs - 1) / (int)b_rs;
      int kk = 0;
      for (; kk + 4 <= kn && kc + kk + 3 <= safe; kk += 4) {
        const float4 a = *reinterpret_cast<const float4*>(mine + kk);
        const float* w = B + (int64_t)(kc + kk) * b_rs;                // wave-uniform
        float w0[NMAX], w1[NMAX], w2[NMAX], w3[NMAX];
#pragma unroll
        for (int n = 0; n < NMAX; ++n) { w0[n] = w[n]; w1[n] = w[b_rs + n]; w2[n] = w[2 * b_rs + n]; w3[n] = w[3 * b_rs + n]; }
#pragma unroll
        for (int n = 0; n < NMAX; ++n) acc[n] = fmaf(a.w, w3[n], fmaf(a.z, w2[n], fmaf(a.y, w1[n], fmaf(a.x, w0[n], acc[n]))));
      }
      for (; kk < kn; ++kk) {
        const float a = mine[kk];
        const float* w = B + (int64_t)(kc + kk) * b_rs;
#pragma unroll
        for (int n = 0; n < NMAX; ++n)
          if (n < N) acc[n] = fmaf(a, w[n], acc[n]);
      }
    } else {
      for (int kk = 0; kk < kn; kk += 4) {
        const float4 a = *reinterpret_cast<const float4*>(mine + kk);
        const float* w = B + (int64_t)(kc + kk) * b_rs;
#pragma unroll
        for (int n = 0; n < NMAX; ++n) {
          if (n < N) {
            const int64_t o = (int64_t)n * b_cs;
            float t = fmaf(a.x, w[o], acc[n]);
            t = fmaf(a.y, w[o + b_rs], t);
            t = fmaf(a.z, w[o + 2 * b_rs], t);
            acc[n] = fmaf(a.w, w[o + 3 * b_rs], t);
          }
        }
      }
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int n = 0; n < NMAX; ++n) red[wave - 1][lane][n] = acc[n];
  }
  __syncthreads();
  if (wave == 0 && row0 + lane < M) {
    float* out = C + (row0 + lane) * ldc;
#pragma unroll
    for (int n = 0; n < NMAX; ++n)
      if (n < N) out[n] = ((acc[n] + red[0][lane][n]) + (red[1][lane][n] + red[2][lane][n])) + (bias ? bias[n] : 0.f);
  }
}

// dx (M x N) = [bits o] (g (M x K) W (K x N) + existing) for a contraction of at most 16 (the input gradient below a
// classifier head): a write-bound kernel.  A thread owns four output columns and their K x 4 weights in registers and walks
// the workgroup's rows, whose g values sit in LDS (one coalesced load; every lane reads the same address: a broadcast).
// (Scalar loads for g: each row's loads were waited for before its multiplies -- 123 us for 268 MB.)
template <int KMAX>
__global__ __launch_bounds__(256) void gemm_narrow_k_kernel(const float* __restrict__ G, int64_t g_rs,
                                                            const float* __restrict__ W, int64_t w_rs, int64_t w_cs,
                                                            const float* __restrict__ existing, const uint32_t* __restrict__ bits,
                                                            float* __restrict__ C, int64_t ldc, int M, int N, int K, int rows_per) {
  extern __shared__ __attribute__((aligned(16))) float gs[];          // [rows_per][KMAX]
  const int m0 = blockIdx.x * rows_per, m1 = min(M, m0 + rows_per);
  for (int i = threadIdx.x; i < (m1 - m0) * KMAX; i += 256) {
    const int r = i / KMAX, k = i - r * KMAX;
    gs[i] = k < K ? G[(int64_t)(m0 + r) * g_rs + k] : 0.f;
  }
  const int n = (blockIdx.y * 256 + threadIdx.x) * 4;
  const bool live = n < N;
  float4 w[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    if (k < K && live) {
      const float* q = W + (int64_t)k * w_rs + (int64_t)n * w_cs;
      w[k] = make_float4(q[0], q[w_cs], q[2 * w_cs], q[3 * w_cs]);
    } else {
      w[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __syncthreads();
  if (!live) return;
  for (int m = m0; m < m1; m += 4) {                                   // four rows per step: their loads issued together
    float4 v[4];
    uint32_t b[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int mr = min(m + r, m1 - 1);
      v[r] = existing ? *reinterpret_cast<const float4*>(existing + (int64_t)mr * ldc + n) : make_float4(0.f, 0.f, 0.f, 0.f);
      b[r] = bits ? bits[(int64_t)mr * (N >> 5) + (n >> 5)] >> (n & 31) : 0xfu;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* g = gs + (min(m + r, m1 - 1) - m0) * KMAX;
#pragma unroll
      for (int k4 = 0; k4 < KMAX; k4 += 4) {
        const float4 gq = *reinterpret_cast<const float4*>(g + k4);
        v[r].x = fmaf(gq.x, w[k4].x, v[r].x); v[r].y = fmaf(gq.x, w[k4].y, v[r].y); v[r].z = fmaf(gq.x, w[k4].z, v[r].z); v[r].w = fmaf(gq.x, w[k4].w, v[r].w);
        v[r].x = fmaf(gq.y, w[k4 + 1].x, v[r].x); v[r].y = fmaf(gq.y, w[k4 + 1].y, v[r].y); v[r].z = fmaf(gq.y, w[k4 + 1].z, v[r].z); v[r].w = fmaf(gq.y, w[k4 + 1].w, v[r].w);
        v[r].x = fmaf(gq.z, w[k4 + 2].x, v[r].x); v[r].y = fmaf(gq.z, w[k4 + 2].y, v[r].y); v[r].z = fmaf(gq.z, w[k4 + 2].z, v[r].z); v[r].w = fmaf(gq.z, w[k4 + 2].w, v[r].w);
        v[r].x = fmaf(gq.w, w[k4 + 3].x, v[r].x); v[r].y = fmaf(gq.w, w[k4 + 3].y, v[r].y); v[r].z = fmaf(gq.w, w[k4 + 3].z, v[r].z); v[r].w = fmaf(gq.w, w[k4 + 3].w, v[r].w);
      }
      v[r].x = (b[r] & 1u) ? v[r].x : 0.f; v[r].y = (b[r] & 2u) ? v[r].y : 0.f;
      v[r].z = (b[r] & 4u) ? v[r].z : 0.f; v[r].w = (b[r] & 8u) ? v[r].w : 0.f;
      if (m + r < m1) *reinterpret_cast<float4*>(C + (int64_t)(m + r) * ldc + n) = v[r];
    }
  }
}

// slab[blockIdx.x] (Mc x N) = sum over this workgroup's tokens of A[t][m] * B[t][n]
template <int NMAX>
__global__ __launch_bounds__(512) void gemm_narrow_tn_kernel(const float* __restrict__ A, int64_t a_cs,
                                                             const float* __restrict__ B, int64_t b_rs,
                                                             float* __restrict__ slabs, int Mc, int N, int K, int kps) {
  __shared__ float4 red[7][64];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int m = blockIdx.y * 256 + 4 * lane;
  const bool live = m < Mc;
  const int t0 = blockIdx.x * kps, t1 = min(K, t0 + kps);
  const int per = (t1 - t0 + 7) / 8;
  const int w0 = min(t1, t0 + wave * per), w1 = min(t1, w0 + per);
  float acc[4][NMAX];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int n = 0; n < NMAX; ++n) acc[j][n] = 0.f;
  const float* ap = A + (live ? m : 0);
  // Tokens below `wide` may have their row of B read NMAX wide (see gemm_narrow_nn_kernel).  Two tokens per round, the
  // operands of round i + 1 requested before round i's multiplies.
  const int wide = min(w1, K - (NMAX + (int)b_rs - 1) / (int)b_rs);
  int t = w0;
  if (t + 1 < wide) {
    float4 ca0 = *reinterpret_cast<const float4*>(ap + (int64_t)t * a_cs);
    float4 ca1 = *reinterpret_cast<const float4*>(ap + (int64_t)(t + 1) * a_cs);
    float c0[NMAX], c1[NMAX];
    {
      const float* g = B + (int64_t)t * b_rs;                          // wave-uniform: scalar loads
#pragma unroll
      for (int n = 0; n < NMAX; ++n) { c0[n] = g[n]; c1[n] = g[b_rs + n]; }
    }
    for (; t + 1 < wide; t += 2) {
      const int tn = min(t + 2, wide - 2);                             // (last round: a valid pair, unused)
      const float4 na0 = *reinterpret_cast<const float4*>(ap + (int64_t)tn * a_cs);
      const float4 na1 = *reinterpret_cast<const float4*>(ap + (int64_t)(tn + 1) * a_cs);
      float n0[NMAX], n1[NMAX];
      {
        const float* g = B + (int64_t)tn * b_rs;
#pragma unroll
        for (int n = 0; n < NMAX; ++n) { n0[n] = g[n]; n1[n] = g[b_rs + n]; }
      }
#pragma unroll
      for (int n = 0; n < NMAX; ++n) {
        acc[0][n] = fmaf(ca1.x, c1[n], fmaf(ca0.x, c0[n], acc[0][n])); acc[1][n] = fmaf(ca1.y, c1[n], fmaf(ca0.y, c0[n], acc[1][n]));
        acc[2][n] = fmaf(ca1.z, c1[n], fmaf(ca0.z, c0[n], acc[2][n])); acc[3][n] = fmaf(ca1.w, c1[n], fmaf(ca0.w, c0[n], acc[3][n]));
      }
      ca0 = na0; ca1 = na1;
#pragma unroll
      for (int n = 0; n < NMAX; ++n) { c0[n] = n0[n]; c1[n] = n1[n]; }
    }
  }
  for (; t < w1; ++t) {
    const float4 a = *reinterpret_cast<const float4*>(ap + (int64_t)t * a_cs);
    const float* g = B + (int64_t)t * b_rs;
#pragma unroll
    for (int n = 0; n < NMAX; ++n) {
      if (n < N) {
        const float gv = g[n];
        acc[0][n] = fmaf(a.x, gv, acc[0][n]); acc[1][n] = fmaf(a.y, gv, acc[1][n]);
        acc[2][n] = fmaf(a.z, gv, acc[2][n]); acc[3][n] = fmaf(a.w, gv, acc[3][n]);
      }
    }
  }
  float* out = slabs + (int64_t)blockIdx.x * Mc * N;
#pragma unroll
  for (int n = 0; n < NMAX; ++n) {
    if (n < N) {                                                       // (N is uniform: every wave reaches the barriers)
      if (wave > 0) red[wave - 1][lane] = make_float4(acc[0][n], acc[1][n], acc[2][n], acc[3][n]);
      __syncthreads();
      if (wave == 0 && live) {
        float4 v = make_float4(acc[0][n], acc[1][n], acc[2][n], acc[3][n]);
#pragma unroll
        for (int w = 0; w < 7; ++w) { const float4 q = red[w][lane]; v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
        out[(int64_t)(m + 0) * N + n] = v.x; out[(int64_t)(m + 1) * N + n] = v.y;
        out[(int64_t)(m + 2) * N + n] = v.z; out[(int64_t)(m + 3) * N + n] = v.w;
      }
      __syncthreads();
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
  if (N <= 8)
    hipLaunchKernelGGL((gemm_narrow_nn_kernel<8>), grid, dim3(256), 0, (hipStream_t)stream, A, lda, B, b_rs, b_cs, bias, C, ldc, M, N, K);
  else
    hipLaunchKernelGGL((gemm_narrow_nn_kernel<16>), grid, dim3(256), 0, (hipStream_t)stream, A, lda, B, b_rs, b_cs, bias, C, ldc, M, N, K);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

bool pdn_gemm_narrow_k_ok(int M, int N, int K, int64_t a_cs, int64_t ldc, const void* C, const void* existing) {
  return K >= 1 && K <= 16 && N >= 256 && N % 4 == 0 && M >= 1024 && a_cs == 1 && ldc % 4 == 0 &&
         (((uintptr_t)C | (uintptr_t)existing) & 15) == 0;
}

int pdn_gemm_narrow_k_launch(const float* G, int64_t g_rs, const float* W, int64_t w_rs, int64_t w_cs, const float* existing,
                             const uint32_t* bits, float* C, int64_t ldc, int M, int N, int K, void* stream) {
  const int ny = (N / 4 + 255) / 256;
  // ~512 workgroups: a thread's K x 4 weights are 4 K scattered 4-byte loads (64 cache lines per wave instruction), paid
  // once per workgroup -- at 32 rows per workgroup they cost more than the rows themselves
  int rows_per = (int)(((int64_t)M * ny + 511) / 512);
  if (rows_per < 16) rows_per = 16;
  if (rows_per > 512) rows_per = 512;                                          // (LDS: rows_per x KMAX floats)
  const dim3 grid((M + rows_per - 1) / rows_per, ny);
  if (K <= 8)
    hipLaunchKernelGGL((gemm_narrow_k_kernel<8>), grid, dim3(256), (size_t)rows_per * 8 * 4, (hipStream_t)stream, G, g_rs, W, w_rs, w_cs, existing, bits, C, ldc, M, N, K, rows_per);
  else
    hipLaunchKernelGGL((gemm_narrow_k_kernel<16>), grid, dim3(256), (size_t)rows_per * 16 * 4, (hipStream_t)stream, G, g_rs, W, w_rs, w_cs, existing, bits, C, ldc, M, N, K, rows_per);
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
  int want = (512 + chunks - 1) / chunks;                                      // ~512 workgroups of 8 waves
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
  if (N <= 8)
    hipLaunchKernelGGL((gemm_narrow_tn_kernel<8>), grid, dim3(512), 0, (hipStream_t)stream, A, a_cs, B, b_rs, slabs, Mc, N, K, kps);
  else
    hipLaunchKernelGGL((gemm_narrow_tn_kernel<16>), grid, dim3(512), 0, (hipStream_t)stream, A, a_cs, B, b_rs, slabs, Mc, N, K, kps);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}
