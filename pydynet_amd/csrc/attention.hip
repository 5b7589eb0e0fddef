// Fused causal self-attention for the training path, fp32 MFMA (gfx950).
//
// Replaces the chain of llm/llama/model.py:112-121 --
//     xq.T(0,2,1,3) @ xk.T(0,2,3,1) / sqrt(hd) + mask -> softmax(-1) -> @ xv.T(0,2,1,3) -> T(0,2,1,3)
// -- i.e. 2 batched matmuls, a divide, an add and the 4-node softmax, which materialise a
// (B, H, L, L) score tensor three times in HBM.  Here the scores never leave registers.
//
// Layout: q, k, v, o are (B, L, H, hd) exactly as the projections produce them (row stride
// H*hd, head offset h*hd); nothing is transposed or copied.
//
// One workgroup (8 wave64, two per SIMD) per (batch, head).  K and V of that head are staged in
// LDS once ([L][hd+4] each, all global loads of a thread issued before the first LDS write).  A
// wave owns one 32-query tile; SIMD s hosts waves s and s+4, which take tiles s and 7-s, so the
// causal work (s+1 and 8-s key tiles) is balanced over the four SIMDs and each SIMD always has a
// second wave to issue MFMAs while the other one does its softmax.
//
// Per query tile (32 rows), with the 32x32x2 f32 MFMA:
//   S^T[key][q]  = K Q^T      (A = K rows from LDS, B = Q rows held in registers, k = head dim)
//     -> accumulator layout: lane = query, registers = keys, so the softmax row reductions are
//        in-lane plus ONE cross-half shuffle;
//   P^T = exp(S^T / sqrt(hd) - rowmax), l = rowsum
//   O^T[d][q]    = V^T P^T    (A = V columns from LDS, B = the P^T accumulator registers as they are:
//        inside one MFMA the two half-waves may contract over any two keys as long as A and B
//        agree, so register r of the accumulator pairs key krow(r) / krow(r)+4 with no data movement)
// Fully masked key tiles (key tile > query tile) are skipped; the diagonal tile is masked per
// element with -inf exactly like the reference's additive mask (masked probabilities are 0).
// The head dim (48) is padded to 64 in the O^T product only (two 32-row MFMA tiles).
// Saved for backward: lse[b,h,q] = rowmax + log(rowsum).
#include "common.h"
#include <stdlib.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define ATT_MAX_TILES 8      // L <= 256
#define ATT_LD(hd) ((hd) + 4)

__device__ __forceinline__ int att_krow(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

// zig-zag owner of query tile i among 4 waves
__device__ __forceinline__ int att_owner(int i) { return ((i >> 2) & 1) ? 3 - (i & 3) : (i & 3); }

// zig-zag owner: wave w (8 per workgroup) takes tile w (w < 4) or 11 - w; SIMD s = w % 4 hosts tiles s and 7 - s
__device__ __forceinline__ int att_tile_of_wave(int w) { return w < 4 ? w : 11 - w; }

// RoPE on one float4 = two interleaved pairs (llm/llama/model.py:23-44): (r, i) -> (r c - i s, r s + i c);
// sign = -1 rotates back (the gradient).  cs / sn: (L, HD/2) tables, `pair0` even.
__device__ __forceinline__ float4 att_rot(float4 v, const float* __restrict__ cs, const float* __restrict__ sn,
                                          int pos, int pair0, int half, float sign) {
  const float2 c = *reinterpret_cast<const float2*>(cs + pos * half + pair0);
  float2 s = *reinterpret_cast<const float2*>(sn + pos * half + pair0);
  s.x *= sign; s.y *= sign;
  float4 o;
  o.x = v.x * c.x - v.y * s.x; o.y = v.x * s.x + v.y * c.x;
  o.z = v.z * c.y - v.w * s.y; o.w = v.z * s.y + v.w * c.y;
  return o;
}

// Stage two [L][HD] row-major matrices (row stride `row_stride` floats) into padded LDS images
// [L][HD+4]; every thread issues all of its global loads before the first LDS write.
// `rot0` / `rot1` rotate rows of matrix 0 / 1 by their position (RoPE fused into the load) when the
// tables are given.
template <int HD, int NT>
__device__ __forceinline__ void att_stage_two(float* __restrict__ s0, float* __restrict__ s1,
                                              const float* __restrict__ g0, const float* __restrict__ g1,
                                              int L, int64_t row_stride, int64_t row_stride1, int tid,
                                              const float* __restrict__ cs, const float* __restrict__ sn,
                                              bool rot0, bool rot1) {
  constexpr int LD = ATT_LD(HD), F4 = HD / 4;
  constexpr int NP = (ATT_MAX_TILES * 32 * F4 + NT - 1) / NT;
  float4 r0[NP], r1[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int u = tid + NT * j;
    if (u < L * F4) {
      const int row = u / F4, c4 = u % F4;
      // component-wise: a whole-float4 store into the array defeats SROA (scratch) on hipcc 7.2
      float4 a = *reinterpret_cast<const float4*>(g0 + (int64_t)row * row_stride + 4 * c4);
      float4 c = *reinterpret_cast<const float4*>(g1 + (int64_t)row * row_stride1 + 4 * c4);
      if (cs && rot0) a = att_rot(a, cs, sn, row, 2 * c4, HD / 2, 1.f);
      if (cs && rot1) c = att_rot(c, cs, sn, row, 2 * c4, HD / 2, 1.f);
      r0[j].x = a.x; r0[j].y = a.y; r0[j].z = a.z; r0[j].w = a.w;
      r1[j].x = c.x; r1[j].y = c.y; r1[j].z = c.z; r1[j].w = c.w;
    }
  }
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int u = tid + NT * j;
    if (u < L * F4) {
      const int row = u / F4, c4 = u % F4;
      *reinterpret_cast<float4*>(s0 + row * LD + 4 * c4) = r0[j];
      *reinterpret_cast<float4*>(s1 + row * LD + 4 * c4) = r1[j];
    }
  }
}

// Forward staging: K as [L][HD+4]; V as [L][64] with column HD = 1 and columns HD+1 .. 63 = 0.  The second 32-row
// tile of O^T = V^T P^T then needs no per-lane select for the rows beyond HD, and its row HD accumulates the
// softmax denominator (the row sums of P) for free -- the padded tile is multiplied anyway.
#define ATT_LDV 64
template <int HD, int NT>
__device__ __forceinline__ void att_stage_kv(float* __restrict__ ks, float* __restrict__ vs,
                                             const float* __restrict__ gk, const float* __restrict__ gv,
                                             int L, int64_t row_stride, int tid,
                                             const float* __restrict__ cs, const float* __restrict__ sn) {
  constexpr int LD = ATT_LD(HD), F4 = HD / 4;
  constexpr int NP = (ATT_MAX_TILES * 32 * F4 + NT - 1) / NT;
  float4 r0[NP], r1[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int u = tid + NT * j;
    if (u < L * F4) {
      const int row = u / F4, c4 = u % F4;
      float4 a = *reinterpret_cast<const float4*>(gk + (int64_t)row * row_stride + 4 * c4);
      const float4 c = *reinterpret_cast<const float4*>(gv + (int64_t)row * row_stride + 4 * c4);
      if (cs) a = att_rot(a, cs, sn, row, 2 * c4, HD / 2, 1.f);
      r0[j].x = a.x; r0[j].y = a.y; r0[j].z = a.z; r0[j].w = a.w;
      r1[j].x = c.x; r1[j].y = c.y; r1[j].z = c.z; r1[j].w = c.w;
    }
  }
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int u = tid + NT * j;
    if (u < L * F4) {
      const int row = u / F4, c4 = u % F4;
      *reinterpret_cast<float4*>(ks + row * LD + 4 * c4) = r0[j];
      *reinterpret_cast<float4*>(vs + row * ATT_LDV + 4 * c4) = r1[j];
    }
  }
  constexpr int PF4 = (ATT_LDV - HD) / 4;                 // pad units per row
  for (int u = tid; u < L * PF4; u += NT) {
    const int row = u / PF4, pc = u % PF4;
    *reinterpret_cast<float4*>(vs + row * ATT_LDV + HD + 4 * pc) = make_float4(pc == 0 ? 1.f : 0.f, 0.f, 0.f, 0.f);
  }
}

// Backward staging: two [L][HD] matrices as [L][ATT_LDP] images whose columns HD .. 63 are ZERO, so that the second
// 32-row MFMA tile over the head dim reads its operand rows HD .. 63 unconditionally (no per-lane select per MFMA).
// ATT_LDP = 68: 272-byte rows keep both access patterns conflict free (ds_read_b128 fragments along the row for
// 16 consecutive rows, ds_read_b32 across 32 consecutive columns).
#define ATT_LDP 68
template <int HD, int NT, bool PAD0, bool PAD1>
__device__ __forceinline__ void att_stage_two_pad(float* __restrict__ s0, float* __restrict__ s1,
                                                  const float* __restrict__ g0, const float* __restrict__ g1,
                                                  int L, int64_t row_stride, int64_t row_stride1, int tid,
                                                  const float* __restrict__ cs, const float* __restrict__ sn,
                                                  bool rot0, bool rot1) {
  constexpr int LD0 = PAD0 ? ATT_LDP : ATT_LD(HD), LD1 = PAD1 ? ATT_LDP : ATT_LD(HD), F4 = HD / 4;
  constexpr int NP = (ATT_MAX_TILES * 32 * F4 + NT - 1) / NT;
  float4 r0[NP], r1[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int u = tid + NT * j;
    if (u < L * F4) {
      const int row = u / F4, c4 = u % F4;
      float4 a = *reinterpret_cast<const float4*>(g0 + (int64_t)row * row_stride + 4 * c4);
      float4 c = *reinterpret_cast<const float4*>(g1 + (int64_t)row * row_stride1 + 4 * c4);
      if (cs && rot0) a = att_rot(a, cs, sn, row, 2 * c4, HD / 2, 1.f);
      if (cs && rot1) c = att_rot(c, cs, sn, row, 2 * c4, HD / 2, 1.f);
      r0[j].x = a.x; r0[j].y = a.y; r0[j].z = a.z; r0[j].w = a.w;
      r1[j].x = c.x; r1[j].y = c.y; r1[j].z = c.z; r1[j].w = c.w;
    }
  }
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int u = tid + NT * j;
    if (u < L * F4) {
      const int row = u / F4, c4 = u % F4;
      *reinterpret_cast<float4*>(s0 + row * LD0 + 4 * c4) = r0[j];
      *reinterpret_cast<float4*>(s1 + row * LD1 + 4 * c4) = r1[j];
    }
  }
  constexpr int PF4 = (64 - HD) / 4;
  for (int u = tid; u < L * PF4; u += NT) {
    const int row = u / PF4, pc = u % PF4;
    if (PAD0) *reinterpret_cast<float4*>(s0 + row * LD0 + HD + 4 * pc) = make_float4(0.f, 0.f, 0.f, 0.f);
    if (PAD1) *reinterpret_cast<float4*>(s1 + row * LD1 + HD + 4 * pc) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// X^T tiles t0 / t1 (lane = row, register r = head-dim index (r & 3) + 8 (r >> 2) + 4 h) -> the lane's own row of a
// (rows, HD) matrix: four consecutive head-dim values per register group, stored as 16-byte pieces straight from the
// accumulators.  `rowp` = first element of the lane's row + 4 h.  cs / sn: rotate back (the gradient of RoPE).
template <int HD>
__device__ __forceinline__ void att_store_rows(const f32x16& t0, const f32x16& t1, float* __restrict__ rowp, int lh,
                                               float scale, const float* __restrict__ cs = nullptr,
                                               const float* __restrict__ sn = nullptr, int pos = 0) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float4 v = make_float4(t0[4 * g] * scale, t0[4 * g + 1] * scale, t0[4 * g + 2] * scale, t0[4 * g + 3] * scale);
    if (cs) v = att_rot(v, cs, sn, pos, 4 * g + 2 * lh, HD / 2, -1.f);
    *reinterpret_cast<float4*>(rowp + 8 * g) = v;
  }
#pragma unroll
  for (int g = 0; g < (HD - 32) / 8; ++g) {
    float4 v = make_float4(t1[4 * g] * scale, t1[4 * g + 1] * scale, t1[4 * g + 2] * scale, t1[4 * g + 3] * scale);
    if (cs) v = att_rot(v, cs, sn, pos, 16 + 4 * g + 2 * lh, HD / 2, -1.f);
    *reinterpret_cast<float4*>(rowp + 32 + 8 * g) = v;
  }
}

// ABLATE (tools/micro/attn_ablate.hip only; 0 in the library): 1 = no K/V staging, 2 = no S^T MFMAs,
// 4 = no softmax arithmetic, 8 = no PV MFMAs, 16 = no output store -- timing experiments, wrong results.
template <int HD, int ABLATE = 0>
__global__ __launch_bounds__(512, 1) void attention_fwd_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
    float* __restrict__ O, float* __restrict__ LSE, int H, int L, int64_t row_stride,
    int64_t batch_stride, int64_t o_row_stride, int64_t o_batch_stride, float sqrt_hd, int causal,
    const float* __restrict__ RC, const float* __restrict__ RS) {
  constexpr int LD = ATT_LD(HD);
  constexpr int NT8 = HD / 8;                    // k-groups of 8 along the head dim
  constexpr int F4 = HD / 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Ks = lds;                               // [L][LD]
  float* Vs = lds + (size_t)L * LD;              // [L][64]: V | 1 | 0 ...

  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int64_t base = (int64_t)b * batch_stride + (int64_t)h * HD;
  const float* Qb = Q + base;
  float* Ob = O + (int64_t)b * o_batch_stride + (int64_t)h * HD;

  if (!(ABLATE & 1)) att_stage_kv<HD, 512>(Ks, Vs, K + base, V + base, L, row_stride, tid, RC, RS);
  __syncthreads();

  const int ntile = L / 32;
  const int qt = att_tile_of_wave(wave);
  if (qt >= ntile) return;                       // no workgroup barrier below this point
  const float inv_sqrt = 1.f / sqrt_hd;
  {
    const int nk = causal ? qt + 1 : ntile;       // key tiles that can be unmasked
    // Q fragments: lane (li, lh) holds Q[q = qt*32+li][8t + 4lh .. +3]
    float4 qf[NT8];
    {
      const float* qrow = Qb + (int64_t)(qt * 32 + li) * row_stride + 4 * lh;
#pragma unroll
      for (int t = 0; t < NT8; ++t) {
        qf[t] = *reinterpret_cast<const float4*>(qrow + 8 * t);
        if (RC) qf[t] = att_rot(qf[t], RC, RS, qt * 32 + li, 4 * t + 2 * lh, HD / 2, 1.f);
      }
    }
    // ---- S^T tiles ---------------------------------------------------------------------
    f32x16 s[ATT_MAX_TILES];
#pragma unroll
    for (int kt = 0; kt < ATT_MAX_TILES; ++kt) {
      if (kt < nk && !(ABLATE & 2)) {
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float* krow = Ks + (kt * 32 + li) * LD + 4 * lh;
#pragma unroll
        for (int t = 0; t < NT8; ++t) {
          const float4 kf = *reinterpret_cast<const float4*>(krow + 8 * t);
          // (the first product of a tile takes the constant 0 as its accumulator input: no 16 v_mov per tile)
          s[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[t].x, t == 0 ? zero16 : s[kt], 0, 0, 0);
          s[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[t].y, s[kt], 0, 0, 0);
          s[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[t].z, s[kt], 0, 0, 0);
          s[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[t].w, s[kt], 0, 0, 0);
        }
      }
    }
    // ---- mask, softmax over keys (per lane = per query) ---------------------------------------
    // The 1/sqrt(hd) scale rides inside the exponential: p = exp2(s * c1 - max(s) * c1), c1 = log2(e) / sqrt(hd)
    // (the row maximum is taken on the unscaled scores; the scale is positive), and only the diagonal key tile
    // needs the causal compare -- four VALU operations per score instead of eight, in kernels whose waves are
    // bound by their own instruction stream.
    const int qpos = qt * 32 + li;
    const float c1 = inv_sqrt * 1.4426950408889634f;
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < ATT_MAX_TILES; ++kt) {
      if (kt < nk && !(ABLATE & 4)) {
        if (causal && kt == qt) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (kt * 32 + att_krow(r, lh) > qpos) s[kt][r] = -INFINITY;
        }
#pragma unroll
        for (int r = 0; r < 16; r += 2) m = fmaxf(m, fmaxf(s[kt][r], s[kt][r + 1]));      // v_max3_f32
      }
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float c2 = -m * c1;
#pragma unroll
    for (int kt = 0; kt < ATT_MAX_TILES; ++kt) {
      if (kt < nk && !(ABLATE & 4)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kt][r] = __builtin_amdgcn_exp2f(fmaf(s[kt][r], c1, c2));
      }
    }
    m *= inv_sqrt;                                  // the maximum of the SCALED scores (for the log-sum-exp)
    // ---- O^T = V^T P^T  (two 32-row tiles over the head dim, the second half empty for hd=48)
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
#pragma unroll
    for (int kt = 0; kt < ATT_MAX_TILES; ++kt) {
      if (kt < nk && !(ABLATE & 8)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float* vrow = Vs + (kt * 32 + att_krow(r, lh)) * ATT_LDV;
          const float a0 = vrow[li];
          const float a1 = vrow[32 + li];            // columns HD.. of the padded row: 1, 0, 0, ...
          o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, s[kt][r], o0, 0, 0, 0);
          o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, s[kt][r], o1, 0, 0, 0);
        }
      }
    }
    // ---- row HD of O^T (register 4 (HD - 32) / 8 of the lower half-wave's second tile) is the softmax denominator
    static_assert((HD - 32) % 8 == 0 && HD > 32 && HD < 64, "the ones column must land in the lower half-wave");
    float l = o1[(HD - 32) / 2];
    l = __shfl(l, li, 64);
    // ---- normalise and store: a lane holds 4 consecutive head-dim values of its query row per register group ----
    const float inv_l = 1.f / l;
    if (lh == 0) LSE[(int64_t)bh * L + qpos] = m + logf(l);
    if (!(ABLATE & 16)) {
      float* orow = Ob + (int64_t)qpos * o_row_stride + 4 * lh;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(orow + 8 * g) =
            make_float4(o0[4 * g] * inv_l, o0[4 * g + 1] * inv_l, o0[4 * g + 2] * inv_l, o0[4 * g + 3] * inv_l);
#pragma unroll
      for (int g = 0; g < (HD - 32) / 8; ++g)
        *reinterpret_cast<float4*>(orow + 32 + 8 * g) =
            make_float4(o1[4 * g] * inv_l, o1[4 * g + 1] * inv_l, o1[4 * g + 2] * inv_l, o1[4 * g + 3] * inv_l);
    }
  }
}

// chunked variants (defined below)
template <int HD> __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attention_fwd2_kernel(const float*, const float*, const float*, float*, float*, int, int,
                                                        int64_t, int64_t, int64_t, int64_t, float, int, const float*, const float*);
template <int HD> __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attention_bwd_dq2_kernel(const float*, const float*, const float*, const float*, const float*,
                                                           const float*, float*, float*, int, int, int64_t, int64_t, int64_t,
                                                           int64_t, float, int, const float*, const float*);
template <int HD> __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attention_bwd_dkv2_kernel(const float*, const float*, const float*, const float*, const float*,
                                                            const float*, float*, float*, int, int, int64_t, int64_t, int64_t,
                                                            int64_t, float, int, const float*, const float*);
static int64_t pdn_attention_chunk_lds_bytes(int head_dim);
// opt-in (PDN_ATTN_CHUNKED=1, read per call): the chunked kernels measured the SAME times as the whole-head ones
static inline bool att_use_chunked() { return getenv("PDN_ATTN_CHUNKED") != nullptr; }

extern "C" int64_t pdn_attention_lds_bytes(int L, int head_dim) {
  return ((int64_t)2 * L + 8 * 32) * ATT_LD(head_dim) * 4;
}
static int64_t att_fwd_lds_bytes(int L, int head_dim) { return (int64_t)L * (ATT_LD(head_dim) + ATT_LDV) * 4; }

// q, k, v (and dq, dk, dv): (B, L, H, head_dim) contiguous in head_dim, `row_stride` between consecutive
// positions, `batch_stride` between batches -- e.g. column blocks of one packed (B*L, 3*H*hd) projection;
// o (and d_o) have strides of their own.  lse: (B, H, L).  causal: keys > query masked.
extern "C" int pdn_attention_fwd_f32(const float* q, const float* k, const float* v, float* o,
                                     float* lse, int B, int H, int L, int head_dim,
                                     int64_t row_stride, int64_t batch_stride, int64_t o_row_stride,
                                     int64_t o_batch_stride, int causal,
                                     const float* rope_cos, const float* rope_sin, void* stream) {
  if (B == 0 || H == 0 || L == 0) return PDN_OK;
  PDN_CHECK_ARG(q && k && v && o && lse, "pdn_attention_fwd_f32: null operand");
  PDN_CHECK_ARG((rope_cos == nullptr) == (rope_sin == nullptr) &&
                    ((((uintptr_t)rope_cos | (uintptr_t)rope_sin) & 7) == 0),
                "pdn_attention_fwd_f32: rope tables must come as an 8-byte aligned pair");
  if (head_dim != 48 || L % 32 != 0 || L > 32 * ATT_MAX_TILES) {
    pdn_set_error("pdn_attention_fwd_f32: fused path supports head_dim 48, L multiple of 32 and <= %d",
                  32 * ATT_MAX_TILES);
    return PDN_EUNSUPPORTED;
  }
  PDN_CHECK_ARG((row_stride % 4) == 0 && (batch_stride % 4) == 0 && (o_row_stride % 4) == 0 &&
                    (o_batch_stride % 4) == 0 &&
                    ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15) == 0),
                "pdn_attention_fwd_f32: 16-byte alignment required");
  const size_t shm = (size_t)att_fwd_lds_bytes(L, head_dim);
  static bool attr_set = false;
  if (!attr_set) {
    PDN_HIP(hipFuncSetAttribute((const void*)attention_fwd_kernel<48>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  if (causal && att_use_chunked()) {
    // four waves per head, K / V in 128-key chunks, two workgroups per CU
    static bool attr2 = false;
    if (!attr2) {
      PDN_HIP(hipFuncSetAttribute((const void*)attention_fwd2_kernel<48>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr2 = true;
    }
    if (getenv("PDN_ATTN_DEBUG")) {
      int nb = -1;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)attention_fwd2_kernel<48>, 256,
                                                         (size_t)pdn_attention_chunk_lds_bytes(head_dim));
      fprintf(stderr, "attention_fwd2_kernel: %d workgroups per CU at %lld bytes of LDS\n", nb,
              (long long)pdn_attention_chunk_lds_bytes(head_dim));
    }
    hipLaunchKernelGGL((attention_fwd2_kernel<48>), dim3(B * H), dim3(256), (size_t)pdn_attention_chunk_lds_bytes(head_dim),
                       (hipStream_t)stream, q, k, v, o, lse, H, L, row_stride, batch_stride, o_row_stride, o_batch_stride,
                       sqrtf((float)head_dim), causal, rope_cos, rope_sin);
  } else {
    hipLaunchKernelGGL((attention_fwd_kernel<48>), dim3(B * H), dim3(512), shm, (hipStream_t)stream, q, k,
                       v, o, lse, H, L, row_stride, batch_stride, o_row_stride, o_batch_stride,
                       sqrtf((float)head_dim), causal, rope_cos, rope_sin);
  }
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

// ======================================================================================
// Backward.  Probabilities are recomputed from the saved log-sum-exp:  P = exp(S/sqrt(hd) - lse).
//   delta[q] = sum_d dO[q,d] * O[q,d]
//   dV = P^T dO        dP = dO V^T        dS = P o (dP - delta) / sqrt(hd)
//   dQ = dS K          dK = dS^T Q
// An MFMA accumulator holds its column index in the lane and its row index in registers, and
// can feed the next MFMA only as the operand that contracts over the ROW index.  dQ contracts
// over keys, dK / dV over queries, so the score tile is needed in both orientations -- two
// kernels, each one workgroup of 8 waves per (batch, head) with the same zig-zag tile ownership
// as the forward:
//   attention_bwd_dq_kernel   K, V resident in LDS; a wave owns a query tile:
//        S^T[key][q] = K Q^T, dP^T = V dO^T  ->  dQ^T += K^T dS^T;  also writes delta[q]
//   attention_bwd_dkv_kernel  Q, dO resident in LDS; a wave owns a key tile:
//        S[q][key] = Q K^T, dP = dO V^T  ->  dV^T += dO^T P,  dK^T += Q^T dS
// 80 + 112 MFMAs per (query tile, key tile) pair, fully masked pairs skipped.  Nothing of size
// L x L touches HBM; the only intermediate is delta (B*H*L floats of workspace).
// ======================================================================================
template <int HD>
__device__ __forceinline__ void att_store_tile_T(float* slot, const f32x16& t0, const f32x16& t1,
                                                 float* dst_rows, int64_t row_stride, int li, int lh,
                                                 int lane, float scale, const float* __restrict__ cs = nullptr,
                                                 const float* __restrict__ sn = nullptr, int pos0 = 0) {
  // t0/t1 hold X^T[d][row]: lane = row, registers = d.  Stage as [row][d] and write rows.
  constexpr int LD = ATT_LD(HD);
  constexpr int F4 = HD / 4;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int d = att_krow(r, lh);
    slot[li * LD + d] = t0[r] * scale;
    if (32 + d < HD) slot[li * LD + 32 + d] = t1[r] * scale;
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  for (int u = lane; u < 32 * F4; u += 64) {
    const int row = u / F4, c4 = u % F4;
    float4 v = *reinterpret_cast<const float4*>(slot + row * LD + 4 * c4);
    if (cs) v = att_rot(v, cs, sn, pos0 + row, 2 * c4, HD / 2, -1.f);   // gradient of RoPE: rotate back
    *reinterpret_cast<float4*>(dst_rows + (int64_t)row * row_stride + 4 * c4) = v;
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
}

template <int HD>
__global__ __launch_bounds__(512, 1) void attention_bwd_dq_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
    const float* __restrict__ O, const float* __restrict__ dO, const float* __restrict__ LSE,
    float* __restrict__ dQ, float* __restrict__ Delta, int H, int L, int64_t row_stride,
    int64_t batch_stride, int64_t o_row_stride, int64_t o_batch_stride, float sqrt_hd, int causal,
    const float* __restrict__ RC, const float* __restrict__ RS) {
  constexpr int LD = ATT_LD(HD);
  constexpr int NT8 = HD / 8;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int LDK = ATT_LDP;
  float* Ks = lds;                                 // [L][68]: K | 0 (read along the row for S, down the columns for dQ)
  float* Vs = Ks + (size_t)L * LDK;                // [L][LD]

  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int64_t base = (int64_t)b * batch_stride + (int64_t)h * HD;
  const int64_t obase = (int64_t)b * o_batch_stride + (int64_t)h * HD;
  const float* Qb = Q + base; const float* Ob = O + obase; const float* dOb = dO + obase;
  float* dQb = dQ + base;

  att_stage_two_pad<HD, 512, true, false>(Ks, Vs, K + base, V + base, L, row_stride, row_stride, tid, RC, RS, true, false);
  __syncthreads();

  const int ntile = L / 32;
  const int qt = att_tile_of_wave(wave);
  if (qt >= ntile) return;
  const float inv_sqrt = 1.f / sqrt_hd;
  const int nk = causal ? qt + 1 : ntile;
  const int qpos = qt * 32 + li;
  float4 qf[NT8], gf[NT8];
  float dpart = 0.f;
  {
    const float* qrow = Qb + (int64_t)qpos * row_stride + 4 * lh;
    const float* grow = dOb + (int64_t)qpos * o_row_stride + 4 * lh;
    const float* orow = Ob + (int64_t)qpos * o_row_stride + 4 * lh;
#pragma unroll
    for (int t = 0; t < NT8; ++t) {
      qf[t] = *reinterpret_cast<const float4*>(qrow + 8 * t);
      if (RC) qf[t] = att_rot(qf[t], RC, RS, qpos, 4 * t + 2 * lh, HD / 2, 1.f);
      gf[t] = *reinterpret_cast<const float4*>(grow + 8 * t);
      const float4 ov = *reinterpret_cast<const float4*>(orow + 8 * t);
      dpart += (ov.x * gf[t].x + ov.y * gf[t].y) + (ov.z * gf[t].z + ov.w * gf[t].w);
    }
  }
  const float delta_q = dpart + __shfl_xor(dpart, 32, 64);
  const float lse_q = LSE[(int64_t)bh * L + qpos];
  const float c1 = inv_sqrt * 1.4426950408889634f, c2q = -lse_q * 1.4426950408889634f;
  if (lh == 0) Delta[(int64_t)bh * L + qpos] = delta_q;
  f32x16 dq0, dq1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dq0[r] = 0.f; dq1[r] = 0.f; }
  for (int kt = 0; kt < nk; ++kt) {
    f32x16 s, dp;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* krow = Ks + (kt * 32 + li) * LDK + 4 * lh;
    const float* vrow = Vs + (kt * 32 + li) * LD + 4 * lh;
#pragma unroll
    for (int t = 0; t < NT8; ++t) {
      const float4 kf = *reinterpret_cast<const float4*>(krow + 8 * t);
      const float4 vf = *reinterpret_cast<const float4*>(vrow + 8 * t);
      // (the first product of a tile takes the constant 0 as its accumulator input: no 32 v_mov per tile pair)
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[t].x, t == 0 ? zero16 : s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.x, gf[t].x, t == 0 ? zero16 : dp, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[t].y, s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.y, gf[t].y, dp, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[t].z, s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.z, gf[t].z, dp, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[t].w, s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.w, gf[t].w, dp, 0, 0, 0);
    }
    // dS^T[key][q] = P^T o (dP^T - delta_q) / sqrt(hd)   (lane = q);  P = exp2(s * c1 - lse * log2(e)), the causal
    // compare only on the diagonal tile
    if (causal && kt == qt) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kt * 32 + att_krow(r, lh) > qpos) s[r] = -INFINITY;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __builtin_amdgcn_exp2f(fmaf(s[r], c1, c2q));
      s[r] = p * (dp[r] - delta_q);                 // (the 1/sqrt(hd) of dS is applied once, when dQ is stored)
    }
    // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float* kr = Ks + (kt * 32 + att_krow(r, lh)) * LDK;
      const float a0 = kr[li];
      const float a1 = kr[32 + li];                    // columns HD .. 63 of the padded row are zero
      dq0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, s[r], dq0, 0, 0, 0);
      dq1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, s[r], dq1, 0, 0, 0);
    }
  }
  att_store_rows<HD>(dq0, dq1, dQb + (int64_t)qpos * row_stride + 4 * lh, lh, inv_sqrt, RC, RS, qpos);
}

template <int HD>
__global__ __launch_bounds__(512, 1) void attention_bwd_dkv_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
    const float* __restrict__ dO, const float* __restrict__ LSE, const float* __restrict__ Delta,
    float* __restrict__ dK, float* __restrict__ dV, int H, int L, int64_t row_stride,
    int64_t batch_stride, int64_t o_row_stride, int64_t o_batch_stride, float sqrt_hd, int causal,
    const float* __restrict__ RC, const float* __restrict__ RS) {
  constexpr int LD = ATT_LD(HD);
  constexpr int NT8 = HD / 8;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int LDP = ATT_LDP;
  float* Qs = lds;                                 // [L][68]: Q | 0
  float* Gs = Qs + (size_t)L * LDP;                // [L][68]: dO | 0
  float* lse_s = Gs + (size_t)L * LDP;             // [L]
  float* delta_s = lse_s + L;                      // [L]

  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int64_t base = (int64_t)b * batch_stride + (int64_t)h * HD;
  const float* Kb = K + base; const float* Vb = V + base;
  float* dKb = dK + base; float* dVb = dV + base;

  att_stage_two_pad<HD, 512, true, true>(Qs, Gs, Q + base, dO + (int64_t)b * o_batch_stride + (int64_t)h * HD, L,
                                         row_stride, o_row_stride, tid, RC, RS, true, false);
  for (int q = tid; q < L; q += 512) {
    lse_s[q] = LSE[(int64_t)bh * L + q] * 1.4426950408889634f;
    delta_s[q] = Delta[(int64_t)bh * L + q];
  }
  __syncthreads();

  const int ntile = L / 32;
  const int kt = att_tile_of_wave(wave);
  if (kt >= ntile) return;
  const float inv_sqrt = 1.f / sqrt_hd;
  const float c1 = inv_sqrt * 1.4426950408889634f;
  const int kpos = kt * 32 + li;
  float4 kf[NT8], vf[NT8];
  {
    const float* krow = Kb + (int64_t)kpos * row_stride + 4 * lh;
    const float* vrow = Vb + (int64_t)kpos * row_stride + 4 * lh;
#pragma unroll
    for (int t = 0; t < NT8; ++t) {
      kf[t] = *reinterpret_cast<const float4*>(krow + 8 * t);
      if (RC) kf[t] = att_rot(kf[t], RC, RS, kpos, 4 * t + 2 * lh, HD / 2, 1.f);
      vf[t] = *reinterpret_cast<const float4*>(vrow + 8 * t);
    }
  }
  f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dk0[r] = 0.f; dk1[r] = 0.f; dv0[r] = 0.f; dv1[r] = 0.f; }
  const int q_first = causal ? kt : 0;
  for (int qt = q_first; qt < ntile; ++qt) {
    f32x16 s, dp;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* qrow = Qs + (qt * 32 + li) * LDP + 4 * lh;
    const float* grow = Gs + (qt * 32 + li) * LDP + 4 * lh;
#pragma unroll
    for (int t = 0; t < NT8; ++t) {
      const float4 q4 = *reinterpret_cast<const float4*>(qrow + 8 * t);
      const float4 g4 = *reinterpret_cast<const float4*>(grow + 8 * t);
      // (the first product of a tile takes the constant 0 as its accumulator input: no 32 v_mov per tile pair)
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.x, kf[t].x, t == 0 ? zero16 : s, 0, 0, 0);    // S[q][key]
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(g4.x, vf[t].x, t == 0 ? zero16 : dp, 0, 0, 0);  // dP[q][key]
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.y, kf[t].y, s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(g4.y, vf[t].y, dp, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.z, kf[t].z, s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(g4.z, vf[t].z, dp, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.w, kf[t].w, s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(g4.w, vf[t].w, dp, 0, 0, 0);
    }
    // lane = key, registers = queries;  P = exp2(s * c1 - lse * log2(e)) (lse_s holds lse * log2(e)), the causal
    // compare only on the diagonal tile
    if (causal && qt == kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kpos > qt * 32 + att_krow(r, lh)) s[r] = -INFINITY;
    }
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {                      // registers 4 g4 .. 4 g4 + 3 are four consecutive queries
      const int q0 = qt * 32 + 8 * g4 + 4 * lh;
      const float4 ls = *reinterpret_cast<const float4*>(lse_s + q0);
      const float4 ds = *reinterpret_cast<const float4*>(delta_s + q0);
      const float lq[4] = {ls.x, ls.y, ls.z, ls.w}, dq4[4] = {ds.x, ds.y, ds.z, ds.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g4 + e;
        const float p = __builtin_amdgcn_exp2f(fmaf(s[r], c1, -lq[e]));
        s[r] = p;                                        // P[q][key]
        dp[r] = p * (dp[r] - dq4[e]);                    // dS[q][key] * sqrt(hd): the scale is applied when dK is stored
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qr = qt * 32 + att_krow(r, lh);
      const float g0 = Gs[qr * LDP + li], q0 = Qs[qr * LDP + li];
      const float g1 = Gs[qr * LDP + 32 + li], q1 = Qs[qr * LDP + 32 + li];     // zero beyond the head dim
      dv0 = __builtin_amdgcn_mfma_f32_32x32x2f32(g0, s[r], dv0, 0, 0, 0);     // dV^T += dO^T P
      dk0 = __builtin_amdgcn_mfma_f32_32x32x2f32(q0, dp[r], dk0, 0, 0, 0);    // dK^T += Q^T dS
      dv1 = __builtin_amdgcn_mfma_f32_32x32x2f32(g1, s[r], dv1, 0, 0, 0);
      dk1 = __builtin_amdgcn_mfma_f32_32x32x2f32(q1, dp[r], dk1, 0, 0, 0);
    }
  }
  att_store_rows<HD>(dk0, dk1, dKb + (int64_t)kpos * row_stride + 4 * lh, lh, inv_sqrt, RC, RS, kpos);
  att_store_rows<HD>(dv0, dv1, dVb + (int64_t)kpos * row_stride + 4 * lh, lh, 1.f);
}

extern "C" int64_t pdn_attention_bwd_lds_bytes(int L, int head_dim) {
  (void)head_dim;
  return (int64_t)2 * L * ATT_LDP * 4 + (int64_t)2 * L * 4;          // Q | 0 and dO | 0 images + lse, delta
}
static int64_t att_dq_lds_bytes(int L, int head_dim) { return (int64_t)L * (ATT_LDP + ATT_LD(head_dim)) * 4; }

// delta[b, h, q] = sum_d dO * O is produced by the dQ kernel and consumed by the dK/dV kernel
extern "C" int64_t pdn_attention_bwd_workspace_bytes(int B, int H, int L) {
  return (int64_t)B * H * L * 4;
}

extern "C" int pdn_attention_bwd_f32(const float* q, const float* k, const float* v, const float* o,
                                     const float* d_o, const float* lse, float* dq, float* dk,
                                     float* dv, int B, int H, int L, int head_dim,
                                     int64_t row_stride, int64_t batch_stride, int64_t o_row_stride,
                                     int64_t o_batch_stride, int causal,
                                     const float* rope_cos, const float* rope_sin, void* workspace,
                                     int64_t workspace_bytes, void* stream) {
  if (B == 0 || H == 0 || L == 0) return PDN_OK;
  PDN_CHECK_ARG((rope_cos == nullptr) == (rope_sin == nullptr) &&
                    ((((uintptr_t)rope_cos | (uintptr_t)rope_sin) & 7) == 0),
                "pdn_attention_bwd_f32: rope tables must come as an 8-byte aligned pair");
  PDN_CHECK_ARG(q && k && v && o && d_o && lse && dq && dk && dv, "pdn_attention_bwd_f32: null operand");
  if (head_dim != 48 || L % 32 != 0 || L > 32 * ATT_MAX_TILES) {
    pdn_set_error("pdn_attention_bwd_f32: fused path supports head_dim 48, L multiple of 32 and <= %d",
                  32 * ATT_MAX_TILES);
    return PDN_EUNSUPPORTED;
  }
  PDN_CHECK_ARG((row_stride % 4) == 0 && (batch_stride % 4) == 0 && (o_row_stride % 4) == 0 &&
                    (o_batch_stride % 4) == 0 &&
                    ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)d_o |
                       (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15) == 0),
                "pdn_attention_bwd_f32: 16-byte alignment required");
  if (!workspace || workspace_bytes < pdn_attention_bwd_workspace_bytes(B, H, L)) {
    pdn_set_error("pdn_attention_bwd_f32: workspace too small");
    return PDN_EWORKSPACE;
  }
  float* delta = (float*)workspace;
  static bool attr_set = false;
  if (!attr_set) {
    PDN_HIP(hipFuncSetAttribute((const void*)attention_bwd_dq_kernel<48>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PDN_HIP(hipFuncSetAttribute((const void*)attention_bwd_dkv_kernel<48>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const float sq = sqrtf((float)head_dim);
  static bool attr2 = false;
  if (!attr2) {
    PDN_HIP(hipFuncSetAttribute((const void*)attention_bwd_dq2_kernel<48>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PDN_HIP(hipFuncSetAttribute((const void*)attention_bwd_dkv2_kernel<48>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr2 = true;
  }
  const int ntile = L / 32;
  const size_t shm2 = (size_t)pdn_attention_chunk_lds_bytes(head_dim);
  if (causal && att_use_chunked())
    hipLaunchKernelGGL((attention_bwd_dq2_kernel<48>), dim3(B * H), dim3(256), shm2, (hipStream_t)stream, q, k, v, o, d_o,
                       lse, dq, delta, H, L, row_stride, batch_stride, o_row_stride, o_batch_stride, sq, causal,
                       rope_cos, rope_sin);
  else
    hipLaunchKernelGGL((attention_bwd_dq_kernel<48>), dim3(B * H), dim3(512),
                       (size_t)att_dq_lds_bytes(L, head_dim), (hipStream_t)stream, q, k, v, o, d_o,
                       lse, dq, delta, H, L, row_stride, batch_stride, o_row_stride, o_batch_stride, sq, causal,
                       rope_cos, rope_sin);
  PDN_LAUNCH_CHECK();
  // (the chunked dK/dV kernel keeps one key tile's accumulators live at a time: with 5 or 6 tiles the pair
  //  (a, ntile - 1 - a) can have both members start in the first query chunk -- those lengths take the whole-head kernel)
  if (causal && att_use_chunked() && (ntile <= 4 || ntile >= 7))
    hipLaunchKernelGGL((attention_bwd_dkv2_kernel<48>), dim3(B * H), dim3(256), shm2, (hipStream_t)stream, q, k, v, d_o,
                       lse, delta, dk, dv, H, L, row_stride, batch_stride, o_row_stride, o_batch_stride, sq, causal,
                       rope_cos, rope_sin);
  else
    hipLaunchKernelGGL((attention_bwd_dkv_kernel<48>), dim3(B * H), dim3(512),
                       (size_t)pdn_attention_bwd_lds_bytes(L, head_dim), (hipStream_t)stream, q, k, v, d_o,
                       lse, delta, dk, dv, H, L, row_stride, batch_stride, o_row_stride, o_batch_stride, sq, causal,
                       rope_cos, rope_sin);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

// ======================================================================================
// Chunked variants (opt-in, PDN_ATTN_CHUNKED=1): the same three kernels with FOUR waves per (batch, head) and the
// LDS-resident pair -- K,V for forward / dQ, Q,dO for dK/dV -- passing through LDS in chunks of 128 rows (53 KB) instead
// of whole (106 KB + 53 KB of staging = one 160 KB workgroup per CU).  With 80 KB per workgroup TWO workgroups share a CU
// (hipOccupancyMaxActiveBlocksPerMultiprocessor = 2) and run out of phase.  Measured on the benchmark shape (1536 heads,
// L 256, hd 48): forward 205 vs 202 us, backward 670 vs 677 us -- the SAME: either way a SIMD hosts two waves, and what
// limits these kernels is each wave's own dependency chain (LDS operand -> MFMA -> softmax VALU -> MFMA; 49 % of the
// wave cycles wait on s_waitcnt, PMC), which a second workgroup does not shorten.  Four waves per SIMD need <= 128
// VGPRs per wave (the dK/dV kernel holds 144 in accumulators and operands alone): not reachable in fp32 at hd 48.
// A wave owns the tile pair (w, 7 - w) -- nine of the 36 causal tile pairs, as before -- and works through it
// sequentially; odd (batch, head) indices mirror the assignment so that two co-resident workgroups load the four
// SIMDs evenly in every chunk phase.  Forward: tile w needs chunk 0 only; tile 7 - w carries (m, l, O) across the
// chunk switch with ONE online rescale.  Backward recomputes P from the saved log-sum-exp, so its accumulators
// simply carry over.  Results are those of the single-chunk kernels up to the rescale's rounding.
// ======================================================================================
#define ATT_CH 128                      // rows per LDS chunk (four 32-row tiles)

// rows [row0, row0 + nrows) of two [L][HD] matrices -> padded LDS images [nrows][HD+4] (see att_stage_two)
template <int HD, int NT>
__device__ __forceinline__ void att_stage_chunk(float* __restrict__ s0, float* __restrict__ s1,
                                                const float* __restrict__ g0, const float* __restrict__ g1,
                                                int row0, int nrows, int64_t row_stride, int64_t row_stride1, int tid,
                                                const float* __restrict__ cs, const float* __restrict__ sn,
                                                bool rot0, bool rot1) {
  constexpr int LD = ATT_LD(HD), F4 = HD / 4;
  constexpr int NP = (ATT_CH * F4 + NT - 1) / NT;
  float4 r0[NP], r1[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int u = tid + NT * j;
    if (u < nrows * F4) {
      const int row = u / F4, c4 = u % F4;
      float4 a = *reinterpret_cast<const float4*>(g0 + (int64_t)(row0 + row) * row_stride + 4 * c4);
      float4 c = *reinterpret_cast<const float4*>(g1 + (int64_t)(row0 + row) * row_stride1 + 4 * c4);
      if (cs && rot0) a = att_rot(a, cs, sn, row0 + row, 2 * c4, HD / 2, 1.f);
      if (cs && rot1) c = att_rot(c, cs, sn, row0 + row, 2 * c4, HD / 2, 1.f);
      r0[j].x = a.x; r0[j].y = a.y; r0[j].z = a.z; r0[j].w = a.w;
      r1[j].x = c.x; r1[j].y = c.y; r1[j].z = c.z; r1[j].w = c.w;
    }
  }
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int u = tid + NT * j;
    if (u < nrows * F4) {
      const int row = u / F4, c4 = u % F4;
      *reinterpret_cast<float4*>(s0 + row * LD + 4 * c4) = r0[j];
      *reinterpret_cast<float4*>(s1 + row * LD + 4 * c4) = r1[j];
    }
  }
}

// tile pair of wave w (0..3) of workgroup bh: (a, b) with a < b, or b = -1 (a alone), or a = -1 (idle)
__device__ __forceinline__ void att_pair_of_wave(int w, int bh, int ntile, int& a, int& b) {
  const int ww = (bh & 1) ? 3 - w : w;
  a = ww; b = ntile - 1 - ww;
  if (a >= ntile || a > b) { a = -1; b = -1; }
  else if (a == b) b = -1;
}

static int64_t pdn_attention_chunk_lds_bytes(int head_dim) {
  return ((int64_t)2 * ATT_CH + 4 * 32) * ATT_LD(head_dim) * 4 + 2 * ATT_CH * 4;
}

template <int HD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attention_fwd2_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
    float* __restrict__ O, float* __restrict__ LSE, int H, int L, int64_t row_stride,
    int64_t batch_stride, int64_t o_row_stride, int64_t o_batch_stride, float sqrt_hd, int causal,
    const float* __restrict__ RC, const float* __restrict__ RS) {
  constexpr int LD = ATT_LD(HD);
  constexpr int NT8 = HD / 8;
  constexpr int F4 = HD / 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Ks = lds;                               // [ATT_CH][LD]
  float* Vs = lds + ATT_CH * LD;                 // [ATT_CH][LD]
  float* Os = Vs + ATT_CH * LD;                  // 4 waves x [32][LD] output staging

  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int64_t base = (int64_t)b * batch_stride + (int64_t)h * HD;
  const float* Qb = Q + base;
  float* Ob = O + (int64_t)b * o_batch_stride + (int64_t)h * HD;
  const int ntile = L / 32, nchunk = (L + ATT_CH - 1) / ATT_CH;
  int tA, tB;
  att_pair_of_wave(wave, bh, ntile, tA, tB);
  const float inv_sqrt = 1.f / sqrt_hd;
  const bool hi_ok = (32 + li) < HD;
  float* Ow = Os + wave * (32 * LD);

  // running state of the tile being worked on
  float4 qf[NT8];
  f32x16 o0, o1;
  float m_run = -INFINITY, l_run = 0.f;
  int qt = -1;
  auto begin_tile = [&](int t) {
    qt = t;
    const float* qrow = Qb + (int64_t)(qt * 32 + li) * row_stride + 4 * lh;
#pragma unroll
    for (int t8 = 0; t8 < NT8; ++t8) {
      qf[t8] = *reinterpret_cast<const float4*>(qrow + 8 * t8);
      if (RC) qf[t8] = att_rot(qf[t8], RC, RS, qt * 32 + li, 4 * t8 + 2 * lh, HD / 2, 1.f);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    m_run = -INFINITY; l_run = 0.f;
  };
  // key tiles [k_lo, k_hi) of the chunk whose first key tile is c0 (all resident in LDS)
  auto run_chunk = [&](int c0, int k_lo, int k_hi) {
    const int qpos = qt * 32 + li;
    f32x16 s[4];
    float m = m_run;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[j][r] = 0.f;
      const int kt = c0 + j;
      if (kt >= k_lo && kt < k_hi) {
        const float* krow = Ks + (j * 32 + li) * LD + 4 * lh;
#pragma unroll
        for (int t8 = 0; t8 < NT8; ++t8) {
          const float4 kf = *reinterpret_cast<const float4*>(krow + 8 * t8);
          s[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[t8].x, s[j], 0, 0, 0);
          s[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[t8].y, s[j], 0, 0, 0);
          s[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[t8].z, s[j], 0, 0, 0);
          s[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[t8].w, s[j], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = s[j][r] * inv_sqrt;
          if (causal && kt * 32 + att_krow(r, lh) > qpos) v = -INFINITY;
          s[j][r] = v;
          m = fmaxf(m, v);
        }
      }
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    // one rescale of what the earlier chunk left (exp(-inf - m) = 0 on the first chunk)
    const float a = __expf(m_run - m);
    l_run *= a;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] *= a; o1[r] *= a; }
    m_run = m;
    float l = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kt = c0 + j;
      if (kt >= k_lo && kt < k_hi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = __expf(s[j][r] - m);
          s[j][r] = pv;
          l += pv;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float* vrow = Vs + (j * 32 + att_krow(r, lh)) * LD;
          const float a0 = vrow[li];
          const float a1 = hi_ok ? vrow[32 + li] : 0.f;
          o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, s[j][r], o0, 0, 0, 0);
          o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, s[j][r], o1, 0, 0, 0);
        }
      }
    }
    l_run += l + __shfl_xor(l, 32, 64);
  };
  auto end_tile = [&]() {
    const float inv_l = 1.f / l_run;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int d = att_krow(r, lh);
      Ow[li * LD + d] = o0[r] * inv_l;
      if (32 + d < HD) Ow[li * LD + 32 + d] = o1[r] * inv_l;
    }
    if (lh == 0) LSE[(int64_t)bh * L + qt * 32 + li] = m_run + logf(l_run);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    for (int u = lane; u < 32 * F4; u += 64) {
      const int row = u / F4, c4 = u % F4;
      *reinterpret_cast<float4*>(Ob + (int64_t)(qt * 32 + row) * o_row_stride + 4 * c4) =
          *reinterpret_cast<const float4*>(Ow + row * LD + 4 * c4);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  };

  for (int c = 0; c < nchunk; ++c) {
    const int c0 = 4 * c, rows = min(ATT_CH, L - ATT_CH * c);
    if (c > 0) __syncthreads();                    // everyone is done reading the previous chunk
    att_stage_chunk<HD, 256>(Ks, Vs, K + base, V + base, ATT_CH * c, rows, row_stride, row_stride, tid, RC, RS, true, false);
    __syncthreads();
    const int c_end = c0 + rows / 32;              // key tiles [c0, c_end) are resident
    // tile A first (causal: all of its keys lie in the chunk that holds its own rows, so it finishes there),
    // then tile B, which is resumed in the next chunk
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      const int t = which == 0 ? tA : tB;
      if (t < 0) continue;
      const int nk = t + 1;                        // causal: key tiles [0, t] matter for query tile t
      if (nk <= c0) continue;                      // finished in an earlier chunk
      if (qt != t) begin_tile(t);
      run_chunk(c0, c0, min(nk, c_end));
      if (nk <= c_end) end_tile();
    }
  }
}

template <int HD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attention_bwd_dq2_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
    const float* __restrict__ O, const float* __restrict__ dO, const float* __restrict__ LSE,
    float* __restrict__ dQ, float* __restrict__ Delta, int H, int L, int64_t row_stride,
    int64_t batch_stride, int64_t o_row_stride, int64_t o_batch_stride, float sqrt_hd, int causal,
    const float* __restrict__ RC, const float* __restrict__ RS) {
  constexpr int LD = ATT_LD(HD);
  constexpr int NT8 = HD / 8;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Ks = lds;                                 // [ATT_CH][LD]
  float* Vs = Ks + ATT_CH * LD;                    // [ATT_CH][LD]
  float* slots = Vs + ATT_CH * LD;                 // 4 waves x [32][LD]

  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int64_t base = (int64_t)b * batch_stride + (int64_t)h * HD;
  const int64_t obase = (int64_t)b * o_batch_stride + (int64_t)h * HD;
  const float* Qb = Q + base; const float* Ob = O + obase; const float* dOb = dO + obase;
  float* dQb = dQ + base;
  const int ntile = L / 32, nchunk = (L + ATT_CH - 1) / ATT_CH;
  int tA, tB;
  att_pair_of_wave(wave, bh, ntile, tA, tB);
  const float inv_sqrt = 1.f / sqrt_hd;
  const bool hi_ok = (32 + li) < HD;

  float4 qf[NT8], gf[NT8];
  float delta_q = 0.f, lse_q = 0.f;
  f32x16 dq0, dq1;
  int qt = -1;
  auto begin_tile = [&](int t) {
    qt = t;
    const int qpos = qt * 32 + li;
    const float* qrow = Qb + (int64_t)qpos * row_stride + 4 * lh;
    const float* grow = dOb + (int64_t)qpos * o_row_stride + 4 * lh;
    const float* orow = Ob + (int64_t)qpos * o_row_stride + 4 * lh;
    float dpart = 0.f;
#pragma unroll
    for (int t8 = 0; t8 < NT8; ++t8) {
      qf[t8] = *reinterpret_cast<const float4*>(qrow + 8 * t8);
      if (RC) qf[t8] = att_rot(qf[t8], RC, RS, qpos, 4 * t8 + 2 * lh, HD / 2, 1.f);
      gf[t8] = *reinterpret_cast<const float4*>(grow + 8 * t8);
      const float4 ov = *reinterpret_cast<const float4*>(orow + 8 * t8);
      dpart += (ov.x * gf[t8].x + ov.y * gf[t8].y) + (ov.z * gf[t8].z + ov.w * gf[t8].w);
    }
    delta_q = dpart + __shfl_xor(dpart, 32, 64);
    lse_q = LSE[(int64_t)bh * L + qpos];
    if (lh == 0) Delta[(int64_t)bh * L + qpos] = delta_q;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dq0[r] = 0.f; dq1[r] = 0.f; }
  };
  auto run_chunk = [&](int c0, int k_hi) {          // key tiles [c0, k_hi), chunk rows start at tile c0
    const int qpos = qt * 32 + li;
    for (int kt = c0; kt < k_hi; ++kt) {
      const int j = kt - c0;
      f32x16 sv, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sv[r] = 0.f; dp[r] = 0.f; }
      const float* krow = Ks + (j * 32 + li) * LD + 4 * lh;
      const float* vrow = Vs + (j * 32 + li) * LD + 4 * lh;
#pragma unroll
      for (int t8 = 0; t8 < NT8; ++t8) {
        const float4 kf = *reinterpret_cast<const float4*>(krow + 8 * t8);
        const float4 vf = *reinterpret_cast<const float4*>(vrow + 8 * t8);
        sv = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[t8].x, sv, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.x, gf[t8].x, dp, 0, 0, 0);
        sv = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[t8].y, sv, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.y, gf[t8].y, dp, 0, 0, 0);
        sv = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[t8].z, sv, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.z, gf[t8].z, dp, 0, 0, 0);
        sv = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[t8].w, sv, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.w, gf[t8].w, dp, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool masked = causal && (kt * 32 + att_krow(r, lh) > qpos);
        const float pv = masked ? 0.f : __expf(sv[r] * inv_sqrt - lse_q);
        sv[r] = pv * (dp[r] - delta_q) * inv_sqrt;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* kr = Ks + (j * 32 + att_krow(r, lh)) * LD;
        const float a0 = kr[li];
        const float a1 = hi_ok ? kr[32 + li] : 0.f;
        dq0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, sv[r], dq0, 0, 0, 0);
        dq1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, sv[r], dq1, 0, 0, 0);
      }
    }
  };
  for (int c = 0; c < nchunk; ++c) {
    const int c0 = 4 * c, rows = min(ATT_CH, L - ATT_CH * c);
    if (c > 0) __syncthreads();
    att_stage_chunk<HD, 256>(Ks, Vs, K + base, V + base, ATT_CH * c, rows, row_stride, row_stride, tid, RC, RS, true, false);
    __syncthreads();
    const int c_end = c0 + rows / 32;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      const int t = which == 0 ? tA : tB;
      if (t < 0) continue;
      const int nk = t + 1;
      if (nk <= c0) continue;
      if (qt != t) begin_tile(t);
      run_chunk(c0, min(nk, c_end));
      if (nk <= c_end)
        att_store_tile_T<HD>(slots + wave * (32 * LD), dq0, dq1, dQb + (int64_t)(qt * 32) * row_stride,
                             row_stride, li, lh, lane, 1.f, RC, RS, qt * 32);
    }
  }
}

// dK / dV: a wave owns key tiles (w, 7 - w); Q and dO pass through LDS in chunks of 128 QUERY rows.  Key tile t needs
// query tiles t .. ntile - 1: tile A (<= 3) starts in the chunk that holds its own rows and is carried into the
// next one; tile B lies wholly in the last chunk -- so only one tile's accumulators are live at a time.
template <int HD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attention_bwd_dkv2_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
    const float* __restrict__ dO, const float* __restrict__ LSE, const float* __restrict__ Delta,
    float* __restrict__ dK, float* __restrict__ dV, int H, int L, int64_t row_stride,
    int64_t batch_stride, int64_t o_row_stride, int64_t o_batch_stride, float sqrt_hd, int causal,
    const float* __restrict__ RC, const float* __restrict__ RS) {
  constexpr int LD = ATT_LD(HD);
  constexpr int NT8 = HD / 8;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Qs = lds;                                 // [ATT_CH][LD]
  float* Gs = Qs + ATT_CH * LD;                    // [ATT_CH][LD]   dO
  float* slots = Gs + ATT_CH * LD;                 // 4 waves x [32][LD]
  float* lse_s = slots + 4 * 32 * LD;              // [ATT_CH]
  float* delta_s = lse_s + ATT_CH;                 // [ATT_CH]

  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int64_t base = (int64_t)b * batch_stride + (int64_t)h * HD;
  const float* Kb = K + base; const float* Vb = V + base;
  float* dKb = dK + base; float* dVb = dV + base;
  const float* dOb = dO + (int64_t)b * o_batch_stride + (int64_t)h * HD;
  const int ntile = L / 32, nchunk = (L + ATT_CH - 1) / ATT_CH;
  int tA, tB;
  att_pair_of_wave(wave, bh, ntile, tA, tB);
  const float inv_sqrt = 1.f / sqrt_hd;
  const bool hi_ok = (32 + li) < HD;

  float4 kf[NT8], vf[NT8];
  f32x16 dk0, dk1, dv0, dv1;
  int kt = -1;
  auto begin_tile = [&](int t) {
    kt = t;
    const int kpos = kt * 32 + li;
    const float* krow = Kb + (int64_t)kpos * row_stride + 4 * lh;
    const float* vrow = Vb + (int64_t)kpos * row_stride + 4 * lh;
#pragma unroll
    for (int t8 = 0; t8 < NT8; ++t8) {
      kf[t8] = *reinterpret_cast<const float4*>(krow + 8 * t8);
      if (RC) kf[t8] = att_rot(kf[t8], RC, RS, kpos, 4 * t8 + 2 * lh, HD / 2, 1.f);
      vf[t8] = *reinterpret_cast<const float4*>(vrow + 8 * t8);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk0[r] = 0.f; dk1[r] = 0.f; dv0[r] = 0.f; dv1[r] = 0.f; }
  };
  auto run_chunk = [&](int c0, int q_lo, int q_hi) {     // query tiles [q_lo, q_hi), chunk rows start at tile c0
    const int kpos = kt * 32 + li;
    for (int qt = q_lo; qt < q_hi; ++qt) {
      const int j = qt - c0;
      f32x16 sv, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sv[r] = 0.f; dp[r] = 0.f; }
      const float* qrow = Qs + (j * 32 + li) * LD + 4 * lh;
      const float* grow = Gs + (j * 32 + li) * LD + 4 * lh;
#pragma unroll
      for (int t8 = 0; t8 < NT8; ++t8) {
        const float4 q4 = *reinterpret_cast<const float4*>(qrow + 8 * t8);
        const float4 g4 = *reinterpret_cast<const float4*>(grow + 8 * t8);
        sv = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.x, kf[t8].x, sv, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(g4.x, vf[t8].x, dp, 0, 0, 0);
        sv = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.y, kf[t8].y, sv, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(g4.y, vf[t8].y, dp, 0, 0, 0);
        sv = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.z, kf[t8].z, sv, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(g4.z, vf[t8].z, dp, 0, 0, 0);
        sv = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.w, kf[t8].w, sv, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(g4.w, vf[t8].w, dp, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ql = j * 32 + att_krow(r, lh);             // row inside the chunk
        const bool masked = causal && (kpos > c0 * 32 + ql);
        const float pv = masked ? 0.f : __expf(sv[r] * inv_sqrt - lse_s[ql]);
        sv[r] = pv;
        dp[r] = pv * (dp[r] - delta_s[ql]) * inv_sqrt;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qr = j * 32 + att_krow(r, lh);
        const float g0 = Gs[qr * LD + li], q0 = Qs[qr * LD + li];
        const float g1 = hi_ok ? Gs[qr * LD + 32 + li] : 0.f;
        const float q1 = hi_ok ? Qs[qr * LD + 32 + li] : 0.f;
        dv0 = __builtin_amdgcn_mfma_f32_32x32x2f32(g0, sv[r], dv0, 0, 0, 0);
        dk0 = __builtin_amdgcn_mfma_f32_32x32x2f32(q0, dp[r], dk0, 0, 0, 0);
        dv1 = __builtin_amdgcn_mfma_f32_32x32x2f32(g1, sv[r], dv1, 0, 0, 0);
        dk1 = __builtin_amdgcn_mfma_f32_32x32x2f32(q1, dp[r], dk1, 0, 0, 0);
      }
    }
  };
  auto end_tile = [&]() {
    float* slot = slots + wave * (32 * LD);
    att_store_tile_T<HD>(slot, dk0, dk1, dKb + (int64_t)(kt * 32) * row_stride, row_stride, li, lh, lane, 1.f,
                         RC, RS, kt * 32);
    att_store_tile_T<HD>(slot, dv0, dv1, dVb + (int64_t)(kt * 32) * row_stride, row_stride, li, lh, lane, 1.f);
  };
  for (int c = 0; c < nchunk; ++c) {
    const int c0 = 4 * c, rows = min(ATT_CH, L - ATT_CH * c);
    if (c > 0) __syncthreads();
    att_stage_chunk<HD, 256>(Qs, Gs, Q + base, dOb, ATT_CH * c, rows, row_stride, o_row_stride, tid, RC, RS, true, false);
    for (int q = tid; q < rows; q += 256) {
      lse_s[q] = LSE[(int64_t)bh * L + ATT_CH * c + q];
      delta_s[q] = Delta[(int64_t)bh * L + ATT_CH * c + q];
    }
    __syncthreads();
    const int c_end = c0 + rows / 32;              // query tiles [c0, c_end) are resident
    const bool last = c == nchunk - 1;
    // causal: key tile t needs query tiles [t, ntile).  A tile is started in the chunk that holds its own rows.
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      const int t = which == 0 ? tA : tB;
      if (t < 0 || t >= c_end) continue;           // its queries start in a later chunk
      if (kt != t) begin_tile(t);
      run_chunk(c0, max(t, c0), c_end);
      if (last) end_tile();
    }
  }
}

// ======================================================================================
// Decode attention (llm/llama/model.py:105-121 in eval mode, one new token per sequence): the query
// (B, 1, H, hd) against the first T positions of the KV cache (max_batch, max_len, H, hd), no mask
// (L = 1).  One 256-thread workgroup per (batch, head): scores to LDS, block max / sum, then the
// probability-weighted sum of the value rows.  Tiny and latency-bound; it replaces two batched GEMMs,
// a softmax and the cache slicing of the generic path.
// ======================================================================================
__global__ __launch_bounds__(256) void attention_decode_kernel(const float* __restrict__ q, const float* __restrict__ kc,
                                                               const float* __restrict__ vc, float* __restrict__ o,
                                                               int H, int T, int hd, int64_t cache_batch_stride,
                                                               float inv_sqrt) {
  extern __shared__ __attribute__((aligned(16))) float sc[];      // [T] scores, then [21][hd] partial sums
  __shared__ float red[16];
  const int b = blockIdx.x / H, h = blockIdx.x % H, tid = threadIdx.x;
  const int D = H * hd, f4 = hd / 4;
  const float4* q4 = reinterpret_cast<const float4*>(q + ((int64_t)b * H + h) * hd);
  const float* kb = kc + (int64_t)b * cache_batch_stride + (int64_t)h * hd;
  const float* vb = vc + (int64_t)b * cache_batch_stride + (int64_t)h * hd;
  float m = -INFINITY;
  for (int t = tid; t < T; t += 256) {
    const float4* k4 = reinterpret_cast<const float4*>(kb + (int64_t)t * D);
    float s = 0.f;
    for (int c = 0; c < f4; ++c) {
      const float4 a = q4[c], k = k4[c];
      s += (a.x * k.x + a.y * k.y) + (a.z * k.z + a.w * k.w);
    }
    s *= inv_sqrt;
    sc[t] = s;
    m = fmaxf(m, s);
  }
  m = block_max(m, red);
  float l = 0.f;
  for (int t = tid; t < T; t += 256) {
    const float pr = expf(sc[t] - m);
    sc[t] = pr;
    l += pr;
  }
  l = block_sum(l, red);                 // (its barriers also publish the probabilities)
  // weighted sum of V rows: thread = (key group tg, float4 column c)
  const int groups = 256 / f4, c = tid % f4, tg = tid / f4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tg < groups) {
    for (int t = tg; t < T; t += groups) {
      const float pr = sc[t];
      const float4 v = *reinterpret_cast<const float4*>(vb + (int64_t)t * D + 4 * c);
      acc.x += pr * v.x; acc.y += pr * v.y; acc.z += pr * v.z; acc.w += pr * v.w;
    }
  }
  __syncthreads();                       // scores are dead: reuse the buffer for the partial sums
  float4* part = reinterpret_cast<float4*>(sc);
  if (tg < groups) part[tg * f4 + c] = acc;
  __syncthreads();
  if (tid < f4) {
    float4 r = part[tid];
    for (int g = 1; g < groups; ++g) { const float4 t = part[g * f4 + tid]; r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w; }
    const float inv = 1.f / l;
    r.x *= inv; r.y *= inv; r.z *= inv; r.w *= inv;
    reinterpret_cast<float4*>(o + ((int64_t)b * H + h) * hd)[tid] = r;
  }
}

// q, o: (B, H, head_dim) contiguous; k_cache / v_cache: (max_batch, max_len, H, head_dim) with
// `cache_batch_stride` floats between sequences; attends to positions [0, T).
extern "C" int pdn_attention_decode_f32(const float* q, const float* k_cache, const float* v_cache, float* o,
                                        int B, int H, int T, int head_dim, int64_t cache_batch_stride,
                                        void* stream) {
  if (B == 0 || H == 0) return PDN_OK;
  PDN_CHECK_ARG(q && k_cache && v_cache && o && T > 0, "pdn_attention_decode_f32: bad arguments");
  PDN_CHECK_ARG(head_dim % 4 == 0 && head_dim <= 256 && (cache_batch_stride % 4) == 0 &&
                    ((((uintptr_t)q | (uintptr_t)k_cache | (uintptr_t)v_cache | (uintptr_t)o) & 15) == 0),
                "pdn_attention_decode_f32: head_dim %% 4, 16-byte alignment required");
  const int f4 = head_dim / 4, groups = 256 / f4;
  const size_t shm = sizeof(float) * (size_t)(T > groups * head_dim ? T : groups * head_dim);
  PDN_CHECK_ARG(shm <= 64 * 1024, "pdn_attention_decode_f32: T=%d too long", T);
  hipLaunchKernelGGL(attention_decode_kernel, dim3(B * H), dim3(256), shm, (hipStream_t)stream, q, k_cache,
                     v_cache, o, H, T, head_dim, cache_batch_stride, 1.f / sqrtf((float)head_dim));
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}
