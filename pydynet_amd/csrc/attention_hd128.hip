// Resident attention for head dim 128 (gfx950, fp32 MFMA): forward, dQ, dK / dV.
//
// The shape of examples/pydynet/transformer.py:53-130 (dim 512, 4 heads; the third benchmark the reference's README
// publishes): its `matmul -> / sqrt(hd) -> + padding mask -> softmax -> matmul` (transformer.py:120-128, the same chain as
// llm/llama/model.py:112-121) ran on the general streaming kernels of csrc/attention_stream.hip (20-27 % causal-useful at
// this head dim), because the resident kernels of csrc/attention.hip plan their registers for two 32-row head-dim tiles
// (head dim <= 64: the 256-key score tiles alone are 128 registers).  Here the register plan is FOUR head-dim tiles and
// the scores of ONE 32-key tile at a time:
//   * same MFMA formulation as attention.hip -- S^T[key][q] = K Q^T with lane = query, so the softmax row reductions are
//     in-lane + one cross-half shuffle, and the accumulator registers of P^T feed O^T[d][q] = V^T P^T as they are (inside one
//     MFMA the two half-waves may contract over any two keys as long as A and B agree: register r pairs keys krow(r) and
//     krow(r) + 4);
//   * K and V (backward: also Q and dO) pass through LDS in CHUNKS OF 64 ROWS, [64][hd + 4] (row fragments = conflict-free
//     ds_read_b128, four k-steps each; columns = ds_read_b32 with consecutive lanes on consecutive floats), the next chunk
//     prefetched into registers while the current one is multiplied; an online softmax carries (m, l, O) from tile to tile;
//   * ANY length up to 1024: rows beyond L are staged as zeros and masked with -inf through the same per-key bias that
//     carries a (batch, key) padding mask (transformer.py:92-96); queries beyond L are computed and never stored.  Causal
//     masks skip the key tiles above the diagonal; query tiles are dealt zig-zag over the waves so that the two waves of a
//     SIMD carry 9 of the 36 tile pairs each.
//   * forward, causal: 8 waves per 256 queries, 64-row chunks (two waves per SIMD: one does its softmax while the other
//     multiplies); forward, no causal mask: 4-wave workgroups per 128 queries, 32-row chunks, two workgroups per CU (same-box
//     A/B at L = 256, 2048 heads: causal 519 vs 589 us, full 651 vs 616 us); backward kernels: 4 waves with the whole
//     register file of a SIMD each (dK / dV: 2 x 64 operand + 2 x 64 accumulator + 32 score registers).
//   * measured (tools/attn_hd128_probe.py, L = 256, 2048 heads, MI355X): forward 70.5 % of the fp32-MFMA peak without a mask
//     (streaming kernels: 53.7 %), 42 % causal-useful (27 %) -- under a causal mask the waves of a workgroup meet at every
//     chunk's barriers while the late chunks concern only its last query tiles; backward 46 % / 23 %.
// Backward recomputes P from the saved log-sum-exp (natural log of the row sums of exp(s / sqrt(hd) + bias)):
//   dQ kernel   a wave owns a query tile: S^T, dP^T = V dO^T, dS^T = P^T o (dP^T - delta), dQ^T += K^T dS^T; writes delta
//   dK/dV kernel a wave owns a key tile:  S = Q K^T (lane = key), dP = dO V^T, dV^T += dO^T P, dK^T += Q^T dS
#include "common.h"
#include <math.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define A8_HD 128
#define A8_LD (A8_HD + 4)          // LDS row stride: 33 sixteen-byte slots (odd: row fragments are conflict free)
#define A8_KC 64                   // rows per chunk
#define A8_NT8 (A8_HD / 8)         // float4 fragments per row and half-wave
#define A8_DT (A8_HD / 32)         // head-dim tiles
#define A8_MAX_L 1024
#define A8_LOG2E 1.4426950408889634f

namespace {

__device__ __forceinline__ int a8_krow(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }
__device__ __forceinline__ int a8_tile_of_wave8(int w) { return w < 4 ? w : 11 - w; }   // SIMD s hosts tiles s and 7 - s
__device__ __forceinline__ int a8_tile_of_wave4(int w) { return w; }

// Chunk staging: `rows` rows starting at row0 of a (L, 128) operand with `rs` floats between rows -> [A8_KC][A8_LD];
// rows >= L are zeros.  NT threads; piece p = (row, 16-byte unit): 32 units per row.
template <int NT, int KC = A8_KC>
struct A8Stage {
  static constexpr int NP = KC * (A8_HD / 4) / NT;
  float4 v[NP];
  __device__ __forceinline__ void issue(const float* __restrict__ g, int row0, int L, int64_t rs, int tid) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int p = tid + i * NT, row = p >> 5, u = p & 31;
      const int rc = min(row0 + row, L - 1);
      v[i] = *reinterpret_cast<const float4*>(g + (int64_t)rc * rs + 4 * u);
      if (row0 + row >= L) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __device__ __forceinline__ void commit(float* __restrict__ dst, int tid) const {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int p = tid + i * NT, row = p >> 5, u = p & 31;
      *reinterpret_cast<float4*>(dst + row * A8_LD + 4 * u) = v[i];
    }
  }
};

// the lane's row of a (L, 128) operand as MFMA fragments: f[t] = row[8 t + 4 lh .. + 3]; rows >= L read row L - 1
__device__ __forceinline__ void a8_load_row(f32x4 (&f)[A8_NT8], const float* __restrict__ g, int row, int L, int64_t rs, int lh) {
  const float* p = g + (int64_t)min(row, L - 1) * rs + 4 * lh;
#pragma unroll
  for (int t = 0; t < A8_NT8; ++t) f[t] = *reinterpret_cast<const f32x4*>(p + 8 * t);
}

// acc^T tiles (lane = row of the output, register r of tile dt = column dt * 32 + krow(r, lh)) -> the lane's own row, 16-byte pieces
__device__ __forceinline__ void a8_store_row(const f32x16 (&acc)[A8_DT], float* __restrict__ rowp, float scale, int lh) {
#pragma unroll
  for (int dt = 0; dt < A8_DT; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(rowp + dt * 32 + 8 * g + 4 * lh) =
          make_float4(acc[dt][4 * g] * scale, acc[dt][4 * g + 1] * scale, acc[dt][4 * g + 2] * scale, acc[dt][4 * g + 3] * scale);
}

// rows x fragments product: D^T[a-row][b-lane] += sum over the head dim; A rows from the LDS image (ds_read_b128), B = f.
// The fragment of step t + 1 is requested in FRONT of step t's four MFMAs (pinned with sched_barrier: left to itself hipcc
// reads, waits, multiplies -- a full LDS round trip per four MFMAs).
__device__ __forceinline__ void a8_rows_times_frag(f32x16& d, const float* __restrict__ img_row, const f32x4 (&f)[A8_NT8]) {
  f32x4 a[2];
  a[0] = *reinterpret_cast<const f32x4*>(img_row);
#pragma unroll
  for (int t = 0; t < A8_NT8; ++t) {
    if (t + 1 < A8_NT8) a[(t + 1) & 1] = *reinterpret_cast<const f32x4*>(img_row + 8 * (t + 1));
    __builtin_amdgcn_sched_barrier(0);
    d = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t & 1][0], f[t][0], d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t & 1][1], f[t][1], d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t & 1][2], f[t][2], d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t & 1][3], f[t][3], d, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// two of them at once (S and dP of the backward kernels): eight independent MFMAs per pair of reads
__device__ __forceinline__ void a8_rows_times_frag2(f32x16& d0, const float* __restrict__ row0, const f32x4 (&f0)[A8_NT8],
                                                    f32x16& d1, const float* __restrict__ row1, const f32x4 (&f1)[A8_NT8]) {
  f32x4 a[2], b[2];
  a[0] = *reinterpret_cast<const f32x4*>(row0);
  b[0] = *reinterpret_cast<const f32x4*>(row1);
#pragma unroll
  for (int t = 0; t < A8_NT8; ++t) {
    if (t + 1 < A8_NT8) {
      a[(t + 1) & 1] = *reinterpret_cast<const f32x4*>(row0 + 8 * (t + 1));
      b[(t + 1) & 1] = *reinterpret_cast<const f32x4*>(row1 + 8 * (t + 1));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t & 1][j], f0[t][j], d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[t & 1][j], f1[t][j], d1, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// columns x accumulator product: acc[dt]^T[d][lane] += sum over the tile's 32 rows of img[row][d] * s[row][lane]; A = column reads
// (lane = d index, ds_read_b32), B = the accumulator registers of s as they are (register r = rows krow(r, 0 / 1)); the four
// column values of step r + 1 are requested in front of step r's MFMAs
__device__ __forceinline__ void a8_cols_times_acc(f32x16 (&acc)[A8_DT], const float* __restrict__ img_tile, const f32x16& s, int li, int lh) {
  const float* base = img_tile + 4 * lh * A8_LD + li;
  float a[2][A8_DT];
#pragma unroll
  for (int dt = 0; dt < A8_DT; ++dt) a[0][dt] = base[32 * dt];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    if (r + 1 < 16) {
      const float* row = base + (((r + 1) & 3) + 8 * ((r + 1) >> 2)) * A8_LD;
#pragma unroll
      for (int dt = 0; dt < A8_DT; ++dt) a[(r + 1) & 1][dt] = row[32 * dt];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int dt = 0; dt < A8_DT; ++dt) acc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r & 1][dt], s[r], acc[dt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// two of them at once (dV and dK of the backward kernel): eight independent accumulators per step
__device__ __forceinline__ void a8_cols_times_acc2(f32x16 (&acc0)[A8_DT], const float* __restrict__ img0, const f32x16& s0,
                                                   f32x16 (&acc1)[A8_DT], const float* __restrict__ img1, const f32x16& s1, int li, int lh) {
  const float* b0 = img0 + 4 * lh * A8_LD + li;
  const float* b1 = img1 + 4 * lh * A8_LD + li;
  float a[2][A8_DT], b[2][A8_DT];
#pragma unroll
  for (int dt = 0; dt < A8_DT; ++dt) { a[0][dt] = b0[32 * dt]; b[0][dt] = b1[32 * dt]; }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    if (r + 1 < 16) {
      const int off = (((r + 1) & 3) + 8 * ((r + 1) >> 2)) * A8_LD;
#pragma unroll
      for (int dt = 0; dt < A8_DT; ++dt) { a[(r + 1) & 1][dt] = b0[off + 32 * dt]; b[(r + 1) & 1][dt] = b1[off + 32 * dt]; }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int dt = 0; dt < A8_DT; ++dt) {
      acc0[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r & 1][dt], s0[r], acc0[dt], 0, 0, 0);
      acc1[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[r & 1][dt], s1[r], acc1[dt], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// per-key bias of a chunk in the log2 domain: key_bias * log2(e), -inf for keys >= L
template <int KC = A8_KC>
__device__ __forceinline__ void a8_stage_bias(float* __restrict__ bs, const float* __restrict__ kb, int key0, int L, int tid) {
  if (tid < KC) {
    const int key = key0 + tid;
    bs[tid] = key < L ? (kb ? kb[key] * A8_LOG2E : 0.f) : -INFINITY;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// forward: grid (query groups of 128, B * H); 4 waves, 32-row chunks, TWO workgroups per CU (34 KB of LDS, <= 256 registers).
// (First form: one 8-wave workgroup per head and 256 queries, 64-row chunks.  Under a causal mask the waves of a workgroup
//  meet at every chunk's barriers while the late chunks concern only its last query tiles: 38 % causal-useful at L = 256
//  against 59 % without a mask.  Four-tile workgroups need 2 or 4 chunks' worth of barriers each and the dispatcher packs
//  short and long ones onto the CUs.)
// ------------------------------------------------------------------------------------------------------------------
template <int NW, int KC>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void att128_fwd_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                            const float* __restrict__ V, float* __restrict__ O,
                                                            float* __restrict__ LSE, int H, int L, int64_t rs, int64_t bs,
                                                            int64_t ors, int64_t obs, float sqrt_hd, int causal,
                                                            const float* __restrict__ KB, int64_t kb_bs) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Ks = lds;
  float* Vs = Ks + KC * A8_LD;
  float* Bs = Vs + KC * A8_LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const int64_t base = (int64_t)b * bs + (int64_t)h * A8_HD, obase = (int64_t)b * obs + (int64_t)h * A8_HD;
  const float* kb = KB ? KB + (int64_t)b * kb_bs : nullptr;
  const int qt = blockIdx.x * NW + (NW == 8 ? a8_tile_of_wave8(wave) : a8_tile_of_wave4(wave)), qpos = qt * 32 + li;
  const bool active = qt * 32 < L;
  const float c1 = A8_LOG2E / sqrt_hd;
  // keys this workgroup needs: all of them, or (causal) up to its last query
  const int q_hi = min(L, (int)(blockIdx.x + 1) * 32 * NW);
  const int nchunk = ((causal ? q_hi : L) + KC - 1) / KC;

  f32x4 qf[A8_NT8];
  a8_load_row(qf, Q + base, qpos, L, rs, lh);
  A8Stage<64 * NW, KC> sk, sv;
  sk.issue(K + base, 0, L, rs, tid);
  sv.issue(V + base, 0, L, rs, tid);
  f32x16 o[A8_DT];
#pragma unroll
  for (int dt = 0; dt < A8_DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m_run = -INFINITY, l_part = 0.f;

  for (int c = 0; c < nchunk; ++c) {
    __syncthreads();                                   // everybody is done with the previous chunk
    sk.commit(Ks, tid);
    sv.commit(Vs, tid);
    a8_stage_bias<KC>(Bs, kb, c * KC, L, tid);
    __syncthreads();
    if (c + 1 < nchunk) {
      sk.issue(K + base, (c + 1) * KC, L, rs, tid);
      sv.issue(V + base, (c + 1) * KC, L, rs, tid);
    }
    if (!active) continue;
#pragma unroll 1
    for (int kt2 = 0; kt2 < KC / 32; ++kt2) {
      const int kt = c * (KC / 32) + kt2;
      if (kt * 32 >= L || (causal && kt > qt)) break;
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
      a8_rows_times_frag(s, Ks + (kt2 * 32 + li) * A8_LD + 4 * lh, qf);       // S^T[key][q]: lane = query
      // t = s / sqrt(hd) * log2 e + bias2[key]; causal: keys above the query's position are -inf
      float mt = -INFINITY;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b4 = *reinterpret_cast<const float4*>(Bs + kt2 * 32 + 8 * g + 4 * lh);
        const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          float t = fmaf(s[r], c1, bb[e]);
          if (causal && kt == qt && a8_krow(r, lh) > li) t = -INFINITY;
          s[r] = t;
          mt = fmaxf(mt, t);
        }
      }
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      const float m_new = fmaxf(m_run, mt);
      const float m_use = m_new == -INFINITY ? 0.f : m_new;                   // (a row with no visible key yet: p = 0)
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = __builtin_amdgcn_exp2f(s[r] - m_use);
        ps += s[r];
      }
      l_part = l_part * alpha + ps;
      m_run = m_new;
      if (__any(alpha != 1.f)) {                       // (after the first tiles the running maxima rarely move)
#pragma unroll
        for (int dt = 0; dt < A8_DT; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
      }
      a8_cols_times_acc(o, Vs + kt2 * 32 * A8_LD, s, li, lh);                  // O^T[d][q] += V^T P^T
    }
  }
  if (!active || qpos >= L) return;
  const float l = l_part + __shfl_xor(l_part, 32, 64);
  // (shuffles are wave-wide: every lane of an active wave reaches this point together; rows >= L leave now)
  a8_store_row(o, O + obase + (int64_t)qpos * ors, 1.f / l, lh);
  if (lh == 0) LSE[(int64_t)bh * L + qpos] = (m_run + log2f(l)) * 0.6931471805599453f;
}

// ------------------------------------------------------------------------------------------------------------------
// dQ: grid (query groups of 128, B * H); 4 waves; LDS: K chunk | V chunk | bias.  Writes Delta = rowsum(dO o O).
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 1) void att128_bwd_dq_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                               const float* __restrict__ V, const float* __restrict__ O,
                                                               const float* __restrict__ dO, const float* __restrict__ LSE,
                                                               float* __restrict__ dQ, float* __restrict__ Delta, int H, int L,
                                                               int64_t rs, int64_t bs, int64_t ors, int64_t obs, float sqrt_hd,
                                                               int causal, const float* __restrict__ KB, int64_t kb_bs) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Ks = lds;
  float* Vs = Ks + A8_KC * A8_LD;
  float* Bs = Vs + A8_KC * A8_LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const int64_t base = (int64_t)b * bs + (int64_t)h * A8_HD, obase = (int64_t)b * obs + (int64_t)h * A8_HD;
  const float* kb = KB ? KB + (int64_t)b * kb_bs : nullptr;
  const int qt = blockIdx.x * 4 + a8_tile_of_wave4(wave), qpos = qt * 32 + li;
  const bool active = qt * 32 < L;
  const float inv_sqrt = 1.f / sqrt_hd, c1 = A8_LOG2E * inv_sqrt;
  const int q_hi = min(L, (int)(blockIdx.x + 1) * 128);
  const int nchunk = ((causal ? q_hi : L) + A8_KC - 1) / A8_KC;

  f32x4 qf[A8_NT8], gf[A8_NT8];
  a8_load_row(qf, Q + base, qpos, L, rs, lh);
  a8_load_row(gf, dO + obase, qpos, L, ors, lh);
  float delta_q;
  {
    f32x4 of_[A8_NT8];
    a8_load_row(of_, O + obase, qpos, L, ors, lh);
    float dpart = 0.f;
#pragma unroll
    for (int t = 0; t < A8_NT8; ++t)
      dpart += (of_[t][0] * gf[t][0] + of_[t][1] * gf[t][1]) + (of_[t][2] * gf[t][2] + of_[t][3] * gf[t][3]);
    delta_q = dpart + __shfl_xor(dpart, 32, 64);
  }
  const float c2q = -LSE[(int64_t)bh * L + min(qpos, L - 1)] * A8_LOG2E;
  A8Stage<256> sk, sv;
  sk.issue(K + base, 0, L, rs, tid);
  sv.issue(V + base, 0, L, rs, tid);
  f32x16 dq[A8_DT];
#pragma unroll
  for (int dt = 0; dt < A8_DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;

  for (int c = 0; c < nchunk; ++c) {
    __syncthreads();
    sk.commit(Ks, tid);
    sv.commit(Vs, tid);
    a8_stage_bias(Bs, kb, c * A8_KC, L, tid);
    __syncthreads();
    if (c + 1 < nchunk) {
      sk.issue(K + base, (c + 1) * A8_KC, L, rs, tid);
      sv.issue(V + base, (c + 1) * A8_KC, L, rs, tid);
    }
    if (!active) continue;
#pragma unroll 1
    for (int kt2 = 0; kt2 < A8_KC / 32; ++kt2) {
      const int kt = c * (A8_KC / 32) + kt2;
      if (kt * 32 >= L || (causal && kt > qt)) break;
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
      a8_rows_times_frag2(s, Ks + (kt2 * 32 + li) * A8_LD + 4 * lh, qf,        // S^T[key][q]
                          dp, Vs + (kt2 * 32 + li) * A8_LD + 4 * lh, gf);      // dP^T[key][q] = V dO^T
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b4 = *reinterpret_cast<const float4*>(Bs + kt2 * 32 + 8 * g + 4 * lh);
        const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          float t = fmaf(s[r], c1, bb[e]) + c2q;
          if (causal && kt == qt && a8_krow(r, lh) > li) t = -INFINITY;
          const float p = __builtin_amdgcn_exp2f(t);
          s[r] = p * (dp[r] - delta_q);              // dS^T (its 1 / sqrt(hd) is applied once, when dQ is stored)
        }
      }
      a8_cols_times_acc(dq, Ks + kt2 * 32 * A8_LD, s, li, lh);                 // dQ^T[d][q] += K^T dS^T
    }
  }
  if (!active || qpos >= L) return;
  if (lh == 0) Delta[(int64_t)bh * L + qpos] = delta_q;
  a8_store_row(dq, dQ + base + (int64_t)qpos * rs, inv_sqrt, lh);
}

// ------------------------------------------------------------------------------------------------------------------
// dK / dV: grid (key groups of 128, B * H); 4 waves; LDS: Q chunk | dO chunk | lse2, delta of the chunk's queries
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 1) void att128_bwd_dkv_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                                const float* __restrict__ V, const float* __restrict__ dO,
                                                                const float* __restrict__ LSE, const float* __restrict__ Delta,
                                                                float* __restrict__ dK, float* __restrict__ dV, int H, int L,
                                                                int64_t rs, int64_t bs, int64_t ors, int64_t obs, float sqrt_hd,
                                                                int causal, const float* __restrict__ KB, int64_t kb_bs) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Qs = lds;
  float* Gs = Qs + A8_KC * A8_LD;
  float* St = Gs + A8_KC * A8_LD;                    // [0, 64): -lse * log2 e (+inf for queries >= L: p = 0), [64, 128): delta
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const int64_t base = (int64_t)b * bs + (int64_t)h * A8_HD, obase = (int64_t)b * obs + (int64_t)h * A8_HD;
  const int kt = blockIdx.x * 4 + a8_tile_of_wave4(wave), kpos = kt * 32 + li;
  const bool active = kt * 32 < L;
  const float inv_sqrt = 1.f / sqrt_hd, c1 = A8_LOG2E * inv_sqrt;
  // queries this workgroup needs: all of them, or (causal) from its first key on
  const int c_lo = causal ? ((int)blockIdx.x * 128) / A8_KC : 0;
  const int nchunk = (L + A8_KC - 1) / A8_KC;
  const float bias2 = kpos < L ? (KB ? KB[(int64_t)b * kb_bs + kpos] * A8_LOG2E : 0.f) : -INFINITY;

  f32x4 kf[A8_NT8], vf[A8_NT8];
  a8_load_row(kf, K + base, kpos, L, rs, lh);
  a8_load_row(vf, V + base, kpos, L, rs, lh);
  A8Stage<256> sq, sg;
  sq.issue(Q + base, c_lo * A8_KC, L, rs, tid);
  sg.issue(dO + obase, c_lo * A8_KC, L, ors, tid);
  float st_next = 0.f;
  auto issue_stats = [&](int c) {
    if (tid < 2 * A8_KC) {
      const int qq = c * A8_KC + (tid & (A8_KC - 1));
      if (tid < A8_KC) st_next = qq < L ? -LSE[(int64_t)bh * L + qq] * A8_LOG2E : -INFINITY;
      else st_next = qq < L ? Delta[(int64_t)bh * L + qq] : 0.f;
    }
  };
  issue_stats(c_lo);
  f32x16 dk[A8_DT], dv[A8_DT];
#pragma unroll
  for (int dt = 0; dt < A8_DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }

  for (int c = c_lo; c < nchunk; ++c) {
    __syncthreads();
    sq.commit(Qs, tid);
    sg.commit(Gs, tid);
    if (tid < 2 * A8_KC) St[tid] = st_next;
    __syncthreads();
    if (c + 1 < nchunk) {
      sq.issue(Q + base, (c + 1) * A8_KC, L, rs, tid);
      sg.issue(dO + obase, (c + 1) * A8_KC, L, ors, tid);
      issue_stats(c + 1);
    }
    if (!active) continue;
#pragma unroll 1
    for (int qt2 = 0; qt2 < A8_KC / 32; ++qt2) {
      const int qt = c * (A8_KC / 32) + qt2;
      if (qt * 32 >= L) break;
      if (causal && qt < kt) continue;
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
      a8_rows_times_frag2(s, Qs + (qt2 * 32 + li) * A8_LD + 4 * lh, kf,        // S[q][key]: lane = key
                          dp, Gs + (qt2 * 32 + li) * A8_LD + 4 * lh, vf);      // dP[q][key] = dO V^T
      f32x16 pr;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 l4 = *reinterpret_cast<const float4*>(St + qt2 * 32 + 8 * g + 4 * lh);
        const float4 d4 = *reinterpret_cast<const float4*>(St + A8_KC + qt2 * 32 + 8 * g + 4 * lh);
        const float ll[4] = {l4.x, l4.y, l4.z, l4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          float t = fmaf(s[r], c1, bias2) + ll[e];
          if (causal && qt == kt && li > a8_krow(r, lh)) t = -INFINITY;     // key above the query's position
          const float p = __builtin_amdgcn_exp2f(t);
          pr[r] = p;
          s[r] = p * (dp[r] - dd[e]);                                          // dS[q][key]
        }
      }
      a8_cols_times_acc2(dv, Gs + qt2 * 32 * A8_LD, pr,                        // dV^T[d][key] += dO^T P
                         dk, Qs + qt2 * 32 * A8_LD, s, li, lh);                // dK^T[d][key] += Q^T dS
    }
  }
  if (!active || kpos >= L) return;
  a8_store_row(dk, dK + base + (int64_t)kpos * rs, inv_sqrt, lh);
  a8_store_row(dv, dV + base + (int64_t)kpos * rs, 1.f, lh);
}

constexpr int kA8LdsFwd = (2 * 32 * A8_LD + 32) * 4;
constexpr int kA8LdsDq = (2 * A8_KC * A8_LD + A8_KC) * 4;
constexpr int kA8LdsDkv = (2 * A8_KC * A8_LD + 2 * A8_KC) * 4;

template <class Kern>
int a8_lds_attr(Kern kern, int bytes) {
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) { pdn_set_error("attention (head dim 128): LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
  return PDN_OK;
}

}  // namespace

bool pdn_attention_hd128_ok(int L, int head_dim) {
  static const bool off = getenv("PDN_NO_ATT_HD128") != nullptr;      // A/B switch: the streaming kernels instead
  return !off && head_dim == A8_HD && L >= 1 && L <= A8_MAX_L;
}

int pdn_attention_hd128_fwd(const float* q, const float* k, const float* v, float* o, float* lse, int B, int H, int L,
                            int64_t rs, int64_t bs, int64_t ors, int64_t obs, int causal, const float* key_bias,
                            int64_t kb_bs, void* stream) {
  // causal: one 8-wave workgroup per 256 queries and 64-row chunks (zig-zag tiles: the two waves of a SIMD carry 9 of the
  // 36 tile pairs each); otherwise 4-wave workgroups, 32-row chunks, two per CU (measured at L = 256: 8 waves 530 us
  // causal / 650 us full, 4 waves 590 / 616)
  static const int force = getenv("PDN_ATT_HD128_NW") ? atoi(getenv("PDN_ATT_HD128_NW")) : 0;
  const bool wide = force ? force == 8 : (causal && L > 128);
  if (wide) {
    int rc = a8_lds_attr(att128_fwd_kernel<8, 64>, kA8LdsDq);
    if (rc) return rc;
    hipLaunchKernelGGL((att128_fwd_kernel<8, 64>), dim3((L + 255) / 256, B * H), dim3(512), kA8LdsDq, (hipStream_t)stream, q, k, v,
                       o, lse, H, L, rs, bs, ors, obs, sqrtf((float)A8_HD), causal, key_bias, kb_bs);
  } else {
    int rc = a8_lds_attr(att128_fwd_kernel<4, 32>, kA8LdsFwd);
    if (rc) return rc;
    hipLaunchKernelGGL((att128_fwd_kernel<4, 32>), dim3((L + 127) / 128, B * H), dim3(256), kA8LdsFwd, (hipStream_t)stream, q, k, v,
                       o, lse, H, L, rs, bs, ors, obs, sqrtf((float)A8_HD), causal, key_bias, kb_bs);
  }
  PDN_LAUNCH_CHECK();
  pdn_count(PDN_CNT_ATT_RES_FWD);
  return PDN_OK;
}

int pdn_attention_hd128_bwd(const float* q, const float* k, const float* v, const float* o, const float* d_o, const float* lse,
                            float* dq, float* dk, float* dv, int B, int H, int L, int64_t rs, int64_t bs, int64_t ors,
                            int64_t obs, int causal, const float* key_bias, int64_t kb_bs, float* delta, void* stream) {
  int rc = a8_lds_attr(att128_bwd_dq_kernel, kA8LdsDq);
  if (rc) return rc;
  rc = a8_lds_attr(att128_bwd_dkv_kernel, kA8LdsDkv);
  if (rc) return rc;
  const dim3 grid((L + 127) / 128, B * H);
  const float sq = sqrtf((float)A8_HD);
  hipLaunchKernelGGL(att128_bwd_dq_kernel, grid, dim3(256), kA8LdsDq, (hipStream_t)stream, q, k, v, o, d_o, lse, dq, delta, H,
                     L, rs, bs, ors, obs, sq, causal, key_bias, kb_bs);
  PDN_LAUNCH_CHECK();
  hipLaunchKernelGGL(att128_bwd_dkv_kernel, grid, dim3(256), kA8LdsDkv, (hipStream_t)stream, q, k, v, d_o, lse, delta, dk, dv, H,
                     L, rs, bs, ors, obs, sq, causal, key_bias, kb_bs);
  PDN_LAUNCH_CHECK();
  pdn_count(PDN_CNT_ATT_RES_BWD);
  return PDN_OK;
}
