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
// One workgroup (8 wave64, two per SIMD) per (batch, head, group of 8 query tiles = 256 queries).  K and V of the
// head pass through LDS in CHUNKS of 256 keys ([256][hd+4] / [256][64], all global loads of a thread issued before
// the first LDS write); L <= 256 is one chunk -- the benchmark shape -- and longer sequences (L <= 1024,
// llm/llama/finetune.py:44 max_seq_len) carry (m, l, O) from chunk to chunk with one online rescale per chunk
// (a causal query group q only visits chunks 0 .. q).  A wave owns one 32-query tile; SIMD s hosts waves s and
// s+4, which take tiles s and 7-s of the group, so the causal work of the diagonal chunk (s+1 and 8-s key tiles)
// is balanced over the four SIMDs and each SIMD always has a second wave to issue MFMAs while the other one does
// its softmax.  Head dims 48 and 64.
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

// -DATT_TRACE (tools/attn_trace.sh, never the shipped build): every wave of the forward kernel leaves 100 MHz
// timestamps of its phases, its tile and the CU it ran on.
#ifdef ATT_TRACE
#define ATT_TRACE_SLOTS (1 << 17)
static __device__ unsigned long long att_trace_buf[ATT_TRACE_SLOTS][10];
static __device__ unsigned att_trace_cnt;
#define ATT_T_BEGIN() unsigned long long tr_[8] = {}; tr_[0] = __builtin_amdgcn_s_memrealtime()
#define ATT_T(i) tr_[i] = __builtin_amdgcn_s_memrealtime()
#define ATT_T_END(tile) do { if ((threadIdx.x & 63) == 0) { const unsigned sl = atomicAdd(&att_trace_cnt, 1u);             \
  if (sl < ATT_TRACE_SLOTS) { for (int i_ = 0; i_ < 8; ++i_) att_trace_buf[sl][i_] = tr_[i_];                             \
    att_trace_buf[sl][8] = ((unsigned long long)blockIdx.x << 8) | (unsigned)(tile);                                      \
    att_trace_buf[sl][9] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 32) |                        \
                           (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20); } } } while (0)
extern "C" int pdn_att_trace_dump(unsigned long long* out, unsigned* n) {
  hipMemcpyFromSymbol(out, HIP_SYMBOL(att_trace_buf), sizeof(unsigned long long) * ATT_TRACE_SLOTS * 10);
  hipMemcpyFromSymbol(n, HIP_SYMBOL(att_trace_cnt), sizeof(unsigned));
  unsigned z = 0; hipMemcpyToSymbol(HIP_SYMBOL(att_trace_cnt), &z, sizeof(unsigned));
  return 0;
}
#else
#define ATT_T_BEGIN()
#define ATT_T(i)
#define ATT_T_END(tile)
#endif

#define ATT_MAX_TILES 8      // key / query tiles per LDS chunk (256 rows)
#define ATT_CHUNK (32 * ATT_MAX_TILES)
#define ATT_MAX_L 1024
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

// (cos, sin) of the pairs a lane's fragments / output row pieces cover, at ITS row: read once, at the top of the
// kernel -- behind the staging -- instead of right after the staging barrier and right before the final store, where
// nothing hides the latency (round 3: RoPE cost the backward 83 us of 572, most of it these two exposed reads).
template <int NT8>
struct AttRowRope {
  float2 c[NT8], s[NT8];
  __device__ __forceinline__ void load(const float* __restrict__ cs, const float* __restrict__ sn, int pos, int lh, int half) {
#pragma unroll
    for (int t = 0; t < NT8; ++t) {
      c[t] = *reinterpret_cast<const float2*>(cs + pos * half + 4 * t + 2 * lh);
      s[t] = *reinterpret_cast<const float2*>(sn + pos * half + 4 * t + 2 * lh);
    }
  }
  __device__ __forceinline__ float4 rot(const float4& v, int t, float sign) const {
    const float sx = s[t].x * sign, sy = s[t].y * sign;
    return make_float4(v.x * c[t].x - v.y * sx, v.x * sx + v.y * c[t].x, v.z * c[t].y - v.w * sy, v.z * sy + v.w * c[t].y);
  }
};

// Forward staging of `nrows` rows (positions pos0 ..): K as [nrows][HD+4]; V as [nrows][64].  For HD < 64 column HD
// of V is 1 and columns HD+1 .. 63 are 0: the second 32-row tile of O^T = V^T P^T then needs no per-lane select for
// the rows beyond HD, and its row HD accumulates the softmax denominator (the row sums of P) for free -- the padded
// tile is multiplied anyway.  `gk` / `gv` point at row pos0.
#define ATT_LDV 64
// the same with the pairs' (cos, sin) already in registers
__device__ __forceinline__ float4 att_rot_pre(float4 v, float2 c, float2 s) {
  float4 o;
  o.x = v.x * c.x - v.y * s.x; o.y = v.x * s.x + v.y * c.x;
  o.z = v.z * c.y - v.w * s.y; o.w = v.z * s.y + v.w * c.y;
  return o;
}

// Every load of the staging -- K, V and the (cos, sin) pairs -- is issued in ONE batch, unconditionally (threads past
// the end re-read unit 0), with 32-bit offsets; rotation and LDS stores follow.  Written as one loop that loads, rotates
// and keeps the result (round 2), every iteration sat in its own branch region and used its load at once: six
// serialised memory round trips per workgroup and kernel, ~12 us of the 30 us a workgroup lives (six workgroups pass
// through a CU one after the other, nothing overlaps them: 98 KB of LDS each).
template <int HD, int NT>
__device__ __forceinline__ void att_stage_kv(float* __restrict__ ks, float* __restrict__ vs,
                                             const float* __restrict__ gk, const float* __restrict__ gv,
                                             int nrows, int pos0, int64_t row_stride, int tid,
                                             const float* __restrict__ cs, const float* __restrict__ sn) {
  constexpr int LD = ATT_LD(HD), F4 = HD / 4;
  constexpr int NP = (ATT_CHUNK * F4 + NT - 1) / NT;
  const int nu = nrows * F4, rs = (int)row_stride;
  float4 r0[NP], r1[NP];
  float2 tc[NP], ts[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int u = tid + NT * j, uc = u < nu ? u : 0;
    const int row = uc / F4, c4 = uc - row * F4;
    const unsigned off = (unsigned)(row * rs + 4 * c4);
    // component-wise: a whole-float4 store into the array defeats SROA (scratch) on hipcc 7.2
    const float4 a = *reinterpret_cast<const float4*>(gk + off);
    const float4 c = *reinterpret_cast<const float4*>(gv + off);
    r0[j].x = a.x; r0[j].y = a.y; r0[j].z = a.z; r0[j].w = a.w;
    r1[j].x = c.x; r1[j].y = c.y; r1[j].z = c.z; r1[j].w = c.w;
    if (cs) {                                                     // (uniform)
      const unsigned to = (unsigned)((pos0 + row) * (HD / 2) + 2 * c4);
      tc[j] = *reinterpret_cast<const float2*>(cs + to);
      ts[j] = *reinterpret_cast<const float2*>(sn + to);
    }
  }
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int u = tid + NT * j;
    if (u < nu) {
      const int row = u / F4, c4 = u - row * F4;
      *reinterpret_cast<float4*>(ks + row * LD + 4 * c4) = cs ? att_rot_pre(r0[j], tc[j], ts[j]) : r0[j];
      *reinterpret_cast<float4*>(vs + row * ATT_LDV + 4 * c4) = r1[j];
    }
  }
  if constexpr (HD < ATT_LDV) {
    constexpr int PF4 = (ATT_LDV - HD) / 4;                 // pad units per row
    for (int u = tid; u < nrows * PF4; u += NT) {
      const int row = u / PF4, pc = u % PF4;
      *reinterpret_cast<float4*>(vs + row * ATT_LDV + HD + 4 * pc) = make_float4(pc == 0 ? 1.f : 0.f, 0.f, 0.f, 0.f);
    }
  }
}

// Backward staging: two [nrows][HD] matrices as [nrows][LD0 / LD1] images; a PADded image has row stride ATT_LDP = 68
// and ZERO columns HD .. 63, so that the second 32-row MFMA tile over the head dim reads its operand rows HD .. 63
// unconditionally (no per-lane select per MFMA).  272-byte rows keep both access patterns conflict free (ds_read_b128
// fragments along the row for 16 consecutive rows, ds_read_b32 across 32 consecutive columns).
#define ATT_LDP 68
template <int HD, int NT, bool PAD0, bool PAD1>
__device__ __forceinline__ void att_stage_two_pad(float* __restrict__ s0, float* __restrict__ s1,
                                                  const float* __restrict__ g0, const float* __restrict__ g1,
                                                  int nrows, int pos0, int64_t row_stride, int64_t row_stride1, int tid,
                                                  const float* __restrict__ cs, const float* __restrict__ sn,
                                                  bool rot0, bool rot1) {
  constexpr int LD0 = PAD0 ? ATT_LDP : ATT_LD(HD), LD1 = PAD1 ? ATT_LDP : ATT_LD(HD), F4 = HD / 4;
  constexpr int NP = (ATT_CHUNK * F4 + NT - 1) / NT;
  const int nu = nrows * F4, rs0 = (int)row_stride, rs1 = (int)row_stride1;
  const bool rot = cs && (rot0 || rot1);                          // (uniform)
  float4 r0[NP], r1[NP];
  float2 tc[NP], ts[NP];
  // (all loads in one batch, unconditional, then rotate + store: see att_stage_kv)
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int u = tid + NT * j, uc = u < nu ? u : 0;
    const int row = uc / F4, c4 = uc - row * F4;
    const float4 a = *reinterpret_cast<const float4*>(g0 + (unsigned)(row * rs0 + 4 * c4));
    const float4 c = *reinterpret_cast<const float4*>(g1 + (unsigned)(row * rs1 + 4 * c4));
    r0[j].x = a.x; r0[j].y = a.y; r0[j].z = a.z; r0[j].w = a.w;
    r1[j].x = c.x; r1[j].y = c.y; r1[j].z = c.z; r1[j].w = c.w;
    if (rot) {
      const unsigned to = (unsigned)((pos0 + row) * (HD / 2) + 2 * c4);
      tc[j] = *reinterpret_cast<const float2*>(cs + to);
      ts[j] = *reinterpret_cast<const float2*>(sn + to);
    }
  }
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int u = tid + NT * j;
    if (u < nu) {
      const int row = u / F4, c4 = u - row * F4;
      *reinterpret_cast<float4*>(s0 + row * LD0 + 4 * c4) = (cs && rot0) ? att_rot_pre(r0[j], tc[j], ts[j]) : r0[j];
      *reinterpret_cast<float4*>(s1 + row * LD1 + 4 * c4) = (cs && rot1) ? att_rot_pre(r1[j], tc[j], ts[j]) : r1[j];
    }
  }
  if constexpr (HD < 64) {
    constexpr int PF4 = (64 - HD) / 4;
    for (int u = tid; u < nrows * PF4; u += NT) {
      const int row = u / PF4, pc = u % PF4;
      if (PAD0) *reinterpret_cast<float4*>(s0 + row * LD0 + HD + 4 * pc) = make_float4(0.f, 0.f, 0.f, 0.f);
      if (PAD1) *reinterpret_cast<float4*>(s1 + row * LD1 + HD + 4 * pc) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
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

// the same with the row's (cos, sin) already in registers (register group g of t0 = pairs of fragment t = g, of t1 = 4 + g)
template <int HD>
__device__ __forceinline__ void att_store_rows_pre(const f32x16& t0, const f32x16& t1, float* __restrict__ rowp, float scale,
                                                   const AttRowRope<HD / 8>& rr) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 v = make_float4(t0[4 * g] * scale, t0[4 * g + 1] * scale, t0[4 * g + 2] * scale, t0[4 * g + 3] * scale);
    *reinterpret_cast<float4*>(rowp + 8 * g) = rr.rot(v, g, -1.f);
  }
#pragma unroll
  for (int g = 0; g < (HD - 32) / 8; ++g) {
    const float4 v = make_float4(t1[4 * g] * scale, t1[4 * g + 1] * scale, t1[4 * g + 2] * scale, t1[4 * g + 3] * scale);
    *reinterpret_cast<float4*>(rowp + 32 + 8 * g) = rr.rot(v, 4 + g, -1.f);
  }
}

// Workgroup -> (batch * head, row group): the groups of one head are consecutive blocks, heaviest first (a causal
// group g visits g + 1 chunks), so the long ones do not start last.
__device__ __forceinline__ void att_block(int G, int& bh, int& grp) {
  bh = blockIdx.x / G;
  grp = G - 1 - (int)(blockIdx.x % G);
}

// ABLATE (tools/micro/attn_ablate.hip only; 0 in the library): 1 = no K/V staging, 2 = no S^T MFMAs,
// 4 = no softmax arithmetic, 8 = no PV MFMAs, 16 = no output store -- timing experiments, wrong results.
// Out-of-line staging for the chunk loops: inlined into a loop body that also carries the accumulators across
// iterations, the staging registers (48 per thread) made the allocator spill 60-100 registers of the compute phase
// (hipcc 7.2); as a call they are dead outside it.  The single-chunk instantiations stage before the loop, inline.
template <int HD>
__device__ __attribute__((noinline)) void att_stage_kv_call(float* ks, float* vs, const float* gk, const float* gv, int nrows,
                                                            int pos0, int64_t row_stride, int tid, const float* cs,
                                                            const float* sn) {
  att_stage_kv<HD, 512>(ks, vs, gk, gv, nrows, pos0, row_stride, tid, cs, sn);
}
template <int HD, bool PAD1>
__device__ __attribute__((noinline)) void att_stage_pad_call(float* s0, float* s1, const float* g0, const float* g1, int nrows,
                                                             int pos0, int64_t rs0, int64_t rs1, int tid, const float* cs,
                                                             const float* sn, bool rot0) {
  att_stage_two_pad<HD, 512, true, PAD1>(s0, s1, g0, g1, nrows, pos0, rs0, rs1, tid, cs, sn, rot0, false);
}

// MULTI = false: L <= 256, one chunk (the chunk loop and the rescale fold away).
template <int HD, int ABLATE = 0, bool MULTI = false>
__global__ __launch_bounds__(512, 1) void attention_fwd_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
    float* __restrict__ O, float* __restrict__ LSE, int H, int L, int64_t row_stride,
    int64_t batch_stride, int64_t o_row_stride, int64_t o_batch_stride, float sqrt_hd, int causal,
    const float* __restrict__ RC, const float* __restrict__ RS, const float* __restrict__ KB, int64_t kb_bs) {
  static_assert(HD % 8 == 0 && HD > 32 && HD <= 64, "head dim: two 32-row MFMA tiles");
  constexpr int LD = ATT_LD(HD);
  constexpr int NT8 = HD / 8;                    // k-groups of 8 along the head dim
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int crows = min(L, ATT_CHUNK);
  float* Ks = lds;                               // [crows][LD]
  float* Vs = lds + (size_t)crows * LD;          // [crows][64]: V | 1 | 0 ...  (HD = 64: V)

  const int ntile = L / 32, G = MULTI ? (ntile + ATT_MAX_TILES - 1) / ATT_MAX_TILES : 1;
  int bh, qg;
  if (MULTI) att_block(G, bh, qg); else { bh = blockIdx.x; qg = 0; }
  const int b = bh / H, h = bh % H;
  const int tid = threadIdx.x, lane = tid & 63;
  ATT_T_BEGIN();
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int64_t base = (int64_t)b * batch_stride + (int64_t)h * HD;
  const float* Qb = Q + base;
  float* Ob = O + (int64_t)b * o_batch_stride + (int64_t)h * HD;

  const int qtl = att_tile_of_wave(wave);        // tile of this wave inside the group
  const int qt = qg * ATT_MAX_TILES + qtl;
  const bool active = qt < ntile;                // (idle waves still meet the workgroup barriers of the chunk loop)
  const float inv_sqrt = 1.f / sqrt_hd;
  const float c1 = inv_sqrt * 1.4426950408889634f;
  const int qpos = qt * 32 + li;
  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m = -INFINITY, lsum = 0.f;               // running row maximum (unscaled scores); lsum: HD = 64 only
  const int c_last = MULTI ? (causal ? qg : G - 1) : 0;
  AttRowRope<NT8> rr;
  // (operand preloads are UNCONDITIONAL -- an idle wave reads tile 0: a load under `if (active)` makes the compiler
  //  copy the loaded registers right behind it (the phi with the untaken path), i.e. wait for the load at once
  //  instead of after the staging it was meant to hide behind)
  const bool pre = !MULTI && RC != nullptr;
  const int qpos_l = (active ? qt : 0) * 32 + li;
  // (without RoPE the same loads read the first floats of Q: unconditional, never used)
  if constexpr (!MULTI) rr.load(RC ? RC : Q, RC ? RS : Q, RC ? qpos_l : 0, lh, HD / 2);
  float4 qraw[MULTI ? 1 : NT8];                  // single chunk: the Q rows go out before the staging (see dQ kernel)
  if (!MULTI) {
    const float* qrow = Qb + (int64_t)qpos_l * row_stride + 4 * lh;
#pragma unroll
    for (int t = 0; t < NT8; ++t) qraw[t] = *reinterpret_cast<const float4*>(qrow + 8 * t);
  }
  if (!MULTI) {
    if (!(ABLATE & 1)) att_stage_kv<HD, 512>(Ks, Vs, K + base, V + base, L, 0, row_stride, tid, RC, RS);
    ATT_T(1);
    __syncthreads();
    ATT_T(2);
    if (!active) return;                         // no workgroup barrier below this point
  }
  for (int c = 0; c <= c_last; ++c) {
    const int row0 = c * ATT_CHUNK, nrows = min(ATT_CHUNK, L - row0);
    if (MULTI) {
      if (c) __syncthreads();                    // every wave is done with the previous chunk's images
      if (!(ABLATE & 1))
        att_stage_kv_call<HD>(Ks, Vs, K + base + (int64_t)row0 * row_stride, V + base + (int64_t)row0 * row_stride, nrows,
                              row0, row_stride, tid, RC, RS);
      __syncthreads();
      if (!active) continue;
    }
    const bool diag = causal && c == qg;         // the chunk that holds this group's own positions
    const int nk = diag ? qtl + 1 : nrows / 32;  // key tiles of this chunk that can be unmasked
    // Q fragments: lane (li, lh) holds Q[q = qt*32+li][8t + 4lh .. +3]  (re-read per chunk -- six cache-resident
    // loads -- rather than kept live across the staging of the next chunk)
    float4 qf[NT8];
    {
      const float* qrow = Qb + (int64_t)qpos * row_stride + 4 * lh;
#pragma unroll
      for (int t = 0; t < NT8; ++t) {
        if constexpr (MULTI) qf[t] = *reinterpret_cast<const float4*>(qrow + 8 * t); else qf[t] = qraw[t];
        if (pre) qf[t] = rr.rot(qf[t], t, 1.f);
        else if (RC) qf[t] = att_rot(qf[t], RC, RS, qpos, 4 * t + 2 * lh, HD / 2, 1.f);
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
        // additive KEY bias (padding masks, padded keys of a ragged length): one more rank-1 step of the same product,
        // [k | b sqrt(hd)] . [q | 1] -- the lower half-wave carries the extra contraction index, the upper one zeros
        if (KB) {
          const float kb = lh == 0 ? KB[(int64_t)b * kb_bs + row0 + kt * 32 + li] * sqrt_hd : 0.f;
          s[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kb, lh == 0 ? 1.f : 0.f, s[kt], 0, 0, 0);
        }
      }
    }
    ATT_T(3);
    // ---- mask, softmax over keys (per lane = per query) ---------------------------------------
    // The 1/sqrt(hd) scale rides inside the exponential: p = exp2(s * c1 - max(s) * c1), c1 = log2(e) / sqrt(hd)
    // (the row maximum is taken on the unscaled scores; the scale is positive), and only the diagonal key tile
    // needs the causal compare -- four VALU operations per score instead of eight, in kernels whose waves are
    // bound by their own instruction stream.
    float mc = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < ATT_MAX_TILES; ++kt) {
      if (kt < nk && !(ABLATE & 4)) {
        if (diag && kt == qtl) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (att_krow(r, lh) > li) s[kt][r] = -INFINITY;
        }
#pragma unroll
        for (int r = 0; r < 16; r += 2) mc = fmaxf(mc, fmaxf(s[kt][r], s[kt][r + 1]));      // v_max3_f32
      }
    }
    mc = fmaxf(mc, __shfl_xor(mc, 32, 64));
    const float mn = fmaxf(m, mc);
    // a query whose keys were ALL masked so far (key bias -inf over a whole leading chunk: a left-padded batch) has
    // mn = -inf: exp2(-inf * c1 + inf) would be NaN and poison the row for good.  Its probabilities are exactly 0 against
    // any finite reference point, so the exponent is taken against 0 until a finite score arrives.
    const float msafe = mn == -INFINITY ? 0.f : mn;
    if (MULTI && c) {
      // online rescale of what the earlier chunks left (one per-lane scalar: a lane owns a query row of O^T)
      const float alpha = __builtin_amdgcn_exp2f((m - msafe) * c1);       // (m = -inf: 0, and nothing had been added)
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
      lsum *= alpha;
    }
    m = mn;
    const float c2 = -msafe * c1;
    float lc = 0.f;
#pragma unroll
    for (int kt = 0; kt < ATT_MAX_TILES; ++kt) {
      if (kt < nk && !(ABLATE & 4)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s[kt][r] = __builtin_amdgcn_exp2f(fmaf(s[kt][r], c1, c2));
          if constexpr (HD >= ATT_LDV) lc += s[kt][r];
        }
      }
    }
    if constexpr (HD >= ATT_LDV) lsum += lc + __shfl_xor(lc, 32, 64);
    ATT_T(4);
    // ---- O^T += V^T P^T  (two 32-row tiles over the head dim, the second one partly padding for hd < 64)
    if (!MULTI) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    }
#pragma unroll
    for (int kt = 0; kt < ATT_MAX_TILES; ++kt) {
      if (kt < nk && !(ABLATE & 8)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float* vrow = Vs + (kt * 32 + att_krow(r, lh)) * ATT_LDV;
          const float a0 = vrow[li];
          const float a1 = vrow[32 + li];            // hd < 64: columns HD.. of the padded row are 1, 0, 0, ...
          o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, s[kt][r], o0, 0, 0, 0);
          o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, s[kt][r], o1, 0, 0, 0);
        }
      }
    }
  }
  ATT_T(5);
  if (!active) return;
  // ---- hd < 64: row HD of O^T (register 4 (HD - 32) / 8 of the lower half-wave's second tile) is the softmax denominator
  float l;
  if constexpr (HD < ATT_LDV) {
    l = o1[(HD - 32) / 2];
    l = __shfl(l, li, 64);
  } else {
    l = lsum;
  }
  // ---- normalise and store: a lane holds 4 consecutive head-dim values of its query row per register group ----
  const float inv_l = 1.f / l;
  if (lh == 0) LSE[(int64_t)bh * L + qpos] = m * inv_sqrt + logf(l);
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
  ATT_T(6);
  ATT_T_END(qtl);
}

// csrc/attention_p.hip
int pdn_attention_p_supported(int L, int head_dim);
int pdn_attention_p_fwd(const float* q, const float* k, const float* v, float* o, float* lse, int B, int H, int L,
                        int head_dim, int64_t row_stride, int64_t batch_stride, int64_t o_row_stride,
                        int64_t o_batch_stride, int causal, void* stream);

int pdn_attention_p_bwd(const float* q, const float* k, const float* v, const float* o, const float* d_o, const float* lse,
                        float* dq, float* dk, float* dv, int B, int H, int L, int head_dim, int64_t row_stride,
                        int64_t batch_stride, int64_t o_row_stride, int64_t o_batch_stride, int causal,
                        const float* rope_cos, const float* rope_sin, float* delta, void* stream);

// csrc/attention_blocks.hip: 512 / 768 / 1024 positions as 256-row block pairs on the persistent kernels
int pdn_attention_blocks_ok(int L, int head_dim);
int pdn_attention_blocks_fwd(const float* q, const float* k, const float* v, float* o, float* lse, int B, int H, int L,
                             int hd, int64_t rs, int64_t bs, int64_t o_rs, int64_t o_bs, int causal, void* stream);
int pdn_attention_blocks_bwd(const float* q, const float* k, const float* v, const float* o, const float* d_o, const float* lse,
                             float* dq, float* dk, float* dv, int B, int H, int L, int hd, int64_t rs, int64_t bs, int64_t o_rs,
                             int64_t o_bs, int causal, const float* rope_cos, const float* rope_sin, float* workspace,
                             void* stream);

// csrc/attention_hd128.hip: head dim 128, any length up to 1024 (four head-dim tiles, 64-row chunks)
bool pdn_attention_hd128_ok(int L, int head_dim);
int pdn_attention_hd128_fwd(const float* q, const float* k, const float* v, float* o, float* lse, int B, int H, int L,
                            int64_t rs, int64_t bs, int64_t ors, int64_t obs, int causal, const float* key_bias,
                            int64_t kb_bs, void* stream);
int pdn_attention_hd128_bwd(const float* q, const float* k, const float* v, const float* o, const float* d_o, const float* lse,
                            float* dq, float* dk, float* dv, int B, int H, int L, int64_t rs, int64_t bs, int64_t ors,
                            int64_t obs, int causal, const float* key_bias, int64_t kb_bs, float* delta, void* stream);

static bool att_shape_ok(int L, int head_dim) {
  if (pdn_attention_hd128_ok(L, head_dim)) return true;
  return (head_dim == 48 || head_dim == 64) && L % 32 == 0 && L >= 32 && L <= ATT_MAX_L;
}
static int att_groups(int L) { return (L / 32 + ATT_MAX_TILES - 1) / ATT_MAX_TILES; }
static int att_crows(int L) { return L < ATT_CHUNK ? L : ATT_CHUNK; }

/* 1 when the resident kernels take (L, head_dim): head_dim 48 / 64, L a multiple of 32 up to 1024 */
extern "C" int pdn_attention_supported(int L, int head_dim) { return att_shape_ok(L, head_dim) ? 1 : 0; }
// 1 when rotation-free operands of this shape run on the persistent kernels (directly, or as 256-row block pairs)
extern "C" int pdn_attention_persistent_supported(int L, int head_dim) {
  return att_shape_ok(L, head_dim) && (pdn_attention_p_supported(L, head_dim) || pdn_attention_blocks_ok(L, head_dim)) ? 1 : 0;
}

extern "C" int64_t pdn_attention_lds_bytes(int L, int head_dim) {
  return (int64_t)att_crows(L) * (ATT_LD(head_dim) + ATT_LDV) * 4;
}

// q, k, v (and dq, dk, dv): (B, L, H, head_dim) contiguous in head_dim, `row_stride` between consecutive
// positions, `batch_stride` between batches -- e.g. column blocks of one packed (B*L, 3*H*hd) projection;
// o (and d_o) have strides of their own.  lse: (B, H, L).  causal: keys > query masked.
static int att_fwd_impl(const float* q, const float* k, const float* v, float* o,
                        float* lse, int B, int H, int L, int head_dim,
                        int64_t row_stride, int64_t batch_stride, int64_t o_row_stride,
                        int64_t o_batch_stride, int causal,
                        const float* rope_cos, const float* rope_sin, const float* key_bias, int64_t kb_bs, void* stream) {
  if (B == 0 || H == 0 || L == 0) return PDN_OK;
  PDN_CHECK_ARG(q && k && v && o && lse, "pdn_attention_fwd_f32: null operand");
  PDN_CHECK_ARG((rope_cos == nullptr) == (rope_sin == nullptr) &&
                    ((((uintptr_t)rope_cos | (uintptr_t)rope_sin) & 7) == 0),
                "pdn_attention_fwd_f32: rope tables must come as an 8-byte aligned pair");
  if (!att_shape_ok(L, head_dim)) {
    pdn_set_error("pdn_attention_fwd_f32: fused path supports head_dim 48 / 64, L a multiple of 32 and <= %d", ATT_MAX_L);
    return PDN_EUNSUPPORTED;
  }
  PDN_CHECK_ARG((row_stride % 4) == 0 && (batch_stride % 4) == 0 && (o_row_stride % 4) == 0 &&
                    (o_batch_stride % 4) == 0 &&
                    ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15) == 0),
                "pdn_attention_fwd_f32: 16-byte alignment required");
  if (pdn_attention_hd128_ok(L, head_dim)) {
    PDN_CHECK_ARG(!rope_cos, "pdn_attention_fwd_f32: no RoPE inside the head-dim-128 kernels");
    return pdn_attention_hd128_fwd(q, k, v, o, lse, B, H, L, row_stride, batch_stride, o_row_stride, o_batch_stride, causal,
                                   key_bias, kb_bs, stream);
  }
  // rotation-free operands at the benchmark shape class: the persistent, DMA-staged kernels (csrc/attention_p.hip)
  if (!rope_cos && !key_bias && pdn_attention_p_supported(L, head_dim))
    return pdn_attention_p_fwd(q, k, v, o, lse, B, H, L, head_dim, row_stride, batch_stride, o_row_stride, o_batch_stride,
                               causal, stream);
  // ... and longer sequences as 256-row block pairs on the same kernels (csrc/attention_blocks.hip)
  if (!rope_cos && !key_bias && pdn_attention_blocks_ok(L, head_dim) && (int64_t)L * row_stride < (1ll << 31))
    return pdn_attention_blocks_fwd(q, k, v, o, lse, B, H, L, head_dim, row_stride, batch_stride, o_row_stride, o_batch_stride,
                                    causal, stream);
  const size_t shm = (size_t)pdn_attention_lds_bytes(L, head_dim);
  static bool attr_set = false;
  if (!attr_set) {
    PDN_HIP(hipFuncSetAttribute((const void*)attention_fwd_kernel<48, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PDN_HIP(hipFuncSetAttribute((const void*)attention_fwd_kernel<64, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PDN_HIP(hipFuncSetAttribute((const void*)attention_fwd_kernel<48, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PDN_HIP(hipFuncSetAttribute((const void*)attention_fwd_kernel<64, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const dim3 grid((unsigned)(B * H * att_groups(L)));
  const float sq = sqrtf((float)head_dim);
  const bool multi = L > ATT_CHUNK;
#define ATT_FWD(HD_, M_)                                                                                              \
  hipLaunchKernelGGL((attention_fwd_kernel<HD_, 0, M_>), grid, dim3(512), shm, (hipStream_t)stream, q, k, v, o, lse, H, L, \
                     row_stride, batch_stride, o_row_stride, o_batch_stride, sq, causal, rope_cos, rope_sin, key_bias, kb_bs)
  if (head_dim == 48) { if (multi) ATT_FWD(48, true); else ATT_FWD(48, false); }
  else { if (multi) ATT_FWD(64, true); else ATT_FWD(64, false); }
#undef ATT_FWD
  pdn_count(PDN_CNT_ATT_RES_FWD);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

extern "C" int pdn_attention_fwd_f32(const float* q, const float* k, const float* v, float* o,
                                     float* lse, int B, int H, int L, int head_dim,
                                     int64_t row_stride, int64_t batch_stride, int64_t o_row_stride,
                                     int64_t o_batch_stride, int causal,
                                     const float* rope_cos, const float* rope_sin, void* stream) {
  return att_fwd_impl(q, k, v, o, lse, B, H, L, head_dim, row_stride, batch_stride, o_row_stride, o_batch_stride, causal,
                      rope_cos, rope_sin, nullptr, 0, stream);
}
// The same with an additive KEY bias (B x L, `kb_batch_stride` floats between batches, 0 = one vector for all): what the
// reference adds to the scores as a (B, 1, 1, L) padding mask (examples/pydynet/transformer.py:92-96; -inf = masked),
// and how keys beyond a length that is not a multiple of 32 are switched off.  No RoPE inside.
extern "C" int pdn_attention_fwd_bias_f32(const float* q, const float* k, const float* v, float* o, float* lse, int B,
                                          int H, int L, int head_dim, int64_t row_stride, int64_t batch_stride,
                                          int64_t o_row_stride, int64_t o_batch_stride, int causal, const float* key_bias,
                                          int64_t kb_batch_stride, void* stream) {
  PDN_CHECK_ARG(key_bias, "pdn_attention_fwd_bias_f32: null key bias");
  return att_fwd_impl(q, k, v, o, lse, B, H, L, head_dim, row_stride, batch_stride, o_row_stride, o_batch_stride, causal,
                      nullptr, nullptr, key_bias, kb_batch_stride, stream);
}

// ======================================================================================
// Backward.  Probabilities are recomputed from the saved log-sum-exp:  P = exp(S/sqrt(hd) - lse).
//   delta[q] = sum_d dO[q,d] * O[q,d]
//   dV = P^T dO        dP = dO V^T        dS = P o (dP - delta) / sqrt(hd)
//   dQ = dS K          dK = dS^T Q
// An MFMA accumulator holds its column index in the lane and its row index in registers, and
// can feed the next MFMA only as the operand that contracts over the ROW index.  dQ contracts
// over keys, dK / dV over queries, so the score tile is needed in both orientations -- two
// kernels, each one workgroup of 8 waves per (batch, head, group of 8 tiles) with the same zig-zag tile
// ownership as the forward, the other side passing through LDS in chunks of 256 rows:
//   attention_bwd_dq_kernel   K, V chunks in LDS; a wave owns a query tile:
//        S^T[key][q] = K Q^T, dP^T = V dO^T  ->  dQ^T += K^T dS^T;  also writes delta[q]
//   attention_bwd_dkv_kernel  Q, dO chunks in LDS; a wave owns a key tile:
//        S[q][key] = Q K^T, dP = dO V^T  ->  dV^T += dO^T P,  dK^T += Q^T dS
// 80 + 112 MFMAs per (query tile, key tile) pair at hd 48, fully masked pairs skipped.  Nothing of size
// L x L touches HBM; the only intermediate is delta (B*H*L floats of workspace).  The accumulators simply carry
// from chunk to chunk (P is recomputed from the saved log-sum-exp: no rescale).
// ======================================================================================
template <int HD, bool MULTI>
__global__ __launch_bounds__(512, 1) void attention_bwd_dq_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
    const float* __restrict__ O, const float* __restrict__ dO, const float* __restrict__ LSE,
    float* __restrict__ dQ, float* __restrict__ Delta, int H, int L, int64_t row_stride,
    int64_t batch_stride, int64_t o_row_stride, int64_t o_batch_stride, float sqrt_hd, int causal,
    const float* __restrict__ RC, const float* __restrict__ RS, int prerot, const float* __restrict__ KB, int64_t kb_bs) {
  constexpr int LD = ATT_LD(HD);
  constexpr int NT8 = HD / 8;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int LDK = ATT_LDP;
  const int crows = min(L, ATT_CHUNK);
  float* Ks = lds;                                 // [crows][68]: K | 0 (read along the row for S, down the columns for dQ)
  float* Vs = Ks + (size_t)crows * LDK;            // [crows][LD]

  const int ntile = L / 32, G = MULTI ? (ntile + ATT_MAX_TILES - 1) / ATT_MAX_TILES : 1;
  int bh, qg;
  if (MULTI) att_block(G, bh, qg); else { bh = blockIdx.x; qg = 0; }
  const int b = bh / H, h = bh % H;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int64_t base = (int64_t)b * batch_stride + (int64_t)h * HD;
  const int64_t obase = (int64_t)b * o_batch_stride + (int64_t)h * HD;
  const float* Qb = Q + base; const float* Ob = O + obase; const float* dOb = dO + obase;
  float* dQb = dQ + base;

  const int qtl = att_tile_of_wave(wave);
  const int qt = qg * ATT_MAX_TILES + qtl;
  const bool active = qt < ntile;
  AttRowRope<NT8> rr;
  const bool pre = !MULTI && RC != nullptr;                 // (the chunk loops have no registers to spare for it)
  const int qpos_l = (active ? qt : 0) * 32 + li;           // (an idle wave preloads tile 0: unconditional loads, see forward)
  if constexpr (!MULTI) rr.load(RC ? RC : Q, RC ? RS : Q, RC ? qpos_l : 0, lh, HD / 2);   // (no RoPE: reads Q, unused)
  const float inv_sqrt = 1.f / sqrt_hd;
  const int qpos = qt * 32 + li;
  // this wave's operands (Q, dO, O rows of its tile, lse): the loads go out BEFORE the staging -- behind which their
  // latency hides -- and are consumed after the barrier (six workgroups pass through a CU one after the other: every
  // exposed round trip is paid six times per kernel)
  float4 qf[NT8], gf[NT8], of_[NT8];
  float dpart = 0.f, lse_q = 0.f;
  {
    const float* qrow = Qb + (int64_t)qpos_l * row_stride + 4 * lh;
    const float* grow = dOb + (int64_t)qpos_l * o_row_stride + 4 * lh;
    const float* orow = Ob + (int64_t)qpos_l * o_row_stride + 4 * lh;
#pragma unroll
    for (int t = 0; t < NT8; ++t) {
      qf[t] = *reinterpret_cast<const float4*>(qrow + 8 * t);
      gf[t] = *reinterpret_cast<const float4*>(grow + 8 * t);
      of_[t] = *reinterpret_cast<const float4*>(orow + 8 * t);
    }
    lse_q = LSE[(int64_t)bh * L + qpos_l];
  }
  if (!MULTI) {
    att_stage_two_pad<HD, 512, true, false>(Ks, Vs, K + base, V + base, L, 0, row_stride, row_stride, tid, RC, RS, !prerot, false);
    __syncthreads();
    if (!active) return;                           // no workgroup barrier below this point
  }
  if (active) {
#pragma unroll
    for (int t = 0; t < NT8; ++t) {
      if (prerot) {}                                // (q, k come rotated: only dq / dk are rotated back at the store)
      else if (pre) qf[t] = rr.rot(qf[t], t, 1.f);
      else if (RC) qf[t] = att_rot(qf[t], RC, RS, qpos, 4 * t + 2 * lh, HD / 2, 1.f);
      dpart += (of_[t].x * gf[t].x + of_[t].y * gf[t].y) + (of_[t].z * gf[t].z + of_[t].w * gf[t].w);
    }
  }
  const float delta_q = dpart + __shfl_xor(dpart, 32, 64);
  const float c1 = inv_sqrt * 1.4426950408889634f, c2q = -lse_q * 1.4426950408889634f;
  if (active && lh == 0) Delta[(int64_t)bh * L + qpos] = delta_q;
  f32x16 dq0, dq1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dq0[r] = 0.f; dq1[r] = 0.f; }
  const int c_last = MULTI ? (causal ? qg : G - 1) : 0;
  for (int c = 0; c <= c_last; ++c) {
    const int row0 = c * ATT_CHUNK, nrows = min(ATT_CHUNK, L - row0);
    if (MULTI) {
      if (c) __syncthreads();
      att_stage_pad_call<HD, false>(Ks, Vs, K + base + (int64_t)row0 * row_stride, V + base + (int64_t)row0 * row_stride,
                                    nrows, row0, row_stride, row_stride, tid, RC, RS, !prerot);
      __syncthreads();
      if (!active) continue;
    }
    const bool diag = causal && c == qg;
    const int nk = diag ? qtl + 1 : nrows / 32;
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
      if (KB) {                                     // additive key bias: the rank-1 step of the forward kernel
        const float kb = lh == 0 ? KB[(int64_t)b * kb_bs + row0 + kt * 32 + li] * sqrt_hd : 0.f;
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kb, lh == 0 ? 1.f : 0.f, s, 0, 0, 0);
      }
      // dS^T[key][q] = P^T o (dP^T - delta_q) / sqrt(hd)   (lane = q);  P = exp2(s * c1 - lse * log2(e)), the causal
      // compare only on the diagonal tile
      if (diag && kt == qtl) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (att_krow(r, lh) > li) s[r] = -INFINITY;
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
        const float a1 = kr[32 + li];                    // hd < 64: columns HD .. 63 of the padded row are zero
        dq0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, s[r], dq0, 0, 0, 0);
        dq1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, s[r], dq1, 0, 0, 0);
      }
    }
  }
  if (active) {
    if (pre) att_store_rows_pre<HD>(dq0, dq1, dQb + (int64_t)qpos * row_stride + 4 * lh, inv_sqrt, rr);
    else att_store_rows<HD>(dq0, dq1, dQb + (int64_t)qpos * row_stride + 4 * lh, lh, inv_sqrt, RC, RS, qpos);
  }
}

template <int HD, bool MULTI>
__global__ __launch_bounds__(512, 1) void attention_bwd_dkv_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
    const float* __restrict__ dO, const float* __restrict__ LSE, const float* __restrict__ Delta,
    float* __restrict__ dK, float* __restrict__ dV, int H, int L, int64_t row_stride,
    int64_t batch_stride, int64_t o_row_stride, int64_t o_batch_stride, float sqrt_hd, int causal,
    const float* __restrict__ RC, const float* __restrict__ RS, int prerot, const float* __restrict__ KB, int64_t kb_bs) {
  constexpr int NT8 = HD / 8;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int LDP = ATT_LDP;
  const int crows = min(L, ATT_CHUNK);
  float* Qs = lds;                                 // [crows][68]: Q | 0
  float* Gs = Qs + (size_t)crows * LDP;            // [crows][68]: dO | 0
  float* lse_s = Gs + (size_t)crows * LDP;         // [crows]
  float* delta_s = lse_s + crows;                  // [crows]

  const int ntile = L / 32, G = MULTI ? (ntile + ATT_MAX_TILES - 1) / ATT_MAX_TILES : 1;
  int bh, kg;
  // (a causal key group g visits query chunks g .. G - 1: the FIRST groups are the heavy ones here)
  bh = blockIdx.x / G; kg = (int)(blockIdx.x % G);
  const int b = bh / H, h = bh % H;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int64_t base = (int64_t)b * batch_stride + (int64_t)h * HD;
  const int64_t obase = (int64_t)b * o_batch_stride + (int64_t)h * HD;
  const float* Kb = K + base; const float* Vb = V + base;
  float* dKb = dK + base; float* dVb = dV + base;

  const int ktl = att_tile_of_wave(wave);
  const int kt = kg * ATT_MAX_TILES + ktl;
  const bool active = kt < ntile;
  AttRowRope<NT8> rr;
  const bool pre = !MULTI && RC != nullptr;
  const int kpos_l = (active ? kt : 0) * 32 + li;           // (an idle wave preloads tile 0: unconditional loads, see forward)
  if constexpr (!MULTI) rr.load(RC ? RC : K, RC ? RS : K, RC ? kpos_l : 0, lh, HD / 2);   // (no RoPE: reads K, unused)
  const float inv_sqrt = 1.f / sqrt_hd;
  const float c1 = inv_sqrt * 1.4426950408889634f;
  const int kpos = kt * 32 + li;
  // additive key bias of this wave's key tile (lane = key): the B operand of one extra rank-1 step per score tile
  const float kbias = (KB && lh == 0) ? KB[(int64_t)b * kb_bs + kpos_l] * sqrt_hd : 0.f;
  float4 kf[NT8], vf[NT8];                          // (issued before the staging, consumed after the barrier: see dQ)
  {
    const float* krow = Kb + (int64_t)kpos_l * row_stride + 4 * lh;
    const float* vrow = Vb + (int64_t)kpos_l * row_stride + 4 * lh;
#pragma unroll
    for (int t = 0; t < NT8; ++t) {
      kf[t] = *reinterpret_cast<const float4*>(krow + 8 * t);
      vf[t] = *reinterpret_cast<const float4*>(vrow + 8 * t);
    }
  }
  if (!MULTI) {
    att_stage_two_pad<HD, 512, true, true>(Qs, Gs, Q + base, dO + obase, L, 0, row_stride, o_row_stride, tid, RC, RS, !prerot,
                                           false);
    for (int q = tid; q < L; q += 512) {
      lse_s[q] = LSE[(int64_t)bh * L + q] * 1.4426950408889634f;
      delta_s[q] = Delta[(int64_t)bh * L + q];
    }
    __syncthreads();
    if (!active) return;                           // no workgroup barrier below this point
  }
  if (active) {
#pragma unroll
    for (int t = 0; t < NT8; ++t) {
      if (prerot) {}
      else if (pre) kf[t] = rr.rot(kf[t], t, 1.f);
      else if (RC) kf[t] = att_rot(kf[t], RC, RS, kpos, 4 * t + 2 * lh, HD / 2, 1.f);
    }
  }
  f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dk0[r] = 0.f; dk1[r] = 0.f; dv0[r] = 0.f; dv1[r] = 0.f; }
  const int c_first = (MULTI && causal) ? kg : 0;
  for (int c = c_first; c < G; ++c) {
    const int row0 = c * ATT_CHUNK, nrows = min(ATT_CHUNK, L - row0);
    if (MULTI) {
      if (c != c_first) __syncthreads();
      att_stage_pad_call<HD, true>(Qs, Gs, Q + base + (int64_t)row0 * row_stride, dO + obase + (int64_t)row0 * o_row_stride,
                                   nrows, row0, row_stride, o_row_stride, tid, RC, RS, !prerot);
      for (int q = tid; q < nrows; q += 512) {
        lse_s[q] = LSE[(int64_t)bh * L + row0 + q] * 1.4426950408889634f;
        delta_s[q] = Delta[(int64_t)bh * L + row0 + q];
      }
      __syncthreads();
      if (!active) continue;
    }
    const bool diag = causal && c == kg;
    const int q_first = diag ? ktl : 0;            // query tiles (inside the chunk) that can see this key tile
    for (int qt = q_first; qt < nrows / 32; ++qt) {
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
      if (KB) s = __builtin_amdgcn_mfma_f32_32x32x2f32(lh == 0 ? 1.f : 0.f, kbias, s, 0, 0, 0);
      // lane = key, registers = queries;  P = exp2(s * c1 - lse * log2(e)) (lse_s holds lse * log2(e)), the causal
      // compare only on the diagonal tile
      if (diag && qt == ktl) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (li > att_krow(r, lh)) s[r] = -INFINITY;
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
        const float g1 = Gs[qr * LDP + 32 + li], q1 = Qs[qr * LDP + 32 + li];     // hd < 64: zero beyond the head dim
        dv0 = __builtin_amdgcn_mfma_f32_32x32x2f32(g0, s[r], dv0, 0, 0, 0);     // dV^T += dO^T P
        dk0 = __builtin_amdgcn_mfma_f32_32x32x2f32(q0, dp[r], dk0, 0, 0, 0);    // dK^T += Q^T dS
        dv1 = __builtin_amdgcn_mfma_f32_32x32x2f32(g1, s[r], dv1, 0, 0, 0);
        dk1 = __builtin_amdgcn_mfma_f32_32x32x2f32(q1, dp[r], dk1, 0, 0, 0);
      }
    }
  }
  if (active) {
    if (pre) att_store_rows_pre<HD>(dk0, dk1, dKb + (int64_t)kpos * row_stride + 4 * lh, inv_sqrt, rr);
    else att_store_rows<HD>(dk0, dk1, dKb + (int64_t)kpos * row_stride + 4 * lh, lh, inv_sqrt, RC, RS, kpos);
    att_store_rows<HD>(dv0, dv1, dVb + (int64_t)kpos * row_stride + 4 * lh, lh, 1.f);
  }
}

extern "C" int64_t pdn_attention_bwd_lds_bytes(int L, int head_dim) {
  (void)head_dim;
  const int64_t c = att_crows(L);
  return 2 * c * ATT_LDP * 4 + 2 * c * 4;                            // Q | 0 and dO | 0 images + lse, delta
}
static int64_t att_dq_lds_bytes(int L, int head_dim) { return (int64_t)att_crows(L) * (ATT_LDP + ATT_LD(head_dim)) * 4; }

// delta[b, h, q] = sum_d dO * O is produced by the dQ kernel and consumed by the dK/dV kernel
extern "C" int64_t pdn_attention_bwd_workspace_bytes(int B, int H, int L) {
  return (int64_t)B * H * L * 4;
}

static int att_bwd_impl(const float* q, const float* k, const float* v, const float* o,
                        const float* d_o, const float* lse, float* dq, float* dk,
                        float* dv, int B, int H, int L, int head_dim,
                        int64_t row_stride, int64_t batch_stride, int64_t o_row_stride,
                        int64_t o_batch_stride, int causal,
                        const float* rope_cos, const float* rope_sin, void* workspace,
                        int64_t workspace_bytes, void* stream, int prerot, const float* key_bias = nullptr,
                        int64_t kb_bs = 0) {
  if (B == 0 || H == 0 || L == 0) return PDN_OK;
  PDN_CHECK_ARG((rope_cos == nullptr) == (rope_sin == nullptr) &&
                    ((((uintptr_t)rope_cos | (uintptr_t)rope_sin) & 7) == 0),
                "pdn_attention_bwd_f32: rope tables must come as an 8-byte aligned pair");
  PDN_CHECK_ARG(q && k && v && o && d_o && lse && dq && dk && dv, "pdn_attention_bwd_f32: null operand");
  if (!att_shape_ok(L, head_dim)) {
    pdn_set_error("pdn_attention_bwd_f32: fused path supports head_dim 48 / 64, L a multiple of 32 and <= %d", ATT_MAX_L);
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
  if (pdn_attention_hd128_ok(L, head_dim)) {
    PDN_CHECK_ARG(!rope_cos, "pdn_attention_bwd_f32: no RoPE inside the head-dim-128 kernels");
    return pdn_attention_hd128_bwd(q, k, v, o, d_o, lse, dq, dk, dv, B, H, L, row_stride, batch_stride, o_row_stride,
                                   o_batch_stride, causal, key_bias, kb_bs, delta, stream);
  }
  // operands that need no rotation on the way in (never rotated, or rotated by the projection's epilogue) at the
  // benchmark shape class: the persistent, DMA-staged kernels (csrc/attention_p.hip); dq / dk are rotated back there
  // when the tables are given
  if ((prerot || !rope_cos) && !key_bias && pdn_attention_p_supported(L, head_dim))
    return pdn_attention_p_bwd(q, k, v, o, d_o, lse, dq, dk, dv, B, H, L, head_dim, row_stride, batch_stride, o_row_stride,
                               o_batch_stride, causal, rope_cos, rope_sin, delta, stream);
  if ((prerot || !rope_cos) && !key_bias && pdn_attention_blocks_ok(L, head_dim) && (int64_t)L * row_stride < (1ll << 31))
    return pdn_attention_blocks_bwd(q, k, v, o, d_o, lse, dq, dk, dv, B, H, L, head_dim, row_stride, batch_stride, o_row_stride,
                                    o_batch_stride, causal, rope_cos, rope_sin, delta, stream);
  static bool attr_set = false;
  if (!attr_set) {
#define ATT_ATTR(K_) PDN_HIP(hipFuncSetAttribute((const void*)K_, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
    ATT_ATTR((attention_bwd_dq_kernel<48, false>)); ATT_ATTR((attention_bwd_dkv_kernel<48, false>));
    ATT_ATTR((attention_bwd_dq_kernel<64, false>)); ATT_ATTR((attention_bwd_dkv_kernel<64, false>));
    ATT_ATTR((attention_bwd_dq_kernel<48, true>)); ATT_ATTR((attention_bwd_dkv_kernel<48, true>));
    ATT_ATTR((attention_bwd_dq_kernel<64, true>)); ATT_ATTR((attention_bwd_dkv_kernel<64, true>));
#undef ATT_ATTR
    attr_set = true;
  }
  const float sq = sqrtf((float)head_dim);
  const dim3 grid((unsigned)(B * H * att_groups(L)));
  hipStream_t st = (hipStream_t)stream;
#define ATT_BWD(HD_, M_)                                                                                               \
  hipLaunchKernelGGL((attention_bwd_dq_kernel<HD_, M_>), grid, dim3(512), (size_t)att_dq_lds_bytes(L, head_dim), st, q, k, v, \
                     o, d_o, lse, dq, delta, H, L, row_stride, batch_stride, o_row_stride, o_batch_stride, sq, causal,    \
                     rope_cos, rope_sin, prerot, key_bias, kb_bs);                                                        \
  PDN_LAUNCH_CHECK();                                                                                                     \
  hipLaunchKernelGGL((attention_bwd_dkv_kernel<HD_, M_>), grid, dim3(512), (size_t)pdn_attention_bwd_lds_bytes(L, head_dim),  \
                     st, q, k, v, d_o, lse, delta, dk, dv, H, L, row_stride, batch_stride, o_row_stride, o_batch_stride,  \
                     sq, causal, rope_cos, rope_sin, prerot, key_bias, kb_bs);                                            \
  PDN_LAUNCH_CHECK();
  const bool multi = L > ATT_CHUNK;
  pdn_count(PDN_CNT_ATT_RES_BWD);
  if (head_dim == 48) { if (multi) { ATT_BWD(48, true) } else { ATT_BWD(48, false) } }
  else { if (multi) { ATT_BWD(64, true) } else { ATT_BWD(64, false) } }
#undef ATT_BWD
  return PDN_OK;
}

extern "C" int pdn_attention_bwd_f32(const float* q, const float* k, const float* v, const float* o,
                                     const float* d_o, const float* lse, float* dq, float* dk,
                                     float* dv, int B, int H, int L, int head_dim,
                                     int64_t row_stride, int64_t batch_stride, int64_t o_row_stride,
                                     int64_t o_batch_stride, int causal,
                                     const float* rope_cos, const float* rope_sin, void* workspace,
                                     int64_t workspace_bytes, void* stream) {
  return att_bwd_impl(q, k, v, o, d_o, lse, dq, dk, dv, B, H, L, head_dim, row_stride, batch_stride, o_row_stride,
                      o_batch_stride, causal, rope_cos, rope_sin, workspace, workspace_bytes, stream, 0);
}
extern "C" int pdn_attention_bwd_bias_f32(const float* q, const float* k, const float* v, const float* o,
                                          const float* d_o, const float* lse, float* dq, float* dk, float* dv, int B,
                                          int H, int L, int head_dim, int64_t row_stride, int64_t batch_stride,
                                          int64_t o_row_stride, int64_t o_batch_stride, int causal, const float* key_bias,
                                          int64_t kb_batch_stride, void* workspace, int64_t workspace_bytes, void* stream) {
  PDN_CHECK_ARG(key_bias, "pdn_attention_bwd_bias_f32: null key bias");
  return att_bwd_impl(q, k, v, o, d_o, lse, dq, dk, dv, B, H, L, head_dim, row_stride, batch_stride, o_row_stride,
                      o_batch_stride, causal, nullptr, nullptr, workspace, workspace_bytes, stream, 0, key_bias,
                      kb_batch_stride);
}
// q and k are given ALREADY ROTATED (the q | k | v projection applied RoPE in its epilogue, pdn_qkv_rope_fwd_f32);
// dq and dk are still rotated back to the un-rotated projections' gradients as they are stored.
extern "C" int pdn_attention_bwd_rotated_f32(const float* q, const float* k, const float* v, const float* o,
                                             const float* d_o, const float* lse, float* dq, float* dk,
                                             float* dv, int B, int H, int L, int head_dim,
                                             int64_t row_stride, int64_t batch_stride, int64_t o_row_stride,
                                             int64_t o_batch_stride, int causal,
                                             const float* rope_cos, const float* rope_sin, void* workspace,
                                             int64_t workspace_bytes, void* stream) {
  PDN_CHECK_ARG(rope_cos && rope_sin, "pdn_attention_bwd_rotated_f32: the rope tables are required");
  return att_bwd_impl(q, k, v, o, d_o, lse, dq, dk, dv, B, H, L, head_dim, row_stride, batch_stride, o_row_stride,
                      o_batch_stride, causal, rope_cos, rope_sin, workspace, workspace_bytes, stream, 1);
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
