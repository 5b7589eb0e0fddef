// Persistent, DMA-staged causal self-attention for the benchmark shape class (L <= 256, head dim 48), fp32 MFMA, gfx950.
//
// Same mathematics, tile ownership and MFMA formulation as csrc/attention.hip (llm/llama/model.py:112-121: S^T = K Q^T
// with lane = query, the P^T accumulators fed as they are to O^T = V^T P^T; backward from the saved log-sum-exp).  What
// changes is how a head's operands reach LDS and what a CU does meanwhile.  In attention.hip a 118 KB workgroup stages
// K and V through registers, computes, stores and retires: six workgroups pass through a CU one after the other and
// nothing overlaps the 6-9 us of staging of each (profiles/r03_attention_fwd_trace.txt).  Here
//   * q and k arrive ALREADY ROTATED (RoPE rides in the store of the q | k | v projection, csrc/gemm_rowres.hip EPI 3,
//     or was applied by a `rope` node), so staging is a plain copy: `global_load_lds_dwordx4` moves rows from HBM to
//     LDS without touching a VGPR or a VALU instruction;
//   * one workgroup per CU is PERSISTENT over heads (bh = blockIdx.x, += gridDim.x) and the LDS holds three [256][48]
//     images: while the S^T / softmax phase of head n reads K_n, V_n is in flight; while its P V phase reads V_n,
//     K_{n+1} is in flight into the other K buffer and the wave's own Q rows of head n + 1 into registers -- two
//     workgroup barriers per head, no staging phase, no workgroup turnover;
//   * LDS images are DENSE (a DMA instruction writes 64 consecutive 16-byte units): rows read along the row as MFMA
//     operands (ds_read_b128, lane = row) have their 12 units ROTATED by (row >> 2) & 3 -- applied on the source side
//     of the DMA -- which makes 16 consecutive rows hit 16 different bank groups; images read down the columns (lane
//     = head-dim index, ds_read_b32) need nothing.  The second 32-row head-dim tile of the O^T / dQ^T / dK^T / dV^T
//     products reads 16 floats past its row (the next row's data): MFMA output rows are independent, so only rows
//     48 .. 63 of that tile see them, and those are never stored.
#include "common.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define AP_ROWS 256

// 64 lanes x 16 bytes from HBM straight into LDS (lane i lands at `l` + 16 i; `l` wave-uniform).  Inline asm on purpose:
// behind the builtin, hipcc (ROCm 7.2) treats every later LDS read it cannot prove disjoint as dependent on the DMA
// and puts `s_waitcnt vmcnt(0)` in front of it -- the prefetched image was waited for at the first tile of the phase
// it was meant to hide behind.  The landing of an image is ordered by hand instead (ap_landed + ap_barrier).
__device__ __forceinline__ void ap_glds16(const float* g, float* l) {
  const unsigned a = (unsigned)(unsigned long)((__attribute__((address_space(3))) void*)l);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(a), "v"(g) : "memory");
}
__device__ __forceinline__ int ap_krow(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }
__device__ __forceinline__ int ap_tile_of_wave(int w) { return w < 4 ? w : 11 - w; }   // SIMD s hosts tiles s and 7 - s

// One [256][HD] image: 256 * HD / 4 units of 16 bytes = NI wave instructions, 8 waves x NI / 8 each (rows >= L re-read
// row L - 1: every wave always issues the same number of instructions, and nothing is written past the image).
// ROT: unit `cu` of row r lands in slot (cu + ((r >> 2) & 3)) mod UPR of that row.
// Q0 / Q1: the slice [Q0, Q1) of a wave's NI / 8 instructions -- instruction q of every wave together cover rows
// 128 q / 3 .. 128 (q + 1) / 3, so [0, 3) is the first 128 rows of a 48-wide image and [3, 6) the rest.
template <int HD, bool ROT, int Q0 = 0, int Q1 = AP_ROWS * (HD / 4) / 64 / 8>
__device__ __forceinline__ void ap_dma_image(float* __restrict__ dst, const float* __restrict__ g, int L, int rs, int wave,
                                             int lane) {
  constexpr int UPR = HD / 4, NI = AP_ROWS * UPR / 64;
  static_assert(NI % 8 == 0, "instructions divide over 8 waves");
#pragma unroll
  for (int q = Q0; q < Q1; ++q) {
    const int I = wave + 8 * q, U = 64 * I + lane;
    const int row = UPR == 12 ? (U * 43691) >> 19 : U / UPR;
    int cu = U - UPR * row;
    if (ROT) { cu -= (row >> 2) & 3; cu += cu < 0 ? UPR : 0; }
    const int rowc = min(row, L - 1);
    ap_glds16(g + (unsigned)(rowc * rs + 4 * cu), dst + 256 * I);
  }
}

// Waits and barriers.  `ap_landed()` = everything this wave has in flight (its DMA share, its operand loads) has
// landed -- the builtin, not inline asm, so that the compiler's own scoreboard knows it and does not drain the NEXT
// DMA burst at the first use of an already-loaded register (hipcc waits vmcnt(0) for an ordinary load it still counts
// as pending whenever an LDS-DMA is in flight).  It is placed in front of a head's STORES, never behind them: loads
// and stores share the counter but complete independently, so a counted wait cannot tell "the loads are back" from
// "a store was acknowledged early", and vmcnt(0) behind the stores would wait a full write round trip per head.
// `ap_barrier()` is a bare workgroup barrier for LDS traffic (no vmcnt wait: stores and DMA stay in flight across it).
__device__ __forceinline__ void ap_landed() {
  __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0), expcnt / lgkmcnt untouched
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void ap_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void ap_sync_all() { ap_landed(); ap_barrier(); }

// Row fragments of a ROTATED image (ds_read_b128, lane = row li of a tile): fragment t = logical 16-byte unit 2 t + lh
// sits in slot (b0 + 2 t) mod UPR with b0 = lh + ((li >> 2) & 3) <= 4.  Only t = 4 (b0 = 4) and t = 5 (b0 >= 2) wrap
// for UPR = 12, so three registers describe all six offsets and t < 4 are immediates of the first.
struct ApRowOff {
  int b, k4, k5;
  __device__ __forceinline__ void init(int li, int lh, int HD) {
    const int b0 = lh + ((li >> 2) & 3);
    b = li * HD + 4 * b0;
    k4 = b + 32 - (b0 == 4 ? 48 : 0);
    k5 = b + 40 - (b0 >= 2 ? 48 : 0);
  }
  __device__ __forceinline__ int at(int t) const { return t < 4 ? b + 8 * t : (t == 4 ? k4 : k5); }
};

// Column reads of a ROTATED image (lane = head-dim index d, ds_read_b32: 32 banks, lane groups {0-31}, {32-63}):
// element d of row kt * 32 + krow(r, lh), whose rotation is (2 (r >> 2) + lh) & 3 -- one offset per register group
// g = r >> 2, relative to row kt * 32 + (r & 3) + 8 g.  c0: d = li (32 consecutive floats of the row: no conflicts);
// c1: d = 32 + li for li < 16, and lanes 16 .. 31 -- rows 48 .. 63 of the padded tile, never stored -- re-read lane
// li - 16's address (a broadcast) instead of the 16 floats behind the row, which would share banks with it.
// The rotation only takes two values per lane (g even / odd), and (li >> 2) + f < UPR never wraps in the first tile:
// three registers, the rest immediates.
template <int HD>
struct ApColOff {
  int c0e, c1e, c1o;
  __device__ __forceinline__ void init(int li, int lh) {
    constexpr int UPR = HD / 4;
    const int l1 = li & 15;
    c0e = 4 * lh * HD + 4 * ((li >> 2) + lh) + (li & 3);                   // f = lh (g even), lh + 2 (g odd): + 8 floats
    c1e = 4 * lh * HD + 4 * ((8 + (l1 >> 2) + lh) % UPR) + (l1 & 3);
    c1o = 4 * lh * HD + 4 * ((8 + (l1 >> 2) + lh + 2) % UPR) + (l1 & 3);
  }
  __device__ __forceinline__ int first(int g) const { return c0e + 8 * (g & 1); }
  __device__ __forceinline__ int second(int g) const { return (g & 1) ? c1o : c1e; }
};

template <int HD>
__global__ __launch_bounds__(512, 1) void attention_p_fwd_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V, float* __restrict__ O,
    float* __restrict__ LSE, int BH, int H, int L, int64_t row_stride, int64_t batch_stride, int64_t o_row_stride,
    int64_t o_batch_stride, float sqrt_hd, int causal) {
  static_assert(HD == 48, "three [256][HD] images fit 160 KB of LDS only for head dim 48");
  constexpr int UPR = HD / 4, NT8 = HD / 8, IMG = AP_ROWS * HD;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Vs = lds + 2 * IMG;                        // lds: K (even heads) | K (odd heads) | V | 64 bytes of slack
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int ntile = L / 32, rs = (int)row_stride;
  const int qtl = ap_tile_of_wave(wave);
  const bool active = qtl < ntile;
  const int qpos = (active ? qtl : 0) * 32 + li;    // (an idle wave reads tile 0: unconditional loads)
  const int nk = active ? (causal ? qtl + 1 : ntile) : 0;
  const float inv_sqrt = 1.f / sqrt_hd, c1 = inv_sqrt * 1.4426950408889634f;
  ApRowOff ko;                                      // the lane's K fragments: row li of a tile, rotated units
  ko.init(li, lh, HD);

  int bh = blockIdx.x;
  if (bh >= BH) return;
  auto head_base = [&](int x) { return (int64_t)(x / H) * batch_stride + (int64_t)(x % H) * HD; };
  float4 qf[NT8], qn[NT8];
  {
    const int64_t base = head_base(bh);
    ap_dma_image<HD, true>(lds, K + base, L, rs, wave, lane);
    const float* qrow = Q + base + (int64_t)qpos * row_stride + 4 * lh;
#pragma unroll
    for (int t = 0; t < NT8; ++t) qf[t] = *reinterpret_cast<const float4*>(qrow + 8 * t);
    ap_landed();
  }
  for (int it = 0; bh < BH; ++it, bh += gridDim.x) {
    const float* Ks = lds + (it & 1) * IMG;
    const int64_t base = head_base(bh);
    // ---- K of this head has landed -- every wave waited for its share before its last stores -- and every wave is
    //      done with the V of the previous head ----
    ap_barrier();
    ap_dma_image<HD, false>(Vs, V + base, L, rs, wave, lane);
    // ---- S^T tiles ---------------------------------------------------------------------
    f32x16 s[8];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
      if (kt < nk) {
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float* kb = Ks + kt * 32 * HD;
#pragma unroll
        for (int t = 0; t < NT8; ++t) {
          const float4 kf = *reinterpret_cast<const float4*>(kb + ko.at(t));
          s[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[t].x, t == 0 ? zero16 : s[kt], 0, 0, 0);
          s[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[t].y, s[kt], 0, 0, 0);
          s[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[t].z, s[kt], 0, 0, 0);
          s[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[t].w, s[kt], 0, 0, 0);
        }
      }
    }
    // ---- mask, softmax over keys (per lane = per query): p = exp2(s c1 - max(s) c1) ----
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
      if (kt < nk) {
        if (causal && kt == qtl) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (ap_krow(r, lh) > li) s[kt][r] = -INFINITY;
        }
#pragma unroll
        for (int r = 0; r < 16; r += 2) m = fmaxf(m, fmaxf(s[kt][r], s[kt][r + 1]));
      }
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float c2 = -m * c1;
    float lc = 0.f;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
      if (kt < nk) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s[kt][r] = __builtin_amdgcn_exp2f(fmaf(s[kt][r], c1, c2));
          lc += s[kt][r];
        }
      }
    }
    const float l = lc + __shfl_xor(lc, 32, 64);
    // ---- V of this head has landed; K of the next one and this wave's next Q rows go out ----
    ap_sync_all();
    const int nxt = bh + gridDim.x;
    if (nxt < BH) {
      const int64_t nb = head_base(nxt);
      ap_dma_image<HD, true>(lds + ((it + 1) & 1) * IMG, K + nb, L, rs, wave, lane);
      const float* qrow = Q + nb + (int64_t)qpos * row_stride + 4 * lh;
#pragma unroll
      for (int t = 0; t < NT8; ++t) qn[t] = *reinterpret_cast<const float4*>(qrow + 8 * t);
    }
    // ---- O^T = V^T P^T  (two 32-row tiles over the head dim; rows 48 .. 63 of the second are never stored) ----
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
      if (kt < nk) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float* vrow = Vs + (kt * 32 + ap_krow(r, lh)) * HD;
          const float a0 = vrow[li];
          const float a1 = vrow[32 + li];
          o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, s[kt][r], o0, 0, 0, 0);
          o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, s[kt][r], o1, 0, 0, 0);
        }
      }
    }
    ap_landed();                                    // next K, next Q rows (a whole P V phase old): before the stores
    if (active) {
      const float inv_l = 1.f / l;
      if (lh == 0) LSE[(int64_t)bh * L + qpos] = m * inv_sqrt + logf(l);
      float* orow = O + (int64_t)(bh / H) * o_batch_stride + (int64_t)(bh % H) * HD + (int64_t)qpos * o_row_stride + 4 * lh;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(orow + 8 * g) =
            make_float4(o0[4 * g] * inv_l, o0[4 * g + 1] * inv_l, o0[4 * g + 2] * inv_l, o0[4 * g + 3] * inv_l);
#pragma unroll
      for (int g = 0; g < (HD - 32) / 8; ++g)
        *reinterpret_cast<float4*>(orow + 32 + 8 * g) =
            make_float4(o1[4 * g] * inv_l, o1[4 * g + 1] * inv_l, o1[4 * g + 2] * inv_l, o1[4 * g + 3] * inv_l);
    }
#pragma unroll
    for (int t = 0; t < NT8; ++t) qf[t] = qn[t];
  }
}

static int ap_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

// 1 when the persistent kernels take the shape (rotation-free operands only: the caller checks its rope arguments)
int pdn_attention_p_supported(int L, int head_dim) {
  static const int off = getenv("PDN_ATT_NO_PERSIST") ? atoi(getenv("PDN_ATT_NO_PERSIST")) : 0;
  return !off && head_dim == 48 && L % 32 == 0 && L >= 32 && L <= AP_ROWS;
}

int pdn_attention_p_fwd(const float* q, const float* k, const float* v, float* o, float* lse, int B, int H, int L,
                        int head_dim, int64_t row_stride, int64_t batch_stride, int64_t o_row_stride,
                        int64_t o_batch_stride, int causal, void* stream) {
  constexpr int HD = 48;
  const size_t shm = (size_t)(3 * AP_ROWS * HD) * 4 + 64;
  static bool attr_set = false;
  if (!attr_set) {
    PDN_HIP(hipFuncSetAttribute((const void*)attention_p_fwd_kernel<48>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  PDN_CHECK_ARG((int64_t)L * row_stride < (1ll << 31), "pdn_attention_fwd_f32: a head's rows must lie within 2^31 floats");
  const int BH = B * H;
  const int grid = BH < ap_num_cus() ? BH : ap_num_cus();
  hipLaunchKernelGGL((attention_p_fwd_kernel<HD>), dim3(grid), dim3(512), shm, (hipStream_t)stream, q, k, v, o, lse, BH, H, L,
                     row_stride, batch_stride, o_row_stride, o_batch_stride, sqrtf((float)head_dim), causal);
  pdn_count(PDN_CNT_ATT_P_FWD);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

// ======================================================================================
// Backward (P recomputed from the saved log-sum-exp, csrc/attention.hip for the algebra):
//   attention_p_bwd_dq_kernel   a wave owns a query tile; K (rows AND columns) and V (rows) of the head in LDS.
//        phase 1, all key tiles:  S^T = K Q^T, dP^T = V dO^T  ->  dS^T = P^T o (dP^T - delta) kept in registers
//        phase 2:                 dQ^T += K^T dS^T
//     V is dead after phase 1, so V of the next head (and K of the next head, into the other K buffer) are in flight
//     during phase 2 -- two barriers per head, nothing exposed; also writes delta[q] = sum_d dO O.
//   attention_p_bwd_dkv_kernel  a wave owns a key tile; Q and dO (rows and columns both) + lse / delta of the head in
//     LDS; Q of the next head is prefetched into the other Q buffer during the head, dO is fetched at the head switch.
// q, k are taken as they come (already rotated or never to be rotated); with `RT` (the (cos, sin) tables of RoPE) the
// gradients dq, dk are rotated BACK as they are stored: they are the gradients of the un-rotated projections.
// ======================================================================================
struct ApRope {                       // (cos, sin) of the pairs a lane's output row pieces cover, at its row
  float2 c[6], s[6];
  __device__ __forceinline__ void load(const float* __restrict__ cs, const float* __restrict__ sn, int pos, int lh) {
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      c[t] = *reinterpret_cast<const float2*>(cs + pos * 24 + 4 * t + 2 * lh);
      s[t] = *reinterpret_cast<const float2*>(sn + pos * 24 + 4 * t + 2 * lh);
    }
  }
  // rotate back (angle -theta): (r, i) -> (r c + i s, -r s + i c)
  __device__ __forceinline__ float4 back(const float4& v, int t) const {
    return make_float4(v.x * c[t].x + v.y * s[t].x, v.y * c[t].x - v.x * s[t].x, v.z * c[t].y + v.w * s[t].y,
                       v.w * c[t].y - v.z * s[t].y);
  }
};

// X^T tiles t0 / t1 (lane = row, register r = head-dim index (r & 3) + 8 (r >> 2) + 4 h) -> the lane's own row:
// 16-byte pieces straight from the accumulators; `rowp` = first element of the lane's row + 4 h.
// ACC: the row already holds the contributions of other (query block, key block) pairs of a longer sequence
// (csrc/attention_blocks.hip): its six pieces are read first and the new values added.
template <bool ROT, bool ACC = false>
__device__ __forceinline__ void ap_store_rows(const f32x16& t0, const f32x16& t1, float* __restrict__ rowp, float scale,
                                              const ApRope& rr) {
  float4 old[6];
  if (ACC) {
#pragma unroll
    for (int g = 0; g < 6; ++g) old[g] = *reinterpret_cast<const float4*>(rowp + 8 * g);
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float4 v = make_float4(t0[4 * g] * scale, t0[4 * g + 1] * scale, t0[4 * g + 2] * scale, t0[4 * g + 3] * scale);
    if (ROT) v = rr.back(v, g);
    if (ACC) { v.x += old[g].x; v.y += old[g].y; v.z += old[g].z; v.w += old[g].w; }
    *reinterpret_cast<float4*>(rowp + 8 * g) = v;
  }
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    float4 v = make_float4(t1[4 * g] * scale, t1[4 * g + 1] * scale, t1[4 * g + 2] * scale, t1[4 * g + 3] * scale);
    if (ROT) v = rr.back(v, 4 + g);
    if (ACC) { v.x += old[4 + g].x; v.y += old[4 + g].y; v.z += old[4 + g].z; v.w += old[4 + g].w; }
    *reinterpret_cast<float4*>(rowp + 32 + 8 * g) = v;
  }
}

// dQ kernel: a wave owns a query tile; the (S^T, dP^T, dS^T, dQ^T) work of a key tile runs back to back -- no dS^T tiles
// kept in registers, no barriers inside a head.  K of the next head is prefetched into the other K buffer; V is fetched at
// the head switch (exposed, like dO in the dK/dV kernel).  (First form of this round: dS^T of four key tiles at a time in
// registers, V halves of the next head issued at two barriers inside the head -- nothing exposed, but three barriers per
// head whose causal skew -- SIMD s hosts tiles s and 7 - s: equal totals, unequal per phase -- cost more: 225 vs 197 us
// at 1536 heads, same box.)
template <int HD, bool ROT, bool ACC = false>
__global__ __launch_bounds__(512, 1) void attention_p_bwd_dq_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V, const float* __restrict__ O,
    const float* __restrict__ dO, const float* __restrict__ LSE, float* __restrict__ dQ, float* __restrict__ Delta, int BH,
    int H, int L, int64_t row_stride, int64_t batch_stride, int64_t o_row_stride, int64_t o_batch_stride, float sqrt_hd,
    int causal, const float* __restrict__ RC, const float* __restrict__ RS) {
  static_assert(HD == 48, "head dim 48");
  constexpr int NT8 = HD / 8, IMG = AP_ROWS * HD;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int v_at = 2 * IMG;                               // (beyond the 64 KB an LDS immediate reaches: kept opaque, see above)
  asm volatile("" : "+s"(v_at));
  float* Vs = lds + v_at;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int ntile = L / 32, rs = (int)row_stride;
  const int qtl = ap_tile_of_wave(wave);
  const bool active = qtl < ntile;
  const int qpos = (active ? qtl : 0) * 32 + li;
  const int nk = active ? (causal ? qtl + 1 : ntile) : 0;
  const float inv_sqrt = 1.f / sqrt_hd, c1 = inv_sqrt * 1.4426950408889634f;
  ApRowOff ko;
  ko.init(li, lh, HD);
  ApColOff<HD> co;
  co.init(li, lh);
  int bh = blockIdx.x;
  if (bh >= BH) return;
  auto head_base = [&](int x) { return (int64_t)(x / H) * batch_stride + (int64_t)(x % H) * HD; };
  auto head_obase = [&](int x) { return (int64_t)(x / H) * o_batch_stride + (int64_t)(x % H) * HD; };
  float4 qf[NT8], gf[NT8], of_[NT8];
  float lse_q;
  auto load_rows = [&](int x) {
    const int64_t base = head_base(x), ob = head_obase(x);
    const float* qrow = Q + base + (int64_t)qpos * row_stride + 4 * lh;
    const float* grow = dO + ob + (int64_t)qpos * o_row_stride + 4 * lh;
    const float* orow = O + ob + (int64_t)qpos * o_row_stride + 4 * lh;
#pragma unroll
    for (int t = 0; t < NT8; ++t) {
      qf[t] = *reinterpret_cast<const float4*>(qrow + 8 * t);
      gf[t] = *reinterpret_cast<const float4*>(grow + 8 * t);
      of_[t] = *reinterpret_cast<const float4*>(orow + 8 * t);
    }
    lse_q = LSE[(int64_t)x * L + qpos];
  };
  ap_dma_image<HD, true>(lds, K + head_base(bh), L, rs, wave, lane);
  load_rows(bh);
  ap_landed();
  for (int it = 0; bh < BH; ++it, bh += gridDim.x) {
    const float* Ks = lds + (it & 1) * IMG;
    // ---- every wave is done with the previous head: its V image is replaced ----
    if (it) ap_barrier();
    ap_dma_image<HD, true>(Vs, V + head_base(bh), L, rs, wave, lane);
    float dpart = 0.f;
#pragma unroll
    for (int t = 0; t < NT8; ++t)
      dpart += (of_[t].x * gf[t].x + of_[t].y * gf[t].y) + (of_[t].z * gf[t].z + of_[t].w * gf[t].w);
    const float delta_q = dpart + __shfl_xor(dpart, 32, 64);
    const float c2q = -lse_q * 1.4426950408889634f;
    ap_sync_all();
    const int nxt = bh + gridDim.x;
    if (nxt < BH) ap_dma_image<HD, true>(lds + ((it + 1) & 1) * IMG, K + head_base(nxt), L, rs, wave, lane);
    f32x16 dq0, dq1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dq0[r] = 0.f; dq1[r] = 0.f; }
    for (int kt = 0; kt < nk; ++kt) {
      f32x16 s, dp;
      const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const float* kb = Ks + kt * 32 * HD;
      const float* vb = Vs + kt * 32 * HD;
#pragma unroll
      for (int t = 0; t < NT8; ++t) {
        const float4 kf = *reinterpret_cast<const float4*>(kb + ko.at(t));
        const float4 vf = *reinterpret_cast<const float4*>(vb + ko.at(t));
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[t].x, t == 0 ? zero16 : s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.x, gf[t].x, t == 0 ? zero16 : dp, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[t].y, s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.y, gf[t].y, dp, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[t].z, s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.z, gf[t].z, dp, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[t].w, s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.w, gf[t].w, dp, 0, 0, 0);
      }
      if (causal && kt == qtl) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (ap_krow(r, lh) > li) s[r] = -INFINITY;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(s[r], c1, c2q));
        s[r] = p * (dp[r] - delta_q);                 // (the 1/sqrt(hd) of dS is applied once, when dQ is stored)
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* kr = Ks + (kt * 32 + (r & 3) + 8 * (r >> 2)) * HD;
        const float a0 = kr[co.first(r >> 2)];
        const float a1 = kr[co.second(r >> 2)];
        dq0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, s[r], dq0, 0, 0, 0);
        dq1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, s[r], dq1, 0, 0, 0);
      }
    }
    ApRope rr;
    if (ROT) rr.load(RC, RS, qpos, lh);
    if (nxt < BH) load_rows(nxt);                   // (qf / gf / of_ of this head are dead)
    ap_landed();                                    // next K (a head old), rope factors, next rows: before the stores
    if (active && lh == 0) Delta[(int64_t)bh * L + qpos] = delta_q;
    if (active) ap_store_rows<ROT, ACC>(dq0, dq1, dQ + head_base(bh) + (int64_t)qpos * row_stride + 4 * lh, inv_sqrt, rr);
  }
}

template <int HD, bool ROT, bool ACC = false>
__global__ __launch_bounds__(512, 1) void attention_p_bwd_dkv_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V, const float* __restrict__ dO,
    const float* __restrict__ LSE, const float* __restrict__ Delta, float* __restrict__ dK, float* __restrict__ dV, int BH,
    int H, int L, int64_t row_stride, int64_t batch_stride, int64_t o_row_stride, int64_t o_batch_stride, float sqrt_hd,
    int causal, const float* __restrict__ RC, const float* __restrict__ RS) {
  static_assert(HD == 48, "head dim 48");
  constexpr int UPR = HD / 4, NT8 = HD / 8, IMG = AP_ROWS * HD;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Gs = lds + 2 * IMG;                        // lds: Q (even) | Q (odd) | dO | lse, delta (two sets of 2 x 256)
  float* stat = lds + 3 * IMG;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int ntile = L / 32, rs = (int)row_stride, ors = (int)o_row_stride;
  const int ktl = ap_tile_of_wave(wave);
  const bool active = ktl < ntile;
  const int kpos = (active ? ktl : 0) * 32 + li;
  const float inv_sqrt = 1.f / sqrt_hd, c1 = inv_sqrt * 1.4426950408889634f;
  ApRowOff ko;
  ko.init(li, lh, HD);
  ApColOff<HD> co;
  co.init(li, lh);
  int bh = blockIdx.x;
  if (bh >= BH) return;
  auto head_base = [&](int x) { return (int64_t)(x / H) * batch_stride + (int64_t)(x % H) * HD; };
  auto head_obase = [&](int x) { return (int64_t)(x / H) * o_batch_stride + (int64_t)(x % H) * HD; };
  // lse (scaled by log2 e at use) and delta of a head: 2 x 256 floats = 128 units, two DMA instructions (waves 0 / 1)
  auto dma_stats = [&](float* dst, int x) {
    if (wave < 2) {
      const float* src = (wave == 0 ? LSE : Delta) + (int64_t)x * L;
      const int u = min(4 * lane, L - 4);
      ap_glds16(src + u, dst + 256 * wave);
    }
  };
  float4 kf[NT8], vf[NT8];
  {
    const int64_t base = head_base(bh);
    ap_dma_image<HD, true>(lds, Q + base, L, rs, wave, lane);
    const float* krow = K + base + (int64_t)kpos * row_stride + 4 * lh;
    const float* vrow = V + base + (int64_t)kpos * row_stride + 4 * lh;
#pragma unroll
    for (int t = 0; t < NT8; ++t) {
      kf[t] = *reinterpret_cast<const float4*>(krow + 8 * t);
      vf[t] = *reinterpret_cast<const float4*>(vrow + 8 * t);
    }
  }
  for (int it = 0; bh < BH; ++it, bh += gridDim.x) {
    const float* Qs = lds + (it & 1) * IMG;
    const float* lse_s = stat + (it & 1) * 512;
    const float* delta_s = lse_s + 256;
    // ---- every wave is done with the previous head: its dO image is replaced (exposed: see the header) ----
    if (it) ap_barrier();
    ap_dma_image<HD, true>(Gs, dO + head_obase(bh), L, ors, wave, lane);
    dma_stats(stat + (it & 1) * 512, bh);
    ap_sync_all();
    const int nxt = bh + gridDim.x;
    if (nxt < BH) ap_dma_image<HD, true>(lds + ((it + 1) & 1) * IMG, Q + head_base(nxt), L, rs, wave, lane);
    f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk0[r] = 0.f; dk1[r] = 0.f; dv0[r] = 0.f; dv1[r] = 0.f; }
    const int q_first = causal ? ktl : 0;
    if (active) {
      for (int qt = q_first; qt < ntile; ++qt) {
        f32x16 s, dp;
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float* qb = Qs + qt * 32 * HD;
        const float* gb = Gs + qt * 32 * HD;
#pragma unroll
        for (int t = 0; t < NT8; ++t) {
          const float4 q4 = *reinterpret_cast<const float4*>(qb + ko.at(t));
          const float4 g4 = *reinterpret_cast<const float4*>(gb + ko.at(t));
          s = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.x, kf[t].x, t == 0 ? zero16 : s, 0, 0, 0);    // S[q][key]
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(g4.x, vf[t].x, t == 0 ? zero16 : dp, 0, 0, 0);  // dP[q][key]
          s = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.y, kf[t].y, s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(g4.y, vf[t].y, dp, 0, 0, 0);
          s = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.z, kf[t].z, s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(g4.z, vf[t].z, dp, 0, 0, 0);
          s = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.w, kf[t].w, s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(g4.w, vf[t].w, dp, 0, 0, 0);
        }
        if (causal && qt == ktl) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (li > ap_krow(r, lh)) s[r] = -INFINITY;
        }
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {                      // registers 4 g4 .. 4 g4 + 3 are four consecutive queries
          const int q0 = qt * 32 + 8 * g4 + 4 * lh;
          const float4 ls = *reinterpret_cast<const float4*>(lse_s + q0);
          const float4 dl = *reinterpret_cast<const float4*>(delta_s + q0);
          const float lq[4] = {ls.x, ls.y, ls.z, ls.w}, dq4[4] = {dl.x, dl.y, dl.z, dl.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * g4 + e;
            const float p = __builtin_amdgcn_exp2f(fmaf(s[r], c1, -1.4426950408889634f * lq[e]));
            s[r] = p;                                        // P[q][key]
            dp[r] = p * (dp[r] - dq4[e]);                    // dS[q][key] * sqrt(hd): the scale is applied when dK is stored
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ro = (qt * 32 + (r & 3) + 8 * (r >> 2)) * HD;
          const float g0 = Gs[ro + co.first(r >> 2)], q0 = Qs[ro + co.first(r >> 2)];
          const float g1 = Gs[ro + co.second(r >> 2)], q1 = Qs[ro + co.second(r >> 2)];
          dv0 = __builtin_amdgcn_mfma_f32_32x32x2f32(g0, s[r], dv0, 0, 0, 0);     // dV^T += dO^T P
          dk0 = __builtin_amdgcn_mfma_f32_32x32x2f32(q0, dp[r], dk0, 0, 0, 0);    // dK^T += Q^T dS
          dv1 = __builtin_amdgcn_mfma_f32_32x32x2f32(g1, s[r], dv1, 0, 0, 0);
          dk1 = __builtin_amdgcn_mfma_f32_32x32x2f32(q1, dp[r], dk1, 0, 0, 0);
        }
      }
    }
    // the next head's Q image (a whole head old) has landed: every wave says so BEFORE its stores; the wave's next K / V
    // rows then travel behind the stores and the barrier (the next head waits for them together with its dO image)
    ap_landed();
    ApRope rr;
    if (ROT) rr.load(RC, RS, kpos, lh);
    float4 kn[NT8], vn[NT8];
    if (nxt < BH) {
      const int64_t nb = head_base(nxt);
      const float* krow = K + nb + (int64_t)kpos * row_stride + 4 * lh;
      const float* vrow = V + nb + (int64_t)kpos * row_stride + 4 * lh;
#pragma unroll
      for (int t = 0; t < NT8; ++t) {
        kn[t] = *reinterpret_cast<const float4*>(krow + 8 * t);
        vn[t] = *reinterpret_cast<const float4*>(vrow + 8 * t);
      }
    }
    if (active) {
      const int64_t base = head_base(bh);
      ap_store_rows<ROT, ACC>(dk0, dk1, dK + base + (int64_t)kpos * row_stride + 4 * lh, inv_sqrt, rr);
      ap_store_rows<false, ACC>(dv0, dv1, dV + base + (int64_t)kpos * row_stride + 4 * lh, 1.f, rr);
    }
#pragma unroll
    for (int t = 0; t < NT8; ++t) { kf[t] = kn[t]; vf[t] = vn[t]; }
  }
}

int64_t pdn_attention_p_bwd_workspace_bytes(int B, int H, int L) { return (int64_t)B * H * L * 4; }

// (rope tables of the QUERY rows for dq and of the KEY rows for dk: the same pointers for a whole sequence, different
//  row offsets when q and k are different 256-row blocks of one -- csrc/attention_blocks.hip)
int pdn_attention_p_bwd_tables(const float* q, const float* k, const float* v, const float* o, const float* d_o, const float* lse,
                               float* dq, float* dk, float* dv, int B, int H, int L, int head_dim, int64_t row_stride,
                               int64_t batch_stride, int64_t o_row_stride, int64_t o_batch_stride, int causal,
                               const float* rope_cos_q, const float* rope_sin_q, const float* rope_cos_k, const float* rope_sin_k,
                               float* delta, void* stream, int acc_q, int acc_kv) {
  const float* rope_cos = rope_cos_q;
  constexpr int HD = 48;
  const size_t shm_dq = (size_t)(3 * AP_ROWS * HD) * 4 + 64, shm_dkv = (size_t)(3 * AP_ROWS * HD + 1024) * 4 + 64;
  static bool attr_set = false;
  if (!attr_set) {
#define AP_ATTR(K_) PDN_HIP(hipFuncSetAttribute((const void*)K_, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
    AP_ATTR((attention_p_bwd_dq_kernel<48, true>)); AP_ATTR((attention_p_bwd_dq_kernel<48, false>));
    AP_ATTR((attention_p_bwd_dkv_kernel<48, true>)); AP_ATTR((attention_p_bwd_dkv_kernel<48, false>));
    AP_ATTR((attention_p_bwd_dq_kernel<48, true, true>)); AP_ATTR((attention_p_bwd_dq_kernel<48, false, true>));
    AP_ATTR((attention_p_bwd_dkv_kernel<48, true, true>)); AP_ATTR((attention_p_bwd_dkv_kernel<48, false, true>));
#undef AP_ATTR
    attr_set = true;
  }
  PDN_CHECK_ARG((int64_t)L * row_stride < (1ll << 31) && (int64_t)L * o_row_stride < (1ll << 31),
                "pdn_attention_bwd_f32: a head's rows must lie within 2^31 floats");
  const int BH = B * H;
  const int grid = BH < ap_num_cus() ? BH : ap_num_cus();
  const float sq = sqrtf((float)head_dim);
  hipStream_t st = (hipStream_t)stream;
#define AP_DQ(R_, A_)                                                                                                       \
  hipLaunchKernelGGL((attention_p_bwd_dq_kernel<HD, R_, A_>), dim3(grid), dim3(512), shm_dq, st, q, k, v, o, d_o, lse, dq, delta, \
                     BH, H, L, row_stride, batch_stride, o_row_stride, o_batch_stride, sq, causal, rope_cos_q, rope_sin_q)
#define AP_DKV(R_, A_)                                                                                                      \
  hipLaunchKernelGGL((attention_p_bwd_dkv_kernel<HD, R_, A_>), dim3(grid), dim3(512), shm_dkv, st, q, k, v, d_o, lse, delta, dk,  \
                     dv, BH, H, L, row_stride, batch_stride, o_row_stride, o_batch_stride, sq, causal, rope_cos_k, rope_sin_k)
  if (rope_cos) { if (acc_q) AP_DQ(true, true); else AP_DQ(true, false); }
  else { if (acc_q) AP_DQ(false, true); else AP_DQ(false, false); }
  PDN_LAUNCH_CHECK();
  if (rope_cos) { if (acc_kv) AP_DKV(true, true); else AP_DKV(true, false); }
  else { if (acc_kv) AP_DKV(false, true); else AP_DKV(false, false); }
  PDN_LAUNCH_CHECK();
#undef AP_DQ
#undef AP_DKV
  pdn_count(PDN_CNT_ATT_P_BWD);
  return PDN_OK;
}

int pdn_attention_p_bwd(const float* q, const float* k, const float* v, const float* o, const float* d_o, const float* lse,
                        float* dq, float* dk, float* dv, int B, int H, int L, int head_dim, int64_t row_stride,
                        int64_t batch_stride, int64_t o_row_stride, int64_t o_batch_stride, int causal,
                        const float* rope_cos, const float* rope_sin, float* delta, void* stream) {
  return pdn_attention_p_bwd_tables(q, k, v, o, d_o, lse, dq, dk, dv, B, H, L, head_dim, row_stride, batch_stride, o_row_stride,
                                    o_batch_stride, causal, rope_cos, rope_sin, rope_cos, rope_sin, delta, stream, 0, 0);
}
