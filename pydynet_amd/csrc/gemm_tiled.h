// The tiled fp32-MFMA GEMM kernel of csrc/gemm.hip (device side): GemmParams, the LDS tile loaders, gemm_f32_mfma_kernel with
// its epilogues (bias / residual / column sums / relu and gradient bit masks / SwiGLU forward and backward) and the split-K
// reduce.  Included by gemm.hip only (one translation unit: the split is for the reader, HISTORY.md 4.1 / 4.14 describe it).
#pragma once
#include "common.h"
#include <type_traits>

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a compile-time constant
template <int N, class F>
__device__ __forceinline__ void gemm_static_for(F&& f) {
  if constexpr (N > 0) {
    gemm_static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GEMM_PAD 4

struct GemmParams {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  const float* residual;   // optional, same indexing as C: C += residual
  float* colsum;           // optional, length N: column sums of B (bias gradient of x^T @ g)
  int colsum_acc;
  // tiled kernel only, one batch, no k-split, N a multiple of 32: one bit per output element, bit c of word
  // [row * (N / 32) + col / 32] <-> column 32 * (col / 32) + c
  uint32_t* relu_mask;       // store max(0, result) and set the bit where result >= 0 (relu gradient passes, functional.py:31-32)
  const uint32_t* grad_mask; // store result where the bit is set, 0 elsewhere
  float* mask_colsum;        // with grad_mask: (ceil(M / 32) x N) partial column sums of what is stored, one row per 32-row band
  // SwiGLU in the store (SWI instantiations, pdn_gateup_swiglu_tiled_fwd_f32 / pdn_swiglu_bwd_tiled_f32):
  //  1: the product's columns alternate 32 gate / 32 up columns (weights packed that way); C = [gate | up] (M x 2 F, ldc),
  //     swi_h (M x F, swi_ldh) = silu(gate) * up
  //  2: the product is dh (M x F); C = d[gate | up] (M x 2 F, ldc) from dh and the saved swi_gu (M x 2 F, ldc)
  float* swi_h;
  const float* swi_gu;
  int64_t swi_ldh;
  int swi_F;
  float* ws;
  int M, N, K;
  int64_t a_rs, a_cs, b_rs, b_cs, ldc;
  int nb2;
  int64_t a_bs1, a_bs2, b_bs1, b_bs2, c_bs1, c_bs2;
  float alpha, beta;
  int splits, k_per_split;
  int tiles_m, tiles_n;
};

// relu epilogue helpers: max(0, v) that keeps a NaN (numpy.maximum propagates it) and turns -0 into +0; bit j of an 8-bit
// value moved to bit 4j
__device__ __forceinline__ float relu_keep_nan(float v) { return v < 0.f ? 0.f : v + 0.f; }
__device__ __forceinline__ float gemm_silu(float g) { return g / (1.f + expf(-g)); }          // (as csrc/fused.hip: silu_f / dsilu_f)
__device__ __forceinline__ float gemm_dsilu(float g) {
  const float sg = 1.f / (1.f + expf(-g));
  return sg * (1.f + g * (1.f - sg));
}
__device__ __forceinline__ uint32_t spread_bits8(uint32_t x) {
  x = (x | (x << 12)) & 0x000F000Fu;
  x = (x | (x << 6)) & 0x03030303u;
  return (x | (x << 3)) & 0x11111111u;
}

// ---- global -> register -> LDS staging -------------------------------------------------------
// Tile of an operand: MN rows (m or n index) x BK contraction columns.
// KIN  : LDS image [MN][BK+PAD]   (contraction contiguous), unit = float4 along k
// !KIN : LDS image [BK][MN+PAD]   (m/n contiguous),         unit = float4 along m/n
template <int MN, int BK, bool KIN, bool VEC, int NT>
struct TileLoader {
  static constexpr int UNITS = MN * BK / 4;
  static constexpr int NP = (UNITS + NT - 1) / NT;
  static constexpr int LD = KIN ? (BK + GEMM_PAD) : (MN + GEMM_PAD);
  static constexpr int SIZE = KIN ? MN * LD : BK * LD;

  __device__ __forceinline__ static void load(float4 (&r)[NP], const float* __restrict__ base,
                                              int64_t s_mn, int64_t s_k, int mn0, int k0,
                                              int mn_end, int k_end, int tid) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int u = tid + p * NT;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (UNITS % NT == 0 || u < UNITS) {
        if (KIN) {
          const int row = u / (BK / 4), c4 = u % (BK / 4);
          const int mn = mn0 + row, k = k0 + 4 * c4;
          if (VEC) {
            if (mn < mn_end && k < k_end)
              v = *reinterpret_cast<const float4*>(base + (int64_t)mn * s_mn + k);
          } else if (mn < mn_end) {
            const float* q = base + (int64_t)mn * s_mn + (int64_t)k * s_k;
            if (k + 0 < k_end) v.x = q[0];
            if (k + 1 < k_end) v.y = q[s_k];
            if (k + 2 < k_end) v.z = q[2 * s_k];
            if (k + 3 < k_end) v.w = q[3 * s_k];
          }
        } else {
          const int kr = u / (MN / 4), c4 = u % (MN / 4);
          const int k = k0 + kr, mn = mn0 + 4 * c4;
          if (VEC) {
            if (k < k_end && mn < mn_end)
              v = *reinterpret_cast<const float4*>(base + (int64_t)k * s_k + mn);
          } else if (k < k_end) {
            const float* q = base + (int64_t)k * s_k + (int64_t)mn * s_mn;
            if (mn + 0 < mn_end) v.x = q[0];
            if (mn + 1 < mn_end) v.y = q[s_mn];
            if (mn + 2 < mn_end) v.z = q[2 * s_mn];
            if (mn + 3 < mn_end) v.w = q[3 * s_mn];
          }
        }
      }
      r[p] = v;
    }
  }

  // Interior tiles (VEC layout): per-thread element offsets are computed ONCE, relative to the
  // first contraction index; a k-tile then costs one 64-bit add and one 16-byte load per piece,
  // with no bounds tests and no zero-fill moves in the loop.
  __device__ __forceinline__ static void init_offsets(int64_t (&off)[NP], int64_t s_mn, int64_t s_k,
                                                      int mn0, int tid) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int u = (UNITS % NT == 0) ? tid + p * NT : min(tid + p * NT, UNITS - 1);
      if (KIN) {
        const int row = u / (BK / 4), c4 = u % (BK / 4);
        off[p] = (int64_t)(mn0 + row) * s_mn + 4 * c4;
      } else {
        const int kr = u / (MN / 4), c4 = u % (MN / 4);
        off[p] = (int64_t)kr * s_k + mn0 + 4 * c4;
      }
    }
  }
  __device__ __forceinline__ static void load_fast(float4 (&r)[NP], const float* __restrict__ base_k,
                                                   const int64_t (&off)[NP]) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      // component-wise copy: a whole-float4 store into the by-reference array defeats SROA on
      // hipcc 7.2 and sends the staging registers to scratch
      const float4 v = *reinterpret_cast<const float4*>(base_k + off[p]);
      r[p].x = v.x; r[p].y = v.y; r[p].z = v.z; r[p].w = v.w;
    }
  }

  __device__ __forceinline__ static void store(const float4 (&r)[NP], float* lds, int tid) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int u = tid + p * NT;
      if (UNITS % NT == 0 || u < UNITS) {
        int off;
        if (KIN) {
          const int row = u / (BK / 4), c4 = u % (BK / 4);
          off = row * LD + 4 * c4;
        } else {
          const int kr = u / (MN / 4), c4 = u % (MN / 4);
          off = kr * LD + 4 * c4;
        }
        *reinterpret_cast<float4*>(lds + off) = r[p];
      }
    }
  }

  // Fragment for MFMA steps j=0..3 of k-group t: lane (i = lane&31, h = lane>>5) needs
  // operand(row0 + i, k = 8t + 4h + j).
  __device__ __forceinline__ static void frag(float (&f)[4], const float* lds, int row0, int t,
                                              int li, int lh) {
    if (KIN) {
      const float4 v =
          *reinterpret_cast<const float4*>(lds + (row0 + li) * LD + 8 * t + 4 * lh);
      f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    } else {
      const float* q = lds + (8 * t + 4 * lh) * LD + row0 + li;
      f[0] = q[0]; f[1] = q[LD]; f[2] = q[2 * LD]; f[3] = q[3 * LD];
    }
  }
};

// One k-loop over [k_begin, k_end) for the tile at (m0, n0).  INTERIOR tiles (fully inside M x N
// with whole k-tiles) take loads with no bounds tests so the staging code is branch-free.
template <int WAVES_M, int WAVES_N, int WM, int WN, int BK, bool A_KIN, bool B_KIN, bool VEC, bool INTERIOR, bool COLSUM>
__device__ __forceinline__ void gemm_mainloop(const GemmParams& p, const float* __restrict__ A,
                                              const float* __restrict__ B, float* smem, int m0, int n0,
                                              int k_begin, int k_end, f32x16 (&acc)[WM][WN],
                                              bool do_colsum, float& csum) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32;
  using LA = TileLoader<BM, BK, A_KIN, VEC, NT>;
  using LB = TileLoader<BN, BK, B_KIN, VEC, NT>;
  constexpr int STAGE = LA::SIZE + LB::SIZE;
  constexpr int NG = BK / 8;                      // k-groups (4 MFMA steps each) per k-tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
  const int li = lane & 31, lh = lane >> 5;
  // interior tiles: lift the bounds so every `<` test folds to true
  const int m_end = INTERIOR ? 0x7fffffff : p.M, n_end = INTERIOR ? 0x7fffffff : p.N;
  const int kk_end = INTERIOR ? 0x7fffffff : k_end;

  // Two register sets: while tile t is multiplied, tile t+1 waits in one set (it is written to
  // LDS in the middle of tile t's MFMA stream) and tile t+2 is in flight into the other.  The
  // global-load window is therefore ~1.5 tiles of MFMA time (3-6k cycles), enough to cover HBM
  // latency when an operand is streamed with no reuse (weight gradients, K = tokens).  The sets
  // are named, not indexed, so they stay in registers (the loop is unrolled by two).
  float4 ra0[LA::NP], rb0[LB::NP], ra1[LA::NP], rb1[LB::NP];
  int64_t offa[LA::NP], offb[LB::NP];
  if (INTERIOR) {
    LA::init_offsets(offa, p.a_rs, p.a_cs, m0, tid);
    LB::init_offsets(offb, p.b_cs, p.b_rs, n0, tid);
  }
  // contraction stride of each operand in the staged layout (1 for K-contiguous operands)
  const int64_t ka = p.a_cs, kb = p.b_rs;
  // (tried in round 5: interior tiles with a trailing partial k-tile -- K = 500, 784 with BK 32 -- taking the guarded loads
  //  for that last k-tile only.  The run-time test puts BOTH load paths into the interior instantiation: the dim-512 Llama
  //  step lost 3 % (74.4 -> 72.0 % at model level) and the K = 500 products it was meant for gained nothing; removed)
#define GEMM_LOAD(RA, RB, K0)                                                        \
  if (INTERIOR) {                                                                    \
    LA::load_fast(RA, A + (int64_t)(K0) * ka, offa);                                 \
    LB::load_fast(RB, B + (int64_t)(K0) * kb, offb);                                 \
  } else {                                                                           \
    LA::load(RA, A, p.a_rs, p.a_cs, m0, (K0), m_end, kk_end, tid);                   \
    LB::load(RB, B, p.b_cs, p.b_rs, n0, (K0), n_end, kk_end, tid);                   \
  }
  const int ntile = (k_end - k_begin + BK - 1) / BK;
  if (ntile > 0) {
    GEMM_LOAD(ra0, rb0, k_begin)
    if (ntile > 1) {
      GEMM_LOAD(ra1, rb1, k_begin + BK)
    }
    LA::store(ra0, smem, tid);
    LB::store(rb0, smem + LA::SIZE, tid);
  }
  __syncthreads();

  // body(t, free set, waiting set): `waiting` holds tile t+1, `free` receives tile t+2
#define GEMM_TILE_BODY(T, FA, FB, WA, WB)                                                          \
  {                                                                                                \
    const int t_ = (T);                                                                            \
    const float* As = smem + (t_ & 1) * STAGE;                                                     \
    const float* Bs = As + LA::SIZE;                                                               \
    float* An = smem + ((t_ + 1) & 1) * STAGE;                                                     \
    if (t_ + 2 < ntile) {                                                                          \
      const int k0 = k_begin + (t_ + 2) * BK;                                                      \
      GEMM_LOAD(FA, FB, k0)                                                                        \
    }                                                                                              \
    if (COLSUM && do_colsum) {                                                                     \
      constexpr int CG = NT / BN > 0 ? NT / BN : 1;                                                \
      if (tid < CG * BN) {                                                                         \
        const int col = tid % BN, grp = tid / BN;                                                  \
        _Pragma("unroll") for (int k = grp; k < BK; k += CG) csum += Bs[k * LB::LD + col];         \
      }                                                                                            \
    }                                                                                              \
    _Pragma("unroll") for (int g = 0; g < NG; ++g) {                                               \
      float a[WM][4], b[WN][4];                                                                    \
      _Pragma("unroll") for (int i = 0; i < WM; ++i)                                               \
          LA::frag(a[i], As, (wave_m * WM + i) * 32, g, li, lh);                                   \
      _Pragma("unroll") for (int j = 0; j < WN; ++j)                                               \
          LB::frag(b[j], Bs, (wave_n * WN + j) * 32, g, li, lh);                                   \
      _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                \
        _Pragma("unroll") for (int i = 0; i < WM; ++i)                                             \
          _Pragma("unroll") for (int j = 0; j < WN; ++j)                                           \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0); \
      if (g == (NG - 1) / 2 && t_ + 1 < ntile) {                                                   \
        LA::store(WA, An, tid);                                                                    \
        LB::store(WB, An + LA::SIZE, tid);                                                         \
      }                                                                                            \
    }                                                                                              \
    __syncthreads();                                                                               \
  }

  // column sums of the staged B tile ([BK][BN+pad], zero-filled outside K x N) are taken by the
  // tile_m == 0 blocks only, so each column is counted once (COLSUM instantiation)
  int t = 0;
  for (; t + 1 < ntile; t += 2) {
    GEMM_TILE_BODY(t, ra0, rb0, ra1, rb1)       // tile t+1 waits in set 1, t+2 loads into set 0
    GEMM_TILE_BODY(t + 1, ra1, rb1, ra0, rb0)   // tile t+2 waits in set 0, t+3 loads into set 1
  }
  if (t < ntile) GEMM_TILE_BODY(t, ra0, rb0, ra1, rb1)
#undef GEMM_TILE_BODY
#undef GEMM_LOAD
}

// MASKS: the instantiations behind pdn_linear_relu_fwd_f32 / pdn_linear_dx_masked_f32 (GemmParams::relu_mask / grad_mask)
// SWI: SwiGLU in the store (GemmParams::swi_*), whole row tiles only (the launcher guarantees M % BM == 0 and aligned operands)
template <int WAVES_M, int WAVES_N, int WM, int WN, int BK, bool A_KIN, bool B_KIN, bool VEC, bool COLSUM, bool MASKS = false, int SWI = 0>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, (BK == 16 && WM * WN <= 3) ? 3 : 2) void gemm_f32_mfma_kernel(GemmParams p) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32;
  using LA = TileLoader<BM, BK, A_KIN, VEC, NT>;
  using LB = TileLoader<BN, BK, B_KIN, VEC, NT>;
  constexpr int STAGE = LA::SIZE + LB::SIZE;
  // epilogue staging: each wave parks ONE 32-row band of its accumulator block at a time
  constexpr int EW = WN * 32 + 4;                       // padded row length (floats)
  constexpr int EPI = WAVES_M * WAVES_N * 32 * EW;
  constexpr int SMEM = (2 * STAGE > EPI) ? 2 * STAGE : EPI;
  __shared__ __attribute__((aligned(16))) float smem[SMEM];

  // ---- which tile / batch / k-split ---------------------------------------------------
  const int nwg = p.tiles_m * p.tiles_n;
  int L;
  {
    const int bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int tile_m, tile_n;
  {
    constexpr int GROUP = 8;
    const int width = GROUP * p.tiles_n;
    const int g = L / width, first = g * GROUP;
    const int gsz = min(p.tiles_m - first, GROUP);
    const int w = L - g * width;
    tile_m = first + w % gsz;
    tile_n = w / gsz;
  }
  const int z = blockIdx.y;
  const int split = z % p.splits, batch = z / p.splits;
  const int b1 = batch / p.nb2, b2 = batch % p.nb2;
  const float* __restrict__ A = p.A + b1 * p.a_bs1 + b2 * p.a_bs2;
  const float* __restrict__ B = p.B + b1 * p.b_bs1 + b2 * p.b_bs2;

  const int k_begin = split * p.k_per_split;
  const int k_end = min(p.K, k_begin + p.k_per_split);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const bool interior = VEC && (m0 + BM <= p.M) && (n0 + BN <= p.N) && ((k_end - k_begin) % BK == 0);

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const bool do_colsum = COLSUM && p.colsum != nullptr && tile_m == 0;
  float csum = 0.f;
  if (interior)
    gemm_mainloop<WAVES_M, WAVES_N, WM, WN, BK, A_KIN, B_KIN, VEC, true, COLSUM>(p, A, B, smem, m0, n0, k_begin, k_end, acc, do_colsum, csum);
  else
    gemm_mainloop<WAVES_M, WAVES_N, WM, WN, BK, A_KIN, B_KIN, VEC, false, COLSUM>(p, A, B, smem, m0, n0, k_begin, k_end, acc, do_colsum, csum);
  if (COLSUM && do_colsum) {
    constexpr int CG = NT / BN > 0 ? NT / BN : 1;
    if (threadIdx.x < CG * BN) smem[threadIdx.x] = csum;
    __syncthreads();
    if (threadIdx.x < BN && n0 + (int)threadIdx.x < p.N) {
      float s = smem[threadIdx.x];
#pragma unroll
      for (int g = 1; g < CG; ++g) s += smem[g * BN + threadIdx.x];
      float* dst = p.colsum + n0 + threadIdx.x;
      *dst = p.colsum_acc ? *dst + s : s;
    }
    __syncthreads();
  }

  // ---- epilogue -----------------------------------------------------------------------
  // 32x32 accumulator map: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
  const int li = lane & 31, lh = lane >> 5;
  const bool partial = p.splits > 1;
  float* __restrict__ C =
      partial ? p.ws + ((int64_t)batch * p.splits + split) * (int64_t)p.M * p.N
              : p.C + b1 * p.c_bs1 + b2 * p.c_bs2;
  const int64_t ldc = partial ? p.N : p.ldc;
  const float* bias = partial ? nullptr : p.bias;
  const float* __restrict__ R = (partial || !p.residual) ? nullptr : p.residual + b1 * p.c_bs1 + b2 * p.c_bs2;
  const float beta = partial ? 0.f : p.beta;

  const bool wide = SWI != 0 || ((m0 + BM <= p.M) && (n0 + BN <= p.N) && ((ldc & 3) == 0) &&
                                 (((uintptr_t)C & 15) == 0) && (!bias || ((uintptr_t)bias & 15) == 0) &&
                                 (!R || ((uintptr_t)R & 15) == 0));
  if (wide) {
    // The main loop's last barrier has retired every read of the staging buffers, and each wave
    // only touches its own band region, so wave-level ordering is all that is needed from here.
    float* ws = smem + wave * (32 * EW);
    constexpr int C4 = WN * 8;                   // float4 per row of the wave block
    constexpr int UNITS = 32 * C4;
    const int col0 = n0 + wave_n * WN * 32;
    constexpr int NU = (UNITS + 63) / 64;
    // MASKS: the gradient-bit words of every band are requested before the first store of this epilogue (a load inside
    // the store loop waits behind the stores before it: +33 us on a 1.09 ms product), and the words a relu store
    // produces leave after the band's loop (one predicated region per band instead of one per 16-byte store)
    uint32_t gm[MASKS ? WM : 1][NU];
    if (MASKS && p.grad_mask) {
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int ui = 0; ui < NU; ++ui) {
          const int u = 64 * ui + lane, r = u / C4, c4 = u % C4;
          gm[i][ui] = p.grad_mask[(int64_t)(m0 + (wave_m * WM + i) * 32 + r) * (p.N >> 5) + (col0 >> 5) + (c4 >> 3)] >> (4 * (c4 & 7));
        }
    }
    // (a compile-time band index: with the mask epilogues on the 128-accumulator tiles `#pragma unroll` gave up on this
    //  loop and the accumulators were indexed at run time = kept in scratch: 576 B per lane, the step ran 2x slower)
    gemm_static_for<WM>([&](auto i_c) {
      constexpr int i = decltype(i_c)::value;
      uint32_t wd[NU];
      float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);   // MASKS + mask_colsum: this lane's columns summed over the band's rows
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          ws[((r & 3) + 8 * (r >> 2) + 4 * lh) * EW + j * 32 + li] = p.alpha * acc[i][j][r];
      __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0): own LDS writes landed
      __builtin_amdgcn_wave_barrier();
      const int row0 = m0 + (wave_m * WM + i) * 32;
      if constexpr (SWI == 1) {
        // the wave's columns are 32 gate | 32 up | 32 gate | ...: a lane takes a gate float4 and the up float4 32 columns on
        static_assert(SWI != 1 || WN % 2 == 0, "gate / up column groups come in pairs");
        constexpr int PC4 = C4 / 2, PUNITS = 32 * PC4;
#pragma unroll
        for (int u0 = 0; u0 < PUNITS; u0 += 64) {
          const int u = u0 + lane;
          if (PUNITS % 64 == 0 || u < PUNITS) {
            const int r = u / PC4, pc = u % PC4, grp = pc >> 3, w4 = pc & 7;
            const float4 g = *reinterpret_cast<const float4*>(ws + r * EW + 64 * grp + 4 * w4);
            const float4 up = *reinterpret_cast<const float4*>(ws + r * EW + 64 * grp + 32 + 4 * w4);
            const int gcol = (col0 >> 1) + 32 * grp + 4 * w4;
            if (gcol < p.swi_F) {
              float* grow = C + (int64_t)(row0 + r) * ldc + gcol;
              *reinterpret_cast<float4*>(grow) = g;
              *reinterpret_cast<float4*>(grow + p.swi_F) = up;
              *reinterpret_cast<float4*>(p.swi_h + (int64_t)(row0 + r) * p.swi_ldh + gcol) =
                  make_float4(gemm_silu(g.x) * up.x, gemm_silu(g.y) * up.y, gemm_silu(g.z) * up.z, gemm_silu(g.w) * up.w);
            }
          }
        }
      } else if constexpr (SWI == 2) {
#pragma unroll
        for (int u0 = 0; u0 < UNITS; u0 += 64) {
          const int u = u0 + lane;
          if (UNITS % 64 == 0 || u < UNITS) {
            const int r = u / C4, c4 = u % C4;
            const int col = col0 + 4 * c4;
            if (col < p.N) {
              const float4 d = *reinterpret_cast<const float4*>(ws + r * EW + 4 * c4);
              const float* srow = p.swi_gu + (int64_t)(row0 + r) * ldc + col;
              const float4 a = *reinterpret_cast<const float4*>(srow), b = *reinterpret_cast<const float4*>(srow + p.swi_F);
              float* drow = C + (int64_t)(row0 + r) * ldc + col;
              *reinterpret_cast<float4*>(drow) = make_float4(d.x * b.x * gemm_dsilu(a.x), d.y * b.y * gemm_dsilu(a.y),
                                                             d.z * b.z * gemm_dsilu(a.z), d.w * b.w * gemm_dsilu(a.w));
              *reinterpret_cast<float4*>(drow + p.swi_F) = make_float4(d.x * gemm_silu(a.x), d.y * gemm_silu(a.y),
                                                                       d.z * gemm_silu(a.z), d.w * gemm_silu(a.w));
            }
          }
        }
      } else {
#pragma unroll
      for (int u0 = 0; u0 < UNITS; u0 += 64) {
        const int u = u0 + lane;
        if (UNITS % 64 == 0 || u < UNITS) {
          const int r = u / C4, c4 = u % C4;
          float4 v = *reinterpret_cast<const float4*>(ws + r * EW + 4 * c4);
          float* dst = C + (int64_t)(row0 + r) * ldc + col0 + 4 * c4;
          if (bias) {
            const float4 bv = *reinterpret_cast<const float4*>(bias + col0 + 4 * c4);
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
          }
          if (R) {
            const float4 rv = *reinterpret_cast<const float4*>(R + (int64_t)(row0 + r) * ldc + col0 + 4 * c4);
            v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
          }
          if (beta != 0.f) {
            const float4 o = *reinterpret_cast<const float4*>(dst);
            v.x += beta * o.x; v.y += beta * o.y; v.z += beta * o.z; v.w += beta * o.w;
          }
          if (MASKS && p.grad_mask) {             // (8 lanes = one 32-column word of this row)
            const uint32_t w = gm[MASKS ? i : 0][u0 / 64];
            v.x = (w & 1u) ? v.x : 0.f; v.y = (w & 2u) ? v.y : 0.f; v.z = (w & 4u) ? v.z : 0.f; v.w = (w & 8u) ? v.w : 0.f;
            cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
          }
          if (MASKS && p.relu_mask) {
            // bit j of a lane group's byte of each ballot = lane 8g + j = columns 4j .. 4j + 3 of the word
            const int sh = lane & 56;
            const uint32_t word = spread_bits8((uint32_t)(__ballot(v.x >= 0.f) >> sh) & 0xffu) |
                                  spread_bits8((uint32_t)(__ballot(v.y >= 0.f) >> sh) & 0xffu) << 1 |
                                  spread_bits8((uint32_t)(__ballot(v.z >= 0.f) >> sh) & 0xffu) << 2 |
                                  spread_bits8((uint32_t)(__ballot(v.w >= 0.f) >> sh) & 0xffu) << 3;
            v.x = relu_keep_nan(v.x); v.y = relu_keep_nan(v.y); v.z = relu_keep_nan(v.z); v.w = relu_keep_nan(v.w);
            wd[u0 / 64] = word;
          }
          *reinterpret_cast<float4*>(dst) = v;
        }
      }
      }
      if (MASKS && 64 % C4 == 0 && p.mask_colsum) {
        // a lane keeps its float4 column over the band (64 is a multiple of the C4 lanes of a row): combine the 64 / C4
        // lanes that share it, one partial row per band
#pragma unroll
        for (int sft = C4; sft < 64; sft <<= 1) {
          cs.x += __shfl_xor(cs.x, sft); cs.y += __shfl_xor(cs.y, sft); cs.z += __shfl_xor(cs.z, sft); cs.w += __shfl_xor(cs.w, sft);
        }
        if (lane < C4) *reinterpret_cast<float4*>(p.mask_colsum + (int64_t)(row0 >> 5) * p.N + col0 + 4 * lane) = cs;
      }
      if (MASKS && p.relu_mask && (lane & 7) == 0) {
#pragma unroll
        for (int ui = 0; ui < NU; ++ui) {
          const int u = 64 * ui + lane, r = u / C4, c4 = u % C4;
          if (UNITS % 64 == 0 || u < UNITS) p.relu_mask[(int64_t)(row0 + r) * (p.N >> 5) + (col0 >> 5) + (c4 >> 3)] = wd[ui];
        }
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);        // band reads done before the next band overwrites
      __builtin_amdgcn_wave_barrier();
    });
    return;
  }
  // edge tiles / unaligned outputs: guarded scalar path
  gemm_static_for<WM>([&](auto i_c) {
    constexpr int i = decltype(i_c)::value;
    gemm_static_for<WN>([&](auto j_c) {
      constexpr int j = decltype(j_c)::value;
      const int col = n0 + (wave_n * WN + j) * 32 + li;
      const int rbase = m0 + (wave_m * WM + i) * 32 + 4 * lh;
      if (col < p.N) {
        const float bv = bias ? bias[col] : 0.f;
        float old[16];
        float csum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + (r & 3) + 8 * (r >> 2);
          old[r] = (beta != 0.f && row < p.M) ? beta * C[(int64_t)row * ldc + col] : 0.f;
          if (R && row < p.M) old[r] += R[(int64_t)row * ldc + col];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + (r & 3) + 8 * (r >> 2);
          float v = p.alpha * acc[i][j][r] + bv + old[r];
          // (masks: N is a multiple of 32, so `col < N` holds for whole waves; a 32-lane half shares its row)
          if (MASKS && p.grad_mask && row < p.M) v = ((p.grad_mask[(int64_t)row * (p.N >> 5) + (col >> 5)] >> li) & 1u) ? v : 0.f;
          if (MASKS && p.mask_colsum && row < p.M) csum += v;
          if (MASKS && p.relu_mask) {
            const uint64_t b = __ballot(v >= 0.f);
            v = relu_keep_nan(v);
            if (li == 0 && row < p.M) p.relu_mask[(int64_t)row * (p.N >> 5) + (col >> 5)] = (uint32_t)(b >> (32 * lh));
          }
          if (row < p.M) C[(int64_t)row * ldc + col] = v;
        }
        if (MASKS && p.mask_colsum) {
          csum += __shfl_xor(csum, 32);
          const int band_row = m0 + (wave_m * WM + i) * 32;
          if (lh == 0 && band_row < p.M) p.mask_colsum[(int64_t)(band_row >> 5) * p.N + col] = csum;
        }
      }
    });
  });
}

// C = beta*C + sum_s ws[s] + bias (+ residual), deterministic order.  Rows of the slabs are N
// floats; when N and ldc are multiples of 4 every thread combines one 16-byte piece.
__global__ void gemm_splitk_reduce_kernel(GemmParams p, int nbatch, int vec) {
  const int64_t mn = (int64_t)p.M * p.N;
  if (vec) {
    const int n4 = p.N >> 2;
    const int64_t total4 = (mn >> 2) * nbatch;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total4;
         i += (int64_t)gridDim.x * blockDim.x) {
      const int batch = (int)(i / (mn >> 2));
      const int64_t e4 = i - batch * (mn >> 2);
      const int row = (int)(e4 / n4), c4 = (int)(e4 - (int64_t)row * n4);
      const float4* w = reinterpret_cast<const float4*>(p.ws + (int64_t)batch * p.splits * mn) + e4;
      float4 s = w[0];
#pragma unroll 8
      for (int k = 1; k < p.splits; ++k) {
        const float4 t = w[(int64_t)k * (mn >> 2)];
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
      }
      const int b1 = batch / p.nb2, b2 = batch % p.nb2;
      const int64_t off = b1 * p.c_bs1 + b2 * p.c_bs2 + (int64_t)row * p.ldc + 4 * c4;
      if (p.bias) { const float4 b = *reinterpret_cast<const float4*>(p.bias + 4 * c4); s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w; }
      if (p.residual) { const float4 r = *reinterpret_cast<const float4*>(p.residual + off); s.x += r.x; s.y += r.y; s.z += r.z; s.w += r.w; }
      float4* dst = reinterpret_cast<float4*>(p.C + off);
      if (p.beta != 0.f) { const float4 o = *dst; s.x += p.beta * o.x; s.y += p.beta * o.y; s.z += p.beta * o.z; s.w += p.beta * o.w; }
      *dst = s;
    }
    return;
  }
  const int64_t total = mn * nbatch;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int batch = (int)(i / mn);
    const int64_t e = i - batch * mn;
    const int row = (int)(e / p.N), col = (int)(e - (int64_t)row * p.N);
    const float* w = p.ws + (int64_t)batch * p.splits * mn + e;
    float s = 0.f;
    for (int k = 0; k < p.splits; ++k) s += w[k * mn];
    const int b1 = batch / p.nb2, b2 = batch % p.nb2;
    const int64_t off = b1 * p.c_bs1 + b2 * p.c_bs2 + (int64_t)row * p.ldc + col;
    if (p.bias) s += p.bias[col];
    if (p.residual) s += p.residual[off];
    if (p.beta != 0.f) s += p.beta * p.C[off];
    p.C[off] = s;
  }
}
