// The wave-streaming weight-gradient kernels (K = tokens >> M, N; HISTORY.md 4.2) and the skinny decode product of
// csrc/gemm.hip (device side).  Included by gemm.hip only, after gemm_tiled.h (GemmParams).
#pragma once
#include "gemm_tiled.h"

// ======================================================================================
// Weight-gradient GEMM  dW = x^T @ g:  A is M-contiguous (a_rs == 1), B is N-contiguous
// (b_cs == 1), the output is small (a few hundred rows/columns) and K = tokens is huge.
// The tiled kernel above cannot fill 1024 SIMDs from a 3x3-tile output without deep k-splits
// whose blocks are too short to amortise prologue / epilogue.  Here every WAVE owns a whole
// (TM*32 x TN*32) output tile in registers and streams its own k-range straight from global
// memory into MFMA operands: lane (li, lh) of step s needs A[k = 2s+lh][m = li + 32 i] and
// B[k][n = li + 32 j], which are 128-byte coalesced dword loads -- no LDS, no barriers, no
// transposes.  A workgroup is NW waves on NW consecutive k-ranges of one tile; they are summed
// through LDS in a fixed order, so a k-split of s blocks leaves only s slabs for the
// deterministic reduce pass.  Blocks are numbered XCD-major: all tiles of one k-range run on the
// same XCD and share the x / g panels in its L2.
// ======================================================================================
template <int TM, int TN, bool EDGE>
__device__ __forceinline__ void tn_load(float (&a)[4][TM], float (&b)[4][TN], const float* __restrict__ ap,
                                        const float* __restrict__ bp, int64_t a_k2, int64_t b_k2,
                                        const bool (&mok)[TM], const bool (&nok)[TN]) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
#pragma unroll
    for (int i = 0; i < TM; ++i) a[s][i] = (!EDGE || mok[i]) ? ap[s * a_k2 + 32 * i] : 0.f;
#pragma unroll
    for (int j = 0; j < TN; ++j) b[s][j] = (!EDGE || nok[j]) ? bp[s * b_k2 + 32 * j] : 0.f;
  }
}

template <int TM, int TN>
__device__ __forceinline__ void tn_mfma(f32x16 (&acc)[TM][TN], const float (&a)[4][TM], const float (&b)[4][TN]) {
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][i], b[s][j], acc[i][j], 0, 0, 0);
}

template <int TM, int TN, int NW, bool EDGE>
__global__ __launch_bounds__(NW * 64, 1) void gemm_tn_stream_kernel(GemmParams p) {
  __shared__ float red[NW][1024];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: scalar loop control
  const int li = lane & 31, lh = lane >> 5;
  const int tiles = p.tiles_m * p.tiles_n;
  int L;
  {
    const int nwg = tiles * p.splits;
    const int bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int split = L / tiles, tile = L - split * tiles;
  const int tile_m = tile % p.tiles_m, tile_n = tile / p.tiles_m;
  const int batch = blockIdx.y, bb1 = batch / p.nb2, bb2 = batch - bb1 * p.nb2;
  p.A += bb1 * p.a_bs1 + bb2 * p.a_bs2;
  p.B += bb1 * p.b_bs1 + bb2 * p.b_bs2;
  p.C += bb1 * p.c_bs1 + bb2 * p.c_bs2;
  if (p.residual) p.residual += bb1 * p.c_bs1 + bb2 * p.c_bs2;
  p.ws += (int64_t)batch * p.splits * p.M * p.N;
  const int m0 = tile_m * (TM * 32), n0 = tile_n * (TN * 32);
  const int kw = p.k_per_split / NW;                       // multiple of 8 (host)
  const int k0 = min(p.K, split * p.k_per_split + wave * kw);
  const int k1 = min(p.K, k0 + kw);

  bool mok[TM], nok[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) mok[i] = m0 + 32 * i + li < p.M;
#pragma unroll
  for (int j = 0; j < TN; ++j) nok[j] = n0 + 32 * j + li < p.N;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const float* __restrict__ ap = p.A + (int64_t)(k0 + lh) * p.a_cs + m0 + li;
  const float* __restrict__ bp = p.B + (int64_t)(k0 + lh) * p.b_rs + n0 + li;
  const int64_t a_k2 = 2 * p.a_cs, b_k2 = 2 * p.b_rs;
  const int ngroups = (k1 - k0) >> 3;                      // groups of 4 MFMA steps = 8 k
  float a0[4][TM], b0[4][TN], a1[4][TM], b1[4][TN];
  if (ngroups > 0) tn_load<TM, TN, EDGE>(a0, b0, ap, bp, a_k2, b_k2, mok, nok);
  int g = 0;
  for (; g + 2 <= ngroups; g += 2) {
    tn_load<TM, TN, EDGE>(a1, b1, ap + 4 * a_k2, bp + 4 * b_k2, a_k2, b_k2, mok, nok);
    tn_mfma<TM, TN>(acc, a0, b0);
    ap += 8 * a_k2; bp += 8 * b_k2;
    if (g + 2 < ngroups) tn_load<TM, TN, EDGE>(a0, b0, ap, bp, a_k2, b_k2, mok, nok);
    tn_mfma<TM, TN>(acc, a1, b1);
  }
  if (g < ngroups) {                                       // odd group count: a0/b0 hold the last one
    tn_mfma<TM, TN>(acc, a0, b0);
    ap += 4 * a_k2; bp += 4 * b_k2;
  }
  // k tail (< 8 values): guarded steps
  for (int k = k0 + 8 * ngroups; k < k1; k += 2) {
    const bool kok = k + lh < k1;
    float av[TM], bv[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) av[i] = (kok && mok[i]) ? ap[32 * i] : 0.f;
#pragma unroll
    for (int j = 0; j < TN; ++j) bv[j] = (kok && nok[j]) ? bp[32 * j] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    ap += a_k2; bp += b_k2;
  }

  // ---- sum the NW waves tile by tile through LDS (fixed order) and store ------------------
  const bool partial = p.splits > 1;
  float* __restrict__ C = partial ? p.ws + (int64_t)split * p.M * p.N : p.C;
  const int64_t ldc = partial ? p.N : p.ldc;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        red[wave][((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = acc[i][j][r];
      __syncthreads();
      for (int e = tid; e < 1024; e += NW * 64) {
        float s = red[0][e];
#pragma unroll
        for (int w = 1; w < NW; ++w) s += red[w][e];
        const int gm = m0 + 32 * i + (e >> 5), gn = n0 + 32 * j + (e & 31);
        if (gm < p.M && gn < p.N) {
          float* dst = C + (int64_t)gm * ldc + gn;
          float v = p.alpha * s;
          if (!partial) {
            if (p.bias) v += p.bias[gn];
            if (p.residual) v += p.residual[(int64_t)gm * ldc + gn];
            if (p.beta != 0.f) v += p.beta * *dst;
          }
          *dst = v;
        }
      }
      __syncthreads();
    }
  }
}

template <int AQ, int BQ, bool EDGE, bool TAIL>
__device__ __forceinline__ void tn_load4(float4 (&ra)[AQ], float4 (&rb)[BQ], const float* __restrict__ ap,
                                         const float* __restrict__ bp, const int64_t (&aoff)[AQ],
                                         const int64_t (&boff)[BQ], const int (&arow)[AQ], const int (&brow)[BQ],
                                         const bool (&aok)[AQ], const bool (&bok)[BQ], int rem) {
#pragma unroll
  for (int q = 0; q < AQ; ++q) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((!EDGE || aok[q]) && (!TAIL || arow[q] < rem)) v = *reinterpret_cast<const float4*>(ap + aoff[q]);
    ra[q].x = v.x; ra[q].y = v.y; ra[q].z = v.z; ra[q].w = v.w;
  }
#pragma unroll
  for (int q = 0; q < BQ; ++q) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((!EDGE || bok[q]) && (!TAIL || brow[q] < rem)) v = *reinterpret_cast<const float4*>(bp + boff[q]);
    rb[q].x = v.x; rb[q].y = v.y; rb[q].z = v.z; rb[q].w = v.w;
  }
}

template <int AQ, int BQ>
__device__ __forceinline__ void tn_park(float* dst, const float4 (&ra)[AQ], const float4 (&rb)[BQ],
                                        const int (&alds)[AQ], const int (&blds)[BQ]) {
#pragma unroll
  for (int q = 0; q < AQ; ++q) *reinterpret_cast<float4*>(dst + alds[q]) = ra[q];
#pragma unroll
  for (int q = 0; q < BQ; ++q) *reinterpret_cast<float4*>(dst + blds[q]) = rb[q];
}

// Same decomposition with 16-byte global loads: a wave fetches 8 k-rows of its A and B panels as
// float4 pieces (3 + 3 instructions instead of 24 + 24 dword loads: the texture addresser is paid
// per instruction), parks them in a wave-private LDS strip [8][TM*32] / [8][TN*32] and reads the
// MFMA operands back as dwords (row 2s+lh, column 32i+li: the two half-waves sit 96 floats = 32
// banks apart, conflict-free).  Only wave-level ordering is involved -- no barriers in the loop.
// ABLATE (tools/micro/stream_ablate.hip only; 0 in the library): 1 = no global loads after the
// prologue, 2 = no LDS parking, 4 = operands read from LDS once -- timing experiments, wrong results.
template <int TM, int TN, int NW, bool EDGE, int ABLATE = 0>
__global__ __launch_bounds__(NW * 64, 1) void gemm_tn_stream_lds_kernel(GemmParams p) {
  constexpr int AW = TM * 32, BW = TN * 32, ROWS = 8;
  constexpr int AQ = ROWS * AW / 4 / 64, BQ = ROWS * BW / 4 / 64;
  static_assert(ROWS * AW / 4 % 64 == 0 && ROWS * BW / 4 % 64 == 0, "strip must be whole wave loads");
  constexpr int STRIP = ROWS * (AW + BW);
  static_assert(2 * STRIP >= 1024, "reduce buffer must fit in the staging strips");
  __shared__ __attribute__((aligned(16))) float smem[NW * 2 * STRIP];   // two strips per wave
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: scalar loop control
  const int li = lane & 31, lh = lane >> 5;
  const int tiles = p.tiles_m * p.tiles_n;
  int L;
  {
    const int nwg = tiles * p.splits;
    const int bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int split = L / tiles, tile = L - split * tiles;
  const int tile_m = tile % p.tiles_m, tile_n = tile / p.tiles_m;
  const int batch = blockIdx.y, bb1 = batch / p.nb2, bb2 = batch - bb1 * p.nb2;
  p.A += bb1 * p.a_bs1 + bb2 * p.a_bs2;
  p.B += bb1 * p.b_bs1 + bb2 * p.b_bs2;
  p.C += bb1 * p.c_bs1 + bb2 * p.c_bs2;
  if (p.residual) p.residual += bb1 * p.c_bs1 + bb2 * p.c_bs2;
  p.ws += (int64_t)batch * p.splits * p.M * p.N;
  const int m0 = tile_m * AW, n0 = tile_n * BW;
  const int kw = p.k_per_split / NW;
  const int k0 = min(p.K, split * p.k_per_split + wave * kw);
  const int k1 = min(p.K, k0 + kw);

  float* strip = smem + wave * (2 * STRIP);
  // per-lane pieces of a strip: piece f = lane + 64 q -> row f / (W/4), float4 column f % (W/4)
  int64_t aoff[AQ], boff[BQ];
  int arow[AQ], brow[BQ], alds[AQ], blds[BQ];
  bool aok[AQ], bok[BQ];
#pragma unroll
  for (int q = 0; q < AQ; ++q) {
    const int f = lane + 64 * q, row = f / (AW / 4), c4 = f % (AW / 4);
    arow[q] = row; alds[q] = row * AW + 4 * c4;
    aok[q] = m0 + 4 * c4 < p.M;
    aoff[q] = (int64_t)row * p.a_cs + m0 + 4 * c4;
  }
#pragma unroll
  for (int q = 0; q < BQ; ++q) {
    const int f = lane + 64 * q, row = f / (BW / 4), c4 = f % (BW / 4);
    brow[q] = row; blds[q] = ROWS * AW + row * BW + 4 * c4;
    bok[q] = n0 + 4 * c4 < p.N;
    boff[q] = (int64_t)row * p.b_rs + n0 + 4 * c4;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const float* __restrict__ ap = p.A + (int64_t)k0 * p.a_cs;
  const float* __restrict__ bp = p.B + (int64_t)k0 * p.b_rs;
  const int64_t a_g = 8 * p.a_cs, b_g = 8 * p.b_rs;
  const int nk = k1 - k0;
  const int ngroups = (nk + 7) >> 3;       // the last group may be partial (rows >= nk read as 0)
  float4 ra[AQ], rb[BQ];
  // Pipeline: strip[g&1] holds group g, ra/rb hold group g+1 (in flight).  Per group: issue the 24
  // operand reads of group g, then (while they fly) park group g+1 in the other strip and issue
  // the global loads of group g+2, then run the 36 MFMAs.
#define TN_FETCH(G)                                                                              \
  {                                                                                              \
    if ((G) < ngroups - 1) tn_load4<AQ, BQ, EDGE, false>(ra, rb, ap, bp, aoff, boff, arow, brow, aok, bok, 8); \
    else if ((G) == ngroups - 1) tn_load4<AQ, BQ, EDGE, true>(ra, rb, ap, bp, aoff, boff, arow, brow, aok, bok, nk - 8 * (G)); \
    ap += a_g; bp += b_g;                                                                        \
  }
  if (ngroups > 0) {
    TN_FETCH(0)
    tn_park<AQ, BQ>(strip, ra, rb, alds, blds);
    TN_FETCH(1)
  }
  for (int g = 0; g < ngroups; ++g) {
    const float* sA = strip + (g & 1) * STRIP;
    const float* sB = sA + ROWS * AW;
    float a[4][TM], b[4][TN];
    if (!(ABLATE & 4) || g == 0) {
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[s2][i] = sA[(2 * s2 + lh) * AW + 32 * i + li];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[s2][j] = sB[(2 * s2 + lh) * BW + 32 * j + li];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (g + 1 < ngroups) {
      if (!(ABLATE & 2)) tn_park<AQ, BQ>(strip + ((g + 1) & 1) * STRIP, ra, rb, alds, blds);
      if (!(ABLATE & 1)) TN_FETCH(g + 2)
    }
    __builtin_amdgcn_sched_barrier(0);
    tn_mfma<TM, TN>(acc, a, b);
  }
#undef TN_FETCH
  __syncthreads();                         // staging strips become the reduce buffer

  float (*red)[1024] = reinterpret_cast<float (*)[1024]>(smem);
  const bool partial = p.splits > 1;
  float* __restrict__ C = partial ? p.ws + (int64_t)split * p.M * p.N : p.C;
  const int64_t ldc = partial ? p.N : p.ldc;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        red[wave][((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = acc[i][j][r];
      __syncthreads();
      for (int e = tid; e < 1024; e += NW * 64) {
        float s = red[0][e];
#pragma unroll
        for (int w = 1; w < NW; ++w) s += red[w][e];
        const int gm = m0 + 32 * i + (e >> 5), gn = n0 + 32 * j + (e & 31);
        if (gm < p.M && gn < p.N) {
          float* dst = C + (int64_t)gm * ldc + gn;
          float v = p.alpha * s;
          if (!partial) {
            if (p.bias) v += p.bias[gn];
            if (p.residual) v += p.residual[(int64_t)gm * ldc + gn];
            if (p.beta != 0.f) v += p.beta * *dst;
          }
          *dst = v;
        }
      }
      __syncthreads();
    }
  }
}

// Exact-fit shapes (M, N multiples of the wave tile, 16-byte aligned panels): the strips are filled
// by LDS-DMA (`global_load_lds_dwordx4`: 1 KiB per wave instruction, destination = wave-uniform
// base + lane*16, which is exactly the row-major [8][96] strip), so the panel data never passes
// through VGPRs, there is no ds_write pass, and three strips per wave give a prefetch distance of
// two groups (the only ordering needed is the issuing wave's own vmcnt: MI355X_MICROARCH item 7).
__device__ __forceinline__ void tn_glds16(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int TM, int TN, int NW, int ABLATE = 0>
__global__ __launch_bounds__(NW * 64, 1) void gemm_tn_stream_dma_kernel(GemmParams p) {
  constexpr int AW = TM * 32, BW = TN * 32, ROWS = 8, NS = 3;
  constexpr int AQ = ROWS * AW / 4 / 64, BQ = ROWS * BW / 4 / 64;
  static_assert(ROWS * AW / 4 % 64 == 0 && ROWS * BW / 4 % 64 == 0, "strip must be whole wave loads");
  constexpr int STRIP = ROWS * (AW + BW);
  static_assert(NS * STRIP >= 1024, "reduce buffer must fit in the staging strips");
  __shared__ __attribute__((aligned(16))) float smem[NW * NS * STRIP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: scalar loop control
  const int li = lane & 31, lh = lane >> 5;
  const int tiles = p.tiles_m * p.tiles_n;
  int L;
  {
    const int nwg = tiles * p.splits;
    const int bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int split = L / tiles, tile = L - split * tiles;
  const int tile_m = tile % p.tiles_m, tile_n = tile / p.tiles_m;
  const int batch = blockIdx.y, bb1 = batch / p.nb2, bb2 = batch - bb1 * p.nb2;
  p.A += bb1 * p.a_bs1 + bb2 * p.a_bs2;
  p.B += bb1 * p.b_bs1 + bb2 * p.b_bs2;
  p.C += bb1 * p.c_bs1 + bb2 * p.c_bs2;
  if (p.residual) p.residual += bb1 * p.c_bs1 + bb2 * p.c_bs2;
  p.ws += (int64_t)batch * p.splits * p.M * p.N;
  const int m0 = tile_m * AW, n0 = tile_n * BW;
  const int kw = p.k_per_split / NW;
  const int k0 = min(p.K, split * p.k_per_split + wave * kw);
  const int k1 = min(p.K, k0 + kw);

  float* strip = smem + wave * (NS * STRIP);
  int64_t aoff[AQ], boff[BQ];
#pragma unroll
  for (int q = 0; q < AQ; ++q) {
    const int f = lane + 64 * q, row = f / (AW / 4), c4 = f % (AW / 4);
    aoff[q] = (int64_t)row * p.a_cs + m0 + 4 * c4;
  }
#pragma unroll
  for (int q = 0; q < BQ; ++q) {
    const int f = lane + 64 * q, row = f / (BW / 4), c4 = f % (BW / 4);
    boff[q] = (int64_t)row * p.b_rs + n0 + 4 * c4;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const float* __restrict__ ap = p.A + (int64_t)k0 * p.a_cs;
  const float* __restrict__ bp = p.B + (int64_t)k0 * p.b_rs;
  const int64_t a_g = 8 * p.a_cs, b_g = 8 * p.b_rs;
  const int nk = k1 - k0;
  const int nfull = nk >> 3;                 // whole 8-row groups: LDS-DMA pipeline
  // one group = AQ + BQ DMA instructions; `ap`/`bp` always point at the next group to fetch
#define TN_DMA(SLOT)                                                                          \
  {                                                                                           \
    float* d = strip + (SLOT) * STRIP;                                                        \
    _Pragma("unroll") for (int q = 0; q < AQ; ++q) tn_glds16(ap + aoff[q], d + q * 256);      \
    _Pragma("unroll") for (int q = 0; q < BQ; ++q) tn_glds16(bp + boff[q], d + ROWS * AW + q * 256); \
    ap += a_g; bp += b_g;                                                                     \
  }
  if (nfull > 0) TN_DMA(0)
  if (nfull > 1) TN_DMA(1)
  int slot = 0;
  for (int g = 0; g < nfull; ++g) {
    if (!(ABLATE & 1) || g < 2) {
      if (g + 2 < nfull) {
        const int s2 = slot >= 1 ? slot - 1 : NS - 1;    // (g + 2) % 3
        TN_DMA(s2)
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  // groups g+1, g+2 may still be in flight
      } else if (g + 1 < nfull) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    const float* sA = strip + slot * STRIP;
    const float* sB = sA + ROWS * AW;
    float a[4][TM], b[4][TN];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
      for (int i = 0; i < TM; ++i) a[s4][i] = sA[(2 * s4 + lh) * AW + 32 * i + li];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[s4][j] = sB[(2 * s4 + lh) * BW + 32 * j + li];
    }
    __builtin_amdgcn_sched_barrier(0);
    tn_mfma<TM, TN>(acc, a, b);
    slot = slot + 1 == NS ? 0 : slot + 1;
  }
#undef TN_DMA
  // k tail (< 8 rows): guarded dword loads straight into MFMA operands
  {
    const float* at = p.A + (int64_t)(k0 + 8 * nfull + lh) * p.a_cs + m0 + li;
    const float* bt = p.B + (int64_t)(k0 + 8 * nfull + lh) * p.b_rs + n0 + li;
    for (int k = k0 + 8 * nfull; k < k1; k += 2) {
      const bool kok = k + lh < k1;
      float av[TM], bv[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) av[i] = kok ? at[32 * i] : 0.f;
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[j] = kok ? bt[32 * j] : 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
      at += 2 * p.a_cs; bt += 2 * p.b_rs;
    }
  }
  __syncthreads();                         // staging strips become the reduce buffer

  float (*red)[1024] = reinterpret_cast<float (*)[1024]>(smem);
  const bool partial = p.splits > 1;
  float* __restrict__ C = partial ? p.ws + (int64_t)split * p.M * p.N : p.C;
  const int64_t ldc = partial ? p.N : p.ldc;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        red[wave][((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = acc[i][j][r];
      __syncthreads();
      for (int e = tid; e < 1024; e += NW * 64) {
        float s = red[0][e];
#pragma unroll
        for (int w = 1; w < NW; ++w) s += red[w][e];
        const int gm = m0 + 32 * i + (e >> 5), gn = n0 + 32 * j + (e & 31);
        float* dst = C + (int64_t)gm * ldc + gn;
        float v = p.alpha * s;
        if (!partial) {
          if (p.bias) v += p.bias[gn];
          if (p.residual) v += p.residual[(int64_t)gm * ldc + gn];
          if (p.beta != 0.f) v += p.beta * *dst;
        }
        *dst = v;
      }
      __syncthreads();
    }
  }
}

// ======================================================================================
// Skinny product for the decode path: C (MR x N) = A (MR x K, rows contiguous) * B (K x N,
// N-contiguous), MR <= 4 (one token per sequence).  Pure weight streaming: a workgroup owns 128
// columns (32 float4 lanes) x 8 k-slices, every lane walks its slice of the K rows with 16-byte
// loads, the slices are combined through LDS in a fixed order.  No MFMA: 2*MR flops per weight.
// ======================================================================================
template <int MR>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmParams p) {
  __shared__ float4 red[8][MR][32];
  const int tx = threadIdx.x & 31, ks = threadIdx.x >> 5;
  const int n = blockIdx.x * 128 + tx * 4;
  float4 acc[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) acc[m] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (n < p.N) {
    const float* __restrict__ bp = p.B + n;
#pragma unroll 4
    for (int k = ks; k < p.K; k += 8) {
      const float4 w = *reinterpret_cast<const float4*>(bp + (int64_t)k * p.b_rs);
#pragma unroll
      for (int m = 0; m < MR; ++m) {
        const float a = p.A[(int64_t)m * p.a_rs + k];
        acc[m].x += a * w.x; acc[m].y += a * w.y; acc[m].z += a * w.z; acc[m].w += a * w.w;
      }
    }
  }
#pragma unroll
  for (int m = 0; m < MR; ++m) red[ks][m][tx] = acc[m];
  __syncthreads();
  if (ks == 0 && n < p.N) {
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      float4 v = red[0][m][tx];
#pragma unroll
      for (int q = 1; q < 8; ++q) {
        const float4 t = red[q][m][tx];
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
      v.x *= p.alpha; v.y *= p.alpha; v.z *= p.alpha; v.w *= p.alpha;
      float* dst = p.C + (int64_t)m * p.ldc + n;
      if (p.bias) { const float4 b = *reinterpret_cast<const float4*>(p.bias + n); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
      if (p.residual) { const float4 r = *reinterpret_cast<const float4*>(p.residual + (int64_t)m * p.ldc + n); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
      if (p.beta != 0.f) { const float4 o = *reinterpret_cast<const float4*>(dst); v.x += p.beta * o.x; v.y += p.beta * o.y; v.z += p.beta * o.z; v.w += p.beta * o.w; }
      *reinterpret_cast<float4*>(dst) = v;
    }
  }
}
