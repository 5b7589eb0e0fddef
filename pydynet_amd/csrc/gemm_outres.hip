// Output-resident fp32 GEMM for products that END in the model width (gfx950 only).
//
//   C (M x 288) = A (M x K, rows contiguous) * B (K x 288)  (+ bias[288]) (+ residual[M x 288])
//
// Every second product of a transformer block has the model width (288 in the benchmarked Llama) as its
// OUTPUT and something longer as its contraction: ctx Wo and h W_down forward (llm/llama/model.py:121, 58),
// the input gradients dqkv Wqkv^T, dgu Wgu^T, dctx Wo^T (`grad @ W^T`, pydynet/core/tensor.py:670) and
// dlogits W_out (K = 32000).  A tile kernel re-reads A once per column tile (three times at 96 columns) and
// stages both operands through LDS.  Here
//   * a wave owns 32 rows of C and keeps ALL 288 columns of them in accumulators (nine 32 x 32 MFMA tiles,
//     144 registers): A is read exactly once, straight from global memory into MFMA A-operand registers
//     (lane (i, h) loads A[row i][8t + 4h .. +3]; a four-group ring keeps one k-piece of loads in flight),
//     and never passes through LDS;
//   * B is streamed through LDS in 32 (k) x 288 (n) pieces by LDS-DMA (`global_load_lds_dwordx4`), double
//     buffered, shared by the four waves of a workgroup, its instructions spread over the MFMA stream:
//     one barrier per 144 MFMAs per wave;
//   * B row-major [k][n] (`x @ W`): LDS image [32][288], fragments are conflict-free ds_read_b32;
//     B^T row-major [n][k] (`grad @ W^T`): LDS image [288][32] with the eight 16-byte units of row n
//     XOR-swizzled by (n >> 1) & 7 on the source side of the DMA: one conflict-free ds_read_b128 per four MFMAs;
//   * C leaves the accumulators once, every store instruction writing two full 128-byte row segments.
// The k-order of every output element is that of csrc/gemm.hip (k = 8t + 4h + j inside an MFMA group, groups
// ascending), so the results are bit-identical to the tiled kernel's when that one does not split K.
#include "common.h"
#include <stdlib.h>

// Timing-ablation switches (tools/*_probe.py): read ONCE per process; a non-zero value makes kernels skip work and return
// WRONG results, so it is announced on stderr instead of taking effect silently.
static int pdn_ablation_switch(const char* name) {
  const char* e = getenv(name);
  const int v = e ? atoi(e) : 0;
  if (v) fprintf(stderr, "[pdnhip] WARNING: %s=%d -- timing ablation active, results of the affected kernels are WRONG\n", name, v);
  return v;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct OutResParams {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  const float* residual;
  int M, K;
  int64_t lda, ldb, ldc;
  // CE instantiation: A holds LOGITS and the operand is formed as it is consumed,
  //   a = (exp(logit - lse[row]) - [column == target[row]]) * gscale * (gdev ? *gdev : 1)
  // -- the gradient of cross entropy w.r.t. the logits (nn/functional.py:364-381) never exists in memory
  const float* lse;
  const int64_t* targets;
  const float* gdev;
  float gscale;
  // K split over grid.y (short M: the row workgroups alone do not fill the chip): split y multiplies pieces
  // [y * kps, (y + 1) * kps) into slab y of `slab` (M x 288 each, no bias / residual); outres_splitk_reduce adds them up
  int kps;
  float* slab;
  // NT form with B^T in BLOCKS along the contraction (dX = [d_1 | d_2 | ...] [W_1 | W_2 | ...]^T with the weights where
  // they live -- equally spaced (288 x kb) matrices -- instead of a column-packed copy made every step): piece s comes
  // from block s / ppb (ppb = kb / 32 pieces per block; `ppb_magic` = ceil(2^32 / ppb)).  ppb = 0: one matrix.
  int ppb;
  unsigned ppb_magic;
  int64_t b_bstride;
  // CE == 2 (deferred normalisation): `lse` holds the ROW MAXIMUM m of the logits; the operand is exp(logit - m), its row
  // sums Z accumulate beside the product, and the rows leave as gscale * (acc / Z - W^T[target]) -- the same gradient,
  // with the softmax denominator found on the way: lse_out[row] = m + log Z (the statistics pass over the logits is gone)
  float* lse_out;
  int max_parts;                  // the row maxima come as `max_parts` vectors of M (the projection's chunk ranges)
  float* zslab;                   // K split over grid.y: split y leaves its UNNORMALISED rows in slab y and its row sums in
                                  // zslab + y * M; outres_ce2_reduce_kernel normalises
  int ablate_rt;                  // timing experiments (PDN_OUTRES_RT_ABLATE; 0 in the library): 1 = no epilogue (nothing stored)
  const float* Wt;                // CE 2: W^T (V x 288) row-major -- the rows W[:, target] of the normalising store are read as
                                  // 1152 contiguous bytes per token (from W itself they are 288 floats 4 V bytes apart: one
                                  // 64-byte sector each, 2.1 GB of extra HBM reads per step at the benchmark shape)
};

// gradient of the mean cross entropy w.r.t. one logit
__device__ __forceinline__ float ce_grad(float logit, float lse, bool is_target, float sc) {
  return (__expf(logit - lse) - (is_target ? 1.f : 0.f)) * sc;
}

// the same without the scale and with the row's  -lse * log2(e)  prepared: one fma, one exp2, one select-subtract per logit;
// the weight-gradient kernel applies the scale to its 144 accumulators (and the column sums) once, at the end
__device__ __forceinline__ float ce_unscaled(float logit, float neg_lse_l2e, bool is_target) {
  const float e = __builtin_amdgcn_exp2f(fmaf(logit, 1.4426950408889634f, neg_lse_l2e));
  return is_target ? e - 1.f : e;
}

__device__ __forceinline__ void or_glds16(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

#define OR_N 288          // output columns (nine 32-wide MFMA tiles)
#define OR_KP 32          // contraction rows per piece (four 8-row groups)

// BT: B is given as the row-major (288 x K) matrix whose transpose is meant.  NW: waves per workgroup.
// ABLATE (timing experiments only): 1 = A loaded once, 2 = no B DMA after the first two pieces.
// STAGE: the B piece reaches LDS by LDS-DMA (0) or through registers, global_load_dwordx4 + ds_write_b128 (1).
template <bool BT, int NW, int STAGE, int ABLATE = 0, int CE = 0>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void gemm_outres_kernel(OutResParams p) {
  static_assert(CE != 2 || BT, "deferred cross-entropy normalisation: NT form (W as stored)");
  constexpr int PIECE = OR_KP * OR_N;             // floats: 36 KiB
  constexpr int NQ = (36 + NW - 1) / NW;          // DMA instructions per wave and piece
  static_assert(NQ <= 9, "the DMA of a piece must be issued within its first three k-groups");
  __shared__ __attribute__((aligned(16))) float smem[2 * PIECE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int m0 = (blockIdx.x * NW + wave) * 32;
  const int npieces_all = p.K / OR_KP;
  const int s0 = p.slab ? (int)blockIdx.y * p.kps : 0;
  const int npieces = p.slab ? min(npieces_all, s0 + p.kps) : npieces_all;   // pieces [s0, npieces) are this block's
  const unsigned ldb = (unsigned)p.ldb;

  // DMA instruction I (0..35) of a piece covers its 16-byte units 64 I .. 64 I + 63 (LDS side linear in the lane).
  // NN: unit u = row k = u / 72, column unit u % 72.   NT: unit u = row n = u / 8, k unit (u % 8) ^ ((n >> 1) & 7).
  // (the lane offset is recomputed per instruction, in the shadow of the MFMAs, instead of occupying registers)
  // (STAGE: the NQ lane offsets live in registers -- recomputed per instruction they were ~8 VALU operations each, 40 of
  //  the ~110 non-MFMA vector instructions a wave issued per 144 MFMAs (PMC, round 4); the LDS-DMA form keeps recomputing:
  //  its 4-wave instantiations have no registers to spare)
  constexpr bool PSO = STAGE && NW == 8;           // (the 4-wave forms carry NQ = 9 offsets: they spill)
  unsigned pso[PSO ? NQ : 1];
  auto lane_off = [&](int q, int ln) -> unsigned {
    const int u = 64 * min(q * NW + wave, 35) + ln;
    if (BT) {
      const int n = u >> 3, cu = (u & 7) ^ ((n >> 1) & 7);
      return (unsigned)n * ldb + 4u * (unsigned)cu;
    }
    const int k = u / 72, n4 = u - 72 * k;
    return (unsigned)k * ldb + 4u * (unsigned)n4;
  };
  if (PSO) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) pso[q] = lane_off(q, lane);
  }
  auto piece_src = [&](int piece, int q, int& I) -> const float* {
    I = min(q * NW + wave, 35);
    unsigned o;
    if (PSO) {
      o = pso[q];
    } else {
      int ln = lane;
      asm volatile("" : "+v"(ln));
      o = lane_off(q, ln);
    }
    const float* src;
    if (BT && p.ppb) {
      const int blk = (int)__umulhi((unsigned)piece, p.ppb_magic);
      src = p.B + (int64_t)blk * p.b_bstride + (piece - blk * p.ppb) * OR_KP;
    } else {
      src = BT ? p.B + (int64_t)piece * OR_KP : p.B + (int64_t)piece * OR_KP * p.ldb;
    }
    return src + o;
  };
  float4 rb[STAGE ? NQ : 1];                      // STAGE 1: the wave's share of the next piece on its way to LDS
  auto issue_one = [&](int buf, int piece, int q) {
    int I;
    const float* src = piece_src(piece, q, I);
    if (STAGE) {
      const float4 v = *reinterpret_cast<const float4*>(src);
      rb[q].x = v.x; rb[q].y = v.y; rb[q].z = v.z; rb[q].w = v.w;
    } else {
      or_glds16(src, smem + buf * PIECE + I * 256);
    }
  };
  auto park = [&](int buf) {                      // STAGE 1: registers -> LDS (unit 64 I + lane, linear: no conflicts)
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int I = min(q * NW + wave, 35);
      *reinterpret_cast<float4*>(smem + buf * PIECE + I * 256 + 4 * lane) = rb[q];
    }
  };

  if (npieces > s0) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) issue_one(0, s0, q);
    if (STAGE) park(0);
  }
  // A operand ring: a[g] holds group g of the current piece, reloaded for the next piece right after use
  const int arow_i = min(m0 + li, p.M - 1);
  const float* arow = p.A + (int64_t)arow_i * p.lda + 4 * lh;
  float4 a[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 v = *reinterpret_cast<const float4*>(arow + (int64_t)s0 * OR_KP + 8 * g);
    a[g].x = v.x; a[g].y = v.y; a[g].z = v.z; a[g].w = v.w;
  }
  float ce_lse = 0.f, ce_sc = 0.f;
  int ce_rel = 0;                                   // target column relative to this lane's first column (4 h)
  if (CE == 1) {
    ce_lse = p.lse[arow_i];
    ce_rel = (int)p.targets[arow_i] - 4 * lh;
    ce_sc = p.gscale * (p.gdev ? p.gdev[0] : 1.f);
  }
  float ce_c2 = 0.f, ce_z = 0.f;                    // CE 2: -max * log2(e); this lane's share of the row's sum of exponentials
  float ce_m = 0.f;
  if (CE == 2) {
    ce_m = p.lse[arow_i];
    for (int q = 1; q < p.max_parts; ++q) ce_m = fmaxf(ce_m, p.lse[(int64_t)q * p.M + arow_i]);
    ce_c2 = -ce_m * 1.4426950408889634f;
  }

  f32x16 acc[9];
#pragma unroll
  for (int j = 0; j < 9; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // B fragment addressing (floats, relative to the piece)
  const int xl = (li >> 1) & 7;
  const int bnn = (4 * lh) * OR_N + li;
  int bq[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bq[g] = BT ? li * OR_KP + 4 * ((2 * g + lh) ^ xl) : 0;

  for (int s = s0; s < npieces; ++s) {
    // every DMA of piece s was issued before the last two A loads of the previous piece (vmcnt retires in order)
    if (STAGE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wave's ds_writes of the piece
    else if (s == s0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    // (a bare s_barrier: __syncthreads() carries a fence that drains vmcnt to 0, i.e. would wait for the A loads
    //  that were just put in flight; LDS reads of the previous piece were consumed by its MFMAs already)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const float* Bp = smem + ((s - s0) & 1) * PIECE;
    const int nxt = min(s + 1, npieces - 1);       // after the last piece: a redundant fetch into the idle buffer
    const float* anext = arow + (int64_t)nxt * OR_KP;
    // Twelve (k-group, column-triple) steps per piece; the fragments of step n + 1 are read while step n
    // multiplies (two named sets, pinned by the scheduling barriers: the compiler would hoist more and spill).
    float b0[3][4], b1[3][4];
    // fragments of column tiles 3 T .. 3 T + 2 for k-group G of the piece
#define OR_LOADB(BX, G, T)                                                                       \
  if (BT) {                                                                                      \
    _Pragma("unroll") for (int j = 0; j < 3; ++j) {                                              \
      const float4 v = *reinterpret_cast<const float4*>(Bp + bq[G] + (3 * (T) + j) * (32 * OR_KP)); \
      BX[j][0] = v.x; BX[j][1] = v.y; BX[j][2] = v.z; BX[j][3] = v.w;                            \
    }                                                                                            \
  } else {                                                                                       \
    _Pragma("unroll") for (int j = 0; j < 3; ++j)                                                \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) BX[j][q] = Bp[bnn + (8 * (G) + q) * OR_N + 32 * (3 * (T) + j)]; \
  }
#define OR_MFMA(BX, AV, T)                                                                       \
  {                                                                                              \
    _Pragma("unroll") for (int j = 0; j < 3; ++j) acc[3 * (T) + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV.x, BX[j][0], acc[3 * (T) + j], 0, 0, 0); \
    _Pragma("unroll") for (int j = 0; j < 3; ++j) acc[3 * (T) + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV.y, BX[j][1], acc[3 * (T) + j], 0, 0, 0); \
    _Pragma("unroll") for (int j = 0; j < 3; ++j) acc[3 * (T) + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV.z, BX[j][2], acc[3 * (T) + j], 0, 0, 0); \
    _Pragma("unroll") for (int j = 0; j < 3; ++j) acc[3 * (T) + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV.w, BX[j][3], acc[3 * (T) + j], 0, 0, 0); \
  }
    // one step: read the next step's fragments into BN, multiply this step's from BC
#define OR_STEP(BC, BN, G, T)                                                                    \
  if (3 * (G) + (T) + 1 < 12) { OR_LOADB(BN, (3 * (G) + (T) + 1) / 3, (3 * (G) + (T) + 1) % 3) }  \
  __builtin_amdgcn_sched_barrier(0);                                                             \
  OR_MFMA(BC, av, T)                                                                             \
  __builtin_amdgcn_sched_barrier(0);
    OR_LOADB(b0, 0, 0)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 av = a[g];
      if (CE == 1) {                                // this lane's four columns: 32 s + 8 g + 4 h + 0..3
        const int c0 = ce_rel - (OR_KP * s + 8 * g);
        av.x = ce_grad(av.x, ce_lse, c0 == 0, ce_sc); av.y = ce_grad(av.y, ce_lse, c0 == 1, ce_sc);
        av.z = ce_grad(av.z, ce_lse, c0 == 2, ce_sc); av.w = ce_grad(av.w, ce_lse, c0 == 3, ce_sc);
      }
      if (CE == 2) {
        const float L2E = 1.4426950408889634f;
        av.x = __builtin_amdgcn_exp2f(fmaf(av.x, L2E, ce_c2)); av.y = __builtin_amdgcn_exp2f(fmaf(av.y, L2E, ce_c2));
        av.z = __builtin_amdgcn_exp2f(fmaf(av.z, L2E, ce_c2)); av.w = __builtin_amdgcn_exp2f(fmaf(av.w, L2E, ce_c2));
        ce_z += (av.x + av.y) + (av.z + av.w);
      }
      // the fetch of the next piece first (three instructions at the head of groups 0..2; spread one per step
      // it measured 66 instead of 77 % at K = 32000), so that only the A loads of groups 2 and 3 follow its last one
      if (g < 3 && (!(ABLATE & 2) || s < 1)) {
#pragma unroll
        for (int e = 0; e < 3; ++e)
          if (3 * g + e < NQ) issue_one((s - s0 + 1) & 1, nxt, 3 * g + e);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (g & 1) {                                  // steps 3g, 3g + 1, 3g + 2: the fragment sets alternate
        OR_STEP(b1, b0, g, 0) OR_STEP(b0, b1, g, 1) OR_STEP(b1, b0, g, 2)
      } else {
        OR_STEP(b0, b1, g, 0) OR_STEP(b1, b0, g, 1) OR_STEP(b0, b1, g, 2)
      }
      if (!(ABLATE & 1)) {                          // a[g] is free: fetch group g of the next piece
        const float4 v = *reinterpret_cast<const float4*>(anext + 8 * g);
        a[g].x = v.x; a[g].y = v.y; a[g].z = v.z; a[g].w = v.w;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#undef OR_STEP
    if (STAGE) park((s - s0 + 1) & 1);              // every wave is past this piece's barrier: that buffer is idle
#undef OR_LOADB
#undef OR_MFMA
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the redundant last fetch must not outlive the workgroup's LDS

  // ---- C rows: accumulator register r of tile j = row (r & 3) + 8 (r >> 2) + 4 h, column 32 j + lane -----------
  if (CE == 2 && p.slab) {                          // (the rows themselves: the plain slab store below)
    const float z = ce_z + __shfl_xor(ce_z, 32, 64);
    if (lh == 0 && m0 + li < p.M) p.zslab[(int64_t)blockIdx.y * p.M + m0 + li] = z;
  }
  if (CE == 2 && !p.slab) {
    // lane l (both halves) knows row l's statistics; the accumulators hold row rho(r, h) in register r: one cross-lane
    // read of 1 / Z and of the target per register row, then nine column tiles of  gscale * (acc / Z - W[:, target])
    const float z = ce_z + __shfl_xor(ce_z, 32, 64);
    const float izl = 1.f / z;
    const int tgl = min(max((int)p.targets[arow_i], 0), p.K - 1);   // (an out-of-range target is reported by the loss kernel)
    if (lh == 0 && m0 + li < p.M) p.lse_out[m0 + li] = ce_m + __logf(z);
    const int mrem2 = p.M - m0;
    float* __restrict__ Cw = p.C + (int64_t)m0 * p.ldc + li;
    const float* __restrict__ Wc = p.Wt + li;
    const float sc = p.gscale;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rho = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float iz = __shfl(izl, rho, 64);
      const int tg = __shfl(tgl, rho, 64);
      if (rho < mrem2) {
#pragma unroll
        for (int j = 0; j < 9; ++j) {
          const float wt = Wc[(unsigned)tg * (unsigned)OR_N + 32 * j];
          Cw[(unsigned)rho * (unsigned)p.ldc + 32 * j] = sc * (acc[j][r] * iz - wt);
        }
      }
    }
    return;
  }
  const bool to_slab = p.slab != nullptr;
  float* __restrict__ Cw = to_slab ? p.slab + ((int64_t)blockIdx.y * p.M + m0) * OR_N : p.C + (int64_t)m0 * p.ldc;
  const float* __restrict__ Rw = (p.residual && !to_slab) ? p.residual + (int64_t)m0 * p.ldc : nullptr;
  const unsigned ldc = to_slab ? (unsigned)OR_N : (unsigned)p.ldc;
  const int mrem = p.M - m0 - 4 * lh;               // rows rr < mrem exist
  const bool full = m0 + 32 <= p.M;
  if (p.ablate_rt & 1) return;
  if (full) {
    // Whole row blocks (every wave of the benchmarked shapes).  The nine tiles used to leave one after the other, each
    // behind a wait for its own residual loads AND for the stores before them (vmcnt(0): nine serialised round trips while
    // the whole chip does the same, ~30 us of a 266 us down projection).  Now the residual rows of tiles j + 1 and j + 2
    // are in flight while tile j is added and stored (three register sets: the operand ring and the staging registers
    // of the loop are dead here), no row tests, byte offsets beside wave-uniform bases.
    float bvs[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) bvs[j] = (p.bias && !to_slab) ? p.bias[32 * j + li] : 0.f;
    const unsigned ldcb = 4u * ldc;
    const unsigned ob0 = (unsigned)(4 * lh) * ldcb + 4u * (unsigned)li;
    float rv[3][16];
    auto load_tile = [&](int j) __attribute__((always_inline)) {
      unsigned o = ob0 + 128u * (unsigned)j;
      asm volatile("" : "+v"(o));
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        rv[j % 3][r] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(Rw) + o);
        o += ((r & 3) == 3) ? 5u * ldcb : ldcb;
      }
    };
    if (Rw) { load_tile(0); load_tile(1); }
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      if (Rw && j + 2 < 9) load_tile(j + 2);
      __builtin_amdgcn_sched_barrier(0);
      unsigned o = ob0 + 128u * (unsigned)j;
      asm volatile("" : "+v"(o));
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[j][r] + bvs[j];
        if (Rw) v += rv[j % 3][r];
        *reinterpret_cast<float*>(reinterpret_cast<char*>(Cw) + o) = v;
        o += ((r & 3) == 3) ? 5u * ldcb : ldcb;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    const float bv = (p.bias && !to_slab) ? p.bias[32 * j + li] : 0.f;
    unsigned o = (unsigned)(4 * lh) * ldc + li + 32 * j;
    float rv[16];
    if (Rw) {
      unsigned orr = o;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        rv[r] = (full || (r & 3) + 8 * (r >> 2) < mrem) ? Rw[orr] : 0.f;
        orr += ((r & 3) == 3) ? 5 * ldc : ldc;
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = acc[j][r] + bv;
      if (Rw) v += rv[r];
      if (full || (r & 3) + 8 * (r >> 2) < mrem) Cw[o] = v;
      o += ((r & 3) == 3) ? 5 * ldc : ldc;
    }
  }
}

// C (M x 288, row stride ldc) = sum of `splits` slabs (M x 288 each) + bias + residual
__global__ __launch_bounds__(256) void outres_splitk_reduce_kernel(const float* __restrict__ slab, int splits, int M,
                                                                    const float* __restrict__ bias,
                                                                    const float* __restrict__ residual,
                                                                    float* __restrict__ C, int64_t ldc) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;        // float4 index over M x 72
  if (i >= (int64_t)M * (OR_N / 4)) return;
  const int64_t m = i / (OR_N / 4);
  const int c4 = (int)(i - m * (OR_N / 4));
  float4 acc = *reinterpret_cast<const float4*>(slab + m * OR_N + 4 * c4);
  for (int sp = 1; sp < splits; ++sp) {
    const float4 v = *reinterpret_cast<const float4*>(slab + ((int64_t)sp * M + m) * OR_N + 4 * c4);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  if (bias) { const float4 v = *reinterpret_cast<const float4*>(bias + 4 * c4); acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
  if (residual) {
    const float4 v = *reinterpret_cast<const float4*>(residual + m * ldc + 4 * c4);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  *reinterpret_cast<float4*>(C + m * ldc + 4 * c4) = acc;
}

// Shape of a launch: 8-wave workgroups (256 rows) when they alone fill the chip, else 4-wave ones; when even those
// are fewer than ~224, K is cut into `splits` ranges of >= 24 pieces (768 contraction values) so that two
// workgroups per CU have work.  Returns the number of splits (1 = none), *nw the waves per workgroup, *kps the
// pieces per split.
extern "C" int pdn_gemm_outres_plan(int M, int K, int* nw, int* kps) {
  const int npieces = K / OR_KP, wg8 = (M + 255) / 256, wg4 = (M + 127) / 128;
  *nw = wg8 >= 224 ? 8 : 4;
  *kps = npieces;
  if (wg8 >= 224 || getenv("PDN_OUTRES_NO_SPLIT")) return 1;
  // fewer rows than fill the chip with 8-wave workgroups: K cut into ranges over grid.y, >= 24 pieces each.
  // 8-wave workgroups first (round 3: the lm_head input gradient at 16384 rows 86.4 % against 82.4 % for four ranges
  // of 4-wave workgroups), 4-wave ones where the contraction is too short to make ~256 of those.
  if (!getenv("PDN_OUTRES_NO_PLAN8")) {
    int s8 = (256 + wg8 - 1) / wg8;
    if (s8 > npieces / 24) s8 = npieces / 24;
    if (s8 >= 2 && wg8 * s8 >= 224) {
      *nw = 8;
      *kps = (npieces + s8 - 1) / s8;
      return (npieces + *kps - 1) / *kps;
    }
  }
  if (wg4 >= 224) return 1;
  int splits = (448 + wg4 - 1) / wg4;
  if (splits > npieces / 24) splits = npieces / 24;
  if (splits < 2) return 1;     // (64-row / 2-wave workgroups for short K measured 29 % against the tiled kernel's 42 %)
  *kps = (npieces + splits - 1) / splits;
  return (npieces + *kps - 1) / *kps;
}

// workgroup size when a planned K split cannot be used (no workspace): what fills the chip without it
static int outres_unsplit_nw(int M) { return (M + 255) / 256 >= 224 ? 8 : 4; }

extern "C" int64_t pdn_gemm_outres_workspace_bytes(int M, int K) {
  int nw, kps;
  const int splits = pdn_gemm_outres_plan(M, K, &nw, &kps);
  return splits > 1 ? (int64_t)splits * M * OR_N * 4 : 0;
}

extern "C" int pdn_gemm_outres_supported(int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int b_trans) {
  return N == OR_N && K >= OR_KP && K % OR_KP == 0 && M >= 1 && lda % 4 == 0 && ldb % 4 == 0 && lda >= K &&
         ldb >= (b_trans ? K : N) && ldc >= N && (int64_t)OR_N * ldb < (1ll << 30) && (int64_t)32 * ldc < (1ll << 30);
}

static int outres_launch(const float* A, const float* B, float* C, const float* bias, const float* residual, int M, int N,
                         int K, int64_t lda, int64_t ldb, int64_t ldc, int b_trans, void* workspace,
                         int64_t workspace_bytes, void* stream, int kb = 0, int64_t b_bstride = 0);

extern "C" int pdn_gemm_outres_f32(const float* A, const float* B, float* C, const float* bias,
                                   const float* residual, int M, int N, int K, int64_t lda, int64_t ldb,
                                   int64_t ldc, int b_trans, void* stream) {
  return outres_launch(A, B, C, bias, residual, M, N, K, lda, ldb, ldc, b_trans, nullptr, 0, stream);
}

// as pdn_gemm_outres_f32, K split over the grid when M is short and `workspace` holds
// pdn_gemm_outres_workspace_bytes(M, K) bytes (else unsplit)
extern "C" int pdn_gemm_outres_ws_f32(const float* A, const float* B, float* C, const float* bias,
                                      const float* residual, int M, int N, int K, int64_t lda, int64_t ldb,
                                      int64_t ldc, int b_trans, void* workspace, int64_t workspace_bytes, void* stream) {
  return outres_launch(A, B, C, bias, residual, M, N, K, lda, ldb, ldc, b_trans, workspace, workspace_bytes, stream);
}

int pdn_gemm_prof_begin(int family, double flops, double bytes, void* stream);    // csrc/gemm.hip
void pdn_gemm_prof_end(int token, void* stream);

// dX (M x 288) = [d_1 | ... | d_nb] (M x nb * kb) * [W_1 | ... | W_nb]^T (+ residual), W_i (288 x kb) row-major and
// `b_block_stride` floats apart -- the input gradient of projections that share their input (q | k | v, gate | up:
// llm/llama/model.py:93-104, 56-58; `grad @ W^T` of tensor.py:670 summed over the projections by ONE contraction)
// with the weights read where they live.  kb a multiple of 32; M large enough for unsplit 8- / 4-wave workgroups to
// fill the chip (pdn_gemm_outres_blocks_supported), else PDN_EUNSUPPORTED and the caller packs the weights.
extern "C" int pdn_gemm_outres_blocks_supported(int M, int kb, int nb) {
  const int K = kb * nb;
  const bool big = (M + 255) / 256 >= 224, mid = (M + 127) / 128 >= 224 && K >= 1536;
  return (kb > 0 && kb % OR_KP == 0 && nb >= 1 && (big || mid)) ? 1 : 0;
}
extern "C" int pdn_gemm_outres_blocks_nt_f32(const float* A, const float* W, int64_t b_block_stride, int kb, int nb, float* C,
                                             const float* residual, int M, int64_t lda, int64_t ldc, void* stream) {
  if (M == 0) return PDN_OK;
  const int K = kb * nb;
  if (!pdn_gemm_outres_blocks_supported(M, kb, nb) || (b_block_stride & 3) || (kb & 3) ||
      !pdn_gemm_outres_supported(M, OR_N, K, lda, K, ldc, 1)) {        // (row stride of a block = kb: checked here)
    pdn_set_error("pdn_gemm_outres_blocks_nt_f32: unsupported shape M=%d kb=%d nb=%d", M, kb, nb);
    return PDN_EUNSUPPORTED;
  }
  const int tk = pdn_gemm_prof_begin(3, 2.0 * M * (double)OR_N * K, 0.0, stream);
  const int rc = outres_launch(A, W, C, nullptr, residual, M, OR_N, K, lda, kb, ldc, 1, nullptr, 0, stream, kb, b_block_stride);
  pdn_gemm_prof_end(tk, stream);
  return rc;
}

static int outres_launch(const float* A, const float* B, float* C, const float* bias, const float* residual, int M, int N,
                         int K, int64_t lda, int64_t ldb, int64_t ldc, int b_trans, void* workspace,
                         int64_t workspace_bytes, void* stream, int kb, int64_t b_bstride) {
  if (M == 0 || N == 0) return PDN_OK;
  PDN_CHECK_ARG(A && B && C, "pdn_gemm_outres_f32: null operand");
  if (!pdn_gemm_outres_supported(M, N, K, lda, kb > 0 ? K : ldb, ldc, b_trans)) {
    pdn_set_error("pdn_gemm_outres_f32: unsupported shape M=%d N=%d K=%d (N 288, K a multiple of 32, leading "
                  "dimensions multiples of 4)", M, N, K);
    return PDN_EUNSUPPORTED;
  }
  PDN_CHECK_ARG(((((uintptr_t)A | (uintptr_t)B) & 15) == 0), "pdn_gemm_outres_f32: 16-byte alignment required");
  OutResParams p{A, B, C, bias, residual, M, K, lda, ldb, ldc, nullptr, nullptr, nullptr, 0.f, K / OR_KP, nullptr};
  static const int s_pdn_outres_rt_ablate = pdn_ablation_switch("PDN_OUTRES_RT_ABLATE");
  p.ablate_rt = s_pdn_outres_rt_ablate;
  if (kb > 0) {
    p.ppb = kb / OR_KP;
    p.ppb_magic = (unsigned)(((1ull << 32) + (unsigned)p.ppb - 1) / (unsigned)p.ppb);
    p.b_bstride = b_bstride;
  }
  hipStream_t st = (hipStream_t)stream;
  int plan_nw = 8, kps = K / OR_KP;
  int splits = pdn_gemm_outres_plan(M, K, &plan_nw, &kps);
  if (splits > 1 && (!workspace || workspace_bytes < (int64_t)splits * M * OR_N * 4 || (ldc & 3) ||
                     ((uintptr_t)C & 15) || ((uintptr_t)workspace & 15) || ((uintptr_t)bias & 15) || ((uintptr_t)residual & 15))) {
    splits = 1;
    plan_nw = outres_unsplit_nw(M);
  }
  if (splits > 1) { p.kps = kps; p.slab = (float*)workspace; }
  static const int nw_env = getenv("PDN_OUTRES_NW") ? atoi(getenv("PDN_OUTRES_NW")) : 0;
  static const int stage_env = getenv("PDN_OUTRES_STAGE") ? atoi(getenv("PDN_OUTRES_STAGE")) : -1;
  static const int ablate = pdn_ablation_switch("PDN_OUTRES_ABLATE");
  // one 8-wave workgroup per CU when that fills the chip, else 4-wave workgroups
  const int nw = nw_env ? nw_env : plan_nw;
  const int stage = stage_env >= 0 ? stage_env : 1;
  const dim3 grid((M + 32 * nw - 1) / (32 * nw), splits);
  pdn_count(PDN_CNT_OUTRES);
#define OR_LAUNCH(BT_, NW_, ST_, AB_) hipLaunchKernelGGL((gemm_outres_kernel<BT_, NW_, ST_, AB_>), grid, dim3(NW_ * 64), 0, st, p)
  if (ablate == 1 && b_trans) OR_LAUNCH(true, 4, 1, 1);
  else if (ablate == 2 && b_trans) OR_LAUNCH(true, 4, 1, 2);
  else if (nw == 8 && b_trans) { if (stage) OR_LAUNCH(true, 8, 1, 0); else OR_LAUNCH(true, 8, 0, 0); }
  else if (nw == 8) { if (stage) OR_LAUNCH(false, 8, 1, 0); else OR_LAUNCH(false, 8, 0, 0); }
  else if (b_trans) { if (stage) OR_LAUNCH(true, 4, 1, 0); else OR_LAUNCH(true, 4, 0, 0); }
  else { if (stage) OR_LAUNCH(false, 4, 1, 0); else OR_LAUNCH(false, 4, 0, 0); }
#undef OR_LAUNCH
  PDN_LAUNCH_CHECK();
  if (splits > 1) {
    const int64_t n4 = (int64_t)M * (OR_N / 4);
    hipLaunchKernelGGL(outres_splitk_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, p.slab, splits, M,
                       bias, residual, C, ldc);
    PDN_LAUNCH_CHECK();
  }
  return PDN_OK;
}

// dX (M x 288) = dlogits (M x V) * W^T, W (288 x V) row-major, with dlogits formed from the logits on the fly
// (see OutResParams): the input gradient of `linear -> cross entropy` without the (M x V) gradient in memory.
int pdn_outres_ce_dx_launch(const float* logits, int64_t ldl, const float* lse, const int64_t* targets, float gscale,
                            const float* gdev, const float* W, int64_t ldw, float* dx, int64_t ldc,
                            const float* residual, int M, int V, void* workspace, int64_t workspace_bytes, void* stream) {
  OutResParams p{logits, W, dx, nullptr, residual, M, V, ldl, ldw, ldc, lse, targets, gdev, gscale, V / OR_KP, nullptr};
  int nw = 8, kps = V / OR_KP;
  int splits = pdn_gemm_outres_plan(M, V, &nw, &kps);
  if (splits > 1 && (!workspace || workspace_bytes < (int64_t)splits * M * OR_N * 4 || (ldc & 3) || ((uintptr_t)dx & 15) ||
                     ((uintptr_t)workspace & 15) || ((uintptr_t)residual & 15))) {
    splits = 1;
    nw = outres_unsplit_nw(M);
  }
  if (splits > 1) { p.kps = kps; p.slab = (float*)workspace; }
  const dim3 grid((M + 32 * nw - 1) / (32 * nw), splits);
  if (nw == 8) hipLaunchKernelGGL((gemm_outres_kernel<true, 8, 1, 0, 1>), grid, dim3(512), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((gemm_outres_kernel<true, 4, 1, 0, 1>), grid, dim3(256), 0, (hipStream_t)stream, p);
  PDN_LAUNCH_CHECK();
  if (splits > 1) {
    const int64_t n4 = (int64_t)M * (OR_N / 4);
    hipLaunchKernelGGL(outres_splitk_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       p.slab, splits, M, (const float*)nullptr, residual, dx, ldc);
    PDN_LAUNCH_CHECK();
  }
  return PDN_OK;
}

// The same product with the softmax denominator found on the way (CE == 2, see OutResParams): row maxima in, lse out.
// Rows that fill the chip: every row's whole vocabulary in one workgroup, normalised in its store.  Fewer rows: the
// vocabulary is cut into ranges over grid.y as for the plain product; the ranges leave unnormalised rows and their row
// sums, and this kernel adds them up:  dx = gscale * (sum_s acc_s / sum_s Z_s - W[:, target]),  lse = max + log sum_s Z_s.
__global__ __launch_bounds__(256) void outres_ce2_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ zslab,
                                                                int splits, int M, const float* __restrict__ rowmax,
                                                                int max_parts, const int64_t* __restrict__ targets,
                                                                const float* __restrict__ W, int64_t ldw, int V, float gscale,
                                                                float* __restrict__ dx, int64_t ldc, float* __restrict__ lse_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // one 16-byte piece of a row
  if (i >= (int64_t)M * (OR_N / 4)) return;
  const int row = (int)(i / (OR_N / 4)), c4 = (int)(i - (int64_t)row * (OR_N / 4));
  float4 a = *reinterpret_cast<const float4*>(slab + (int64_t)row * OR_N + 4 * c4);
  float z = zslab[row];
  for (int s = 1; s < splits; ++s) {
    const float4 b = *reinterpret_cast<const float4*>(slab + ((int64_t)s * M + row) * OR_N + 4 * c4);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    z += zslab[(int64_t)s * M + row];
  }
  const int t = min(max((int)targets[row], 0), V - 1);
  const float iz = 1.f / z;
  const float4 wt = *reinterpret_cast<const float4*>(W + (int64_t)t * OR_N + 4 * c4);      // W = the transposed copy (V x 288)
  float4 o;
  o.x = gscale * (a.x * iz - wt.x); o.y = gscale * (a.y * iz - wt.y);
  o.z = gscale * (a.z * iz - wt.z); o.w = gscale * (a.w * iz - wt.w);
  *reinterpret_cast<float4*>(dx + (int64_t)row * ldc + 4 * c4) = o;
  if (c4 == 0) {
    float m = rowmax[row];
    for (int q = 1; q < max_parts; ++q) m = fmaxf(m, rowmax[(int64_t)q * M + row]);
    lse_out[row] = m + __logf(z);
  }
}

// Wt (V x 288) = W^T for W (288 x V, leading dimension ldw): 32 x 32 tiles through LDS, both sides coalesced
__global__ __launch_bounds__(256) void outres_wt_transpose_kernel(const float* __restrict__ W, int64_t ldw, int V,
                                                                  float* __restrict__ Wt) {
  __shared__ float tile[32][33];
  const int v0 = blockIdx.x * 32, k0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int r = ty; r < 32; r += 8) tile[r][tx] = (v0 + tx < V) ? W[(int64_t)(k0 + r) * ldw + v0 + tx] : 0.f;
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8)
    if (v0 + r < V) Wt[(int64_t)(v0 + r) * OR_N + k0 + tx] = tile[tx][r];
}

extern "C" int pdn_linear_ce_dx_deferred_supported(int64_t M, int V, int K) {
  return (K == OR_N && V % OR_KP == 0 && V >= OR_KP && M >= 1 && M < (1ll << 31) && (int64_t)OR_N * V < (1ll << 30)) ? 1 : 0;
}
// workspace of the split form: [splits x M x 288 rows | splits x M row sums]; 0 when the rows fill the chip
extern "C" int64_t pdn_linear_ce_dx_deferred_workspace_bytes(int64_t M, int V, int K) {
  if (!pdn_linear_ce_dx_deferred_supported(M, V, K)) return 0;
  int nw, kps;
  const int splits = pdn_gemm_outres_plan((int)M, V, &nw, &kps);
  // [W^T copy: V x 288] then, when the vocabulary is cut into ranges, [splits x M x 288 rows | splits x M row sums]
  return (int64_t)V * OR_N * 4 + (splits > 1 ? (int64_t)splits * M * (OR_N + 1) * 4 : 0);
}
int pdn_outres_ce_dx_deferred_launch(const float* logits, int64_t ldl, const float* rowmax, int max_parts,
                                     const int64_t* targets, float gscale, const float* W, int64_t ldw, float* dx,
                                     int64_t ldc, float* lse_out, int M, int V, void* workspace, int64_t workspace_bytes,
                                     void* stream) {
  OutResParams p{logits, W, dx, nullptr, nullptr, M, V, ldl, ldw, ldc, rowmax, targets, nullptr, gscale, V / OR_KP, nullptr};
  p.lse_out = lse_out;
  p.max_parts = max_parts;
  int nw = 8, kps = V / OR_KP;
  const int splits = pdn_gemm_outres_plan(M, V, &nw, &kps);
  const int64_t wt_floats = (int64_t)V * OR_N;
  PDN_CHECK_ARG(workspace && workspace_bytes >= wt_floats * 4 + (splits > 1 ? (int64_t)splits * M * (OR_N + 1) * 4 : 0) &&
                    (((uintptr_t)workspace | (uintptr_t)dx) & 15) == 0 && (ldc & 3) == 0,
                "pdn_linear_ce_dx_deferred_f32: workspace too small or misaligned");
  float* Wt = (float*)workspace;
  p.Wt = Wt;
  if (splits > 1) {
    p.kps = kps;
    p.slab = Wt + wt_floats;
    p.zslab = p.slab + (int64_t)splits * M * OR_N;
  }
  const dim3 grid((M + 32 * nw - 1) / (32 * nw), splits);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(outres_wt_transpose_kernel, dim3((V + 31) / 32, OR_N / 32), dim3(256), 0, st, W, ldw, V, Wt);
  PDN_LAUNCH_CHECK();
  pdn_count(PDN_CNT_CE_DX_DEFERRED);
  if (nw == 8) hipLaunchKernelGGL((gemm_outres_kernel<true, 8, 1, 0, 2>), grid, dim3(512), 0, st, p);
  else hipLaunchKernelGGL((gemm_outres_kernel<true, 4, 1, 0, 2>), grid, dim3(256), 0, st, p);
  PDN_LAUNCH_CHECK();
  if (splits > 1) {
    const int64_t n4 = (int64_t)M * (OR_N / 4);
    hipLaunchKernelGGL(outres_ce2_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, p.slab, p.zslab, splits, M,
                       rowmax, max_parts, targets, Wt, ldw, V, gscale, dx, ldc, lse_out);
    PDN_LAUNCH_CHECK();
  }
  return PDN_OK;
}

// ======================================================================================
// Weight-gradient form with the model width as the OUTPUT ROWS:  C (288 x N) = X^T (288 x K) * G (K x N),
// X (K x 288) and G (K x N) row-major, K = tokens.  (`A^T @ grad`, tensor.py:672-675: the lm_head weight
// gradient x^T dlogits with N = 32000.)  The mirror image of the kernel above: a wave owns 32 COLUMNS of C and
// all 288 rows of them (nine accumulator tiles); G -- the big operand -- is read exactly once, straight from
// global memory (four 16-byte loads per lane and k-piece, eight lanes per 128-byte row segment) through a
// 4 KiB LDS piece PRIVATE to the wave, from which the MFMA B operands are read with conflict-free ds_read_b32
// (loading the operands directly -- sixteen dword loads per piece -- measured 80 instead of 9x % of the matrix
// peak: a memory instruction costs the wave's MFMA stream the same whatever its width); X is streamed through LDS
// in 32 x 288 pieces, register-staged, shared by the workgroup, and read back as A-operand fragments.  K is split over grid.y when the columns alone do not fill the chip; the slabs are
// combined by gemm_splitk_reduce_kernel (csrc/gemm.hip), which also applies beta.
// ======================================================================================
struct OutResTnParams {
  const float* X;
  const float* G;
  float* C;                       // slab s at C + s * slab
  int N, K;
  int64_t ldx, ldg, ldc, slab;
  int k_per_split;
  // CE instantiation: G holds LOGITS (rows = tokens); the operand is (exp(logit - lse[t]) - [n == target[t]]) * scale,
  // formed as it is read from the wave's LDS piece; lse / targets of a piece's 32 tokens are staged beside X.
  // colsum (optional): per-split column sums of the formed operand (the bias gradient), [splits][N]
  const float* lse;
  const int64_t* targets;
  const float* gdev;
  float gscale;
  float* colsum;
  // output in column blocks of `nb_cols` (x^T against dq | dk | dv: one weight gradient per block): block b of split s
  // at C + b * blk_stride + s * slab, rows of `ldc` floats; nb_cols = N for a single matrix
  int nb_cols;
  int64_t blk_stride;
  int xcd_swizzle;                // 1: K ranges dealt to XCDs (the number of K ranges is a multiple of 8)
};

// WIDE: the X fragments of FOUR row tiles in one ds_read_b128.  A lane of the MFMA A operand supplies one row of the
// output tile; WHICH row of C that is, is free: with row 128 g + 4 l + c given to lane l of tile 4 g + c, the four values a
// lane needs for tiles 4 g .. 4 g + 3 at one k are adjacent in the X piece.  Rows 256 .. 287 (tile 8) keep the dword reads:
// 3 LDS instructions per k step instead of 9; the permutation is undone by the addresses of the final stores.
template <int NW, bool CE = false, bool WIDE = false>
__global__ __launch_bounds__(NW * 64, 1) void gemm_outres_tn_kernel(OutResTnParams p) {
  constexpr int PIECE = OR_KP * OR_N;
  constexpr int NQ = (36 + NW - 1) / NW;
  constexpr int GP = OR_KP * 32;                   // a wave's private 32 (k) x 32 (n) piece of G: 4 KiB
  extern __shared__ __attribute__((aligned(16))) float smem[];   // 2 X pieces, then NW x 2 G pieces
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  // Workgroup -> (column block bx, K range by).  Hardware deals consecutive workgroups to the 8 XCDs round-robin: in
  // launch order the column blocks that read the SAME x rows (one K range) land on different XCDs and every private
  // L2 fetches those rows again (counters of the packed layer weight gradients, round 3: 1.77 x the operand bytes,
  // L2 hit rate 0.10).  Re-dealt here so that the column blocks of one K range are consecutive on ONE XCD.
  int bx = blockIdx.x, by = blockIdx.y;
  if (p.xcd_swizzle) {
    const int L = blockIdx.y * gridDim.x + blockIdx.x, xcd = L & 7, slot = L >> 3;
    bx = slot % (int)gridDim.x;
    by = (slot / (int)gridDim.x) * 8 + xcd;
  }
  const int n0_raw = (bx * NW + wave) * 32;
  const bool active = n0_raw < p.N;
  const int n0 = active ? n0_raw : p.N - 32;          // idle waves still stage X and meet the barriers
  const int k_begin = by * p.k_per_split;
  const int k_end = min(p.K, k_begin + p.k_per_split);
  const int npieces = (k_end - k_begin) / OR_KP;
  const unsigned ldx = (unsigned)p.ldx, ldg = (unsigned)p.ldg;
  const float* Xk = p.X + (int64_t)k_begin * p.ldx;
  float* Gs = smem + 2 * PIECE + wave * (2 * GP);
  float* Ls = smem + 2 * PIECE + NW * 2 * GP;      // CE: [2][32] lse, then [2][32] targets (as int bits)
  const float ce_sc = CE ? p.gscale * (p.gdev ? p.gdev[0] : 1.f) : 0.f;
  float ce_row = 0.f;                               // CE, wave 0: lanes 0..31 carry lse[t], lanes 32..63 target[t]
  float csum = 0.f;

  float4 rb[NQ], rg[4];
  unsigned xo[NQ];                                  // lane offsets of the X staging, once (they were ~10 VALU operations per fetch)
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int u = 64 * min(q * NW + wave, 35) + lane, k = u / 72, n4 = u - 72 * k;
    xo[q] = (unsigned)k * ldx + 4u * (unsigned)n4;
  }
  auto fetch = [&](int piece, int q) {
    const float4 v = *reinterpret_cast<const float4*>(Xk + (int64_t)piece * OR_KP * p.ldx + xo[q]);
    rb[q].x = v.x; rb[q].y = v.y; rb[q].z = v.z; rb[q].w = v.w;
  };
  // G piece: instruction e covers rows 8e .. 8e + 7, eight lanes (128 bytes) per row; LDS image [32 k][32 n] linear
  const float* gsrc = p.G + (int64_t)(k_begin + (lane >> 3)) * p.ldg + n0 + 4 * (lane & 7);
  auto fetch_g = [&](int piece) {
    const float* g = gsrc + (int64_t)piece * OR_KP * p.ldg;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float4 v = *reinterpret_cast<const float4*>(g + (unsigned)(8 * e) * ldg);
      rg[e].x = v.x; rg[e].y = v.y; rg[e].z = v.z; rg[e].w = v.w;
    }
  };
  auto fetch_rows = [&](int piece) {                // CE: the piece's 32 tokens
    if (CE && wave == 0) {
      const int t = k_begin + piece * OR_KP + li;
      ce_row = lh == 0 ? -1.4426950408889634f * p.lse[t] : __int_as_float((int)p.targets[t]);   // (-lse log2 e: one fma + exp2 per logit)
    }
  };
  auto park = [&](int buf) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int I = min(q * NW + wave, 35);
      *reinterpret_cast<float4*>(smem + buf * PIECE + I * 256 + 4 * lane) = rb[q];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) *reinterpret_cast<float4*>(Gs + buf * GP + e * 256 + 4 * lane) = rg[e];
    if (CE && wave == 0) Ls[(lh * 2 + buf) * 32 + li] = ce_row;
  };
  if (npieces > 0) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) fetch(0, q);
    fetch_g(0);
    fetch_rows(0);
    park(0);
  }

  f32x16 acc[9];
#pragma unroll
  for (int j = 0; j < 9; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const int bnn = (4 * lh) * OR_N + li;
  const int gnn = (4 * lh) * 32 + li;

  for (int s = 0; s < npieces; ++s) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const float* Bp = smem + (s & 1) * PIECE;
    const float* Gp = Gs + (s & 1) * GP + gnn;
    const int nxt = min(s + 1, npieces - 1);
    if constexpr (WIDE) {
      const float* Bw = Bp + (4 * lh) * OR_N + 4 * li;
      const float* Bt = Bp + (4 * lh) * OR_N + 256 + li;
      float4 wa[2][2];
      float wt[2];
      auto loadx = [&](int slot, int kk) {            // kk = 8 g + q: rows kk (h = 0) and kk + 4 (h = 1) of the piece
        wa[slot][0] = *reinterpret_cast<const float4*>(Bw + kk * OR_N);
        wa[slot][1] = *reinterpret_cast<const float4*>(Bw + kk * OR_N + 128);
        wt[slot] = Bt[kk * OR_N];
      };
      loadx(0, 0);
      float gn[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) gn[q] = Gp[q * 32];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float gv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) gv[q] = gn[q];
        if (CE) {
          const float4 l4 = *reinterpret_cast<const float4*>(Ls + (s & 1) * 32 + 8 * g + 4 * lh);
          const float4 t4 = *reinterpret_cast<const float4*>(Ls + (2 + (s & 1)) * 32 + 8 * g + 4 * lh);
          const int col = n0 + li;
          gv[0] = ce_unscaled(gv[0], l4.x, __float_as_int(t4.x) == col);
          gv[1] = ce_unscaled(gv[1], l4.y, __float_as_int(t4.y) == col);
          gv[2] = ce_unscaled(gv[2], l4.z, __float_as_int(t4.z) == col);
          gv[3] = ce_unscaled(gv[3], l4.w, __float_as_int(t4.w) == col);
          csum += (gv[0] + gv[1]) + (gv[2] + gv[3]);
        }
        if (g + 1 < 4) {
#pragma unroll
          for (int q = 0; q < 4; ++q) gn[q] = Gp[(8 * (g + 1) + q) * 32];
        }
        if (g == 0) { fetch_g(nxt); fetch_rows(nxt); }
        if (g < 3) {
#pragma unroll
          for (int e = 0; e < 3; ++e)
            if (3 * g + e < NQ) fetch(nxt, 3 * g + e);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int cur = q & 1;
          if (4 * g + q + 1 < 16) loadx(cur ^ 1, 8 * ((4 * g + q + 1) >> 2) + ((4 * g + q + 1) & 3));
          __builtin_amdgcn_sched_barrier(0);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[cur][0].x, gv[q], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[cur][0].y, gv[q], acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[cur][0].z, gv[q], acc[2], 0, 0, 0);
          acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[cur][0].w, gv[q], acc[3], 0, 0, 0);
          acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[cur][1].x, gv[q], acc[4], 0, 0, 0);
          acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[cur][1].y, gv[q], acc[5], 0, 0, 0);
          acc[6] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[cur][1].z, gv[q], acc[6], 0, 0, 0);
          acc[7] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[cur][1].w, gv[q], acc[7], 0, 0, 0);
          acc[8] = __builtin_amdgcn_mfma_f32_32x32x2f32(wt[cur], gv[q], acc[8], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
    float x0[3][4], x1[3][4];
#define TN_LOADX(BX, G, T)                                                                       \
  _Pragma("unroll") for (int j = 0; j < 3; ++j)                                                  \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) BX[j][q] = Bp[bnn + (8 * (G) + q) * OR_N + 32 * (3 * (T) + j)];
#define TN_MFMA(BX, GV, T)                                                                       \
  {                                                                                              \
    _Pragma("unroll") for (int j = 0; j < 3; ++j) acc[3 * (T) + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(BX[j][0], GV[0], acc[3 * (T) + j], 0, 0, 0); \
    _Pragma("unroll") for (int j = 0; j < 3; ++j) acc[3 * (T) + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(BX[j][1], GV[1], acc[3 * (T) + j], 0, 0, 0); \
    _Pragma("unroll") for (int j = 0; j < 3; ++j) acc[3 * (T) + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(BX[j][2], GV[2], acc[3 * (T) + j], 0, 0, 0); \
    _Pragma("unroll") for (int j = 0; j < 3; ++j) acc[3 * (T) + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(BX[j][3], GV[3], acc[3 * (T) + j], 0, 0, 0); \
  }
#define TN_STEP(BC, BN, G, T)                                                                    \
  if (3 * (G) + (T) + 1 < 12) { TN_LOADX(BN, (3 * (G) + (T) + 1) / 3, (3 * (G) + (T) + 1) % 3) }  \
  __builtin_amdgcn_sched_barrier(0);                                                             \
  TN_MFMA(BC, gv, T)                                                                             \
  __builtin_amdgcn_sched_barrier(0);
    TN_LOADX(x0, 0, 0)
    float gn[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) gn[q] = Gp[q * 32];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float gv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) gv[q] = gn[q];
      if (CE) {                                     // rows 8 g + 4 h + q of the piece, column n0 + lane
        const float4 l4 = *reinterpret_cast<const float4*>(Ls + (s & 1) * 32 + 8 * g + 4 * lh);
        const float4 t4 = *reinterpret_cast<const float4*>(Ls + (2 + (s & 1)) * 32 + 8 * g + 4 * lh);
        const int col = n0 + li;
        gv[0] = ce_unscaled(gv[0], l4.x, __float_as_int(t4.x) == col);
        gv[1] = ce_unscaled(gv[1], l4.y, __float_as_int(t4.y) == col);
        gv[2] = ce_unscaled(gv[2], l4.z, __float_as_int(t4.z) == col);
        gv[3] = ce_unscaled(gv[3], l4.w, __float_as_int(t4.w) == col);
        csum += (gv[0] + gv[1]) + (gv[2] + gv[3]);
      }
      if (g + 1 < 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) gn[q] = Gp[(8 * (g + 1) + q) * 32];
      }
      if (g == 0) { fetch_g(nxt); fetch_rows(nxt); }
      if (g < 3) {
#pragma unroll
        for (int e = 0; e < 3; ++e)
          if (3 * g + e < NQ) fetch(nxt, 3 * g + e);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (g & 1) {
        TN_STEP(x1, x0, g, 0) TN_STEP(x0, x1, g, 1) TN_STEP(x1, x0, g, 2)
      } else {
        TN_STEP(x0, x1, g, 0) TN_STEP(x1, x0, g, 1) TN_STEP(x0, x1, g, 2)
      }
    }
#undef TN_STEP
#undef TN_LOADX
#undef TN_MFMA
    }
    park((s + 1) & 1);
  }
  if (!active) return;
  if (CE) {                                         // the operand was formed without its scale (ce_unscaled)
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] *= ce_sc;
    csum *= ce_sc;
  }
  if (CE && p.colsum) {                             // both half-waves saw disjoint token rows
    csum += __shfl_xor(csum, 32, 64);
    if (lh == 0) p.colsum[(int64_t)by * p.N + n0 + li] = csum;
  }
  // accumulator register r of tile i = row 32 i + (r & 3) + 8 (r >> 2) + 4 h, column n0 + lane
  // (WIDE, tiles 0 .. 7: row 128 (i >> 2) + 4 ((r & 3) + 8 (r >> 2) + 4 h) + (i & 3), see above)
  const int blk = n0 / p.nb_cols;                   // (a wave's 32 columns never straddle two blocks: nb_cols % 32 == 0)
  float* __restrict__ Cw = p.C + blk * p.blk_stride + (int64_t)by * p.slab + (n0 - blk * p.nb_cols) + li;
  const unsigned ldc = (unsigned)p.ldc;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    if (WIDE && i < 8) {
      const unsigned o = (unsigned)(128 * (i >> 2) + (i & 3) + 16 * lh) * ldc;
#pragma unroll
      for (int r = 0; r < 16; ++r) Cw[o + (unsigned)(4 * ((r & 3) + 8 * (r >> 2))) * ldc] = acc[i][r];
    } else {
      unsigned o = (unsigned)(32 * i + 4 * lh) * ldc;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        Cw[o] = acc[i][r];
        o += ((r & 3) == 3) ? 5 * ldc : ldc;
      }
    }
  }
}

// splits of K so that (column workgroups x splits) fills the chip once; k per split a multiple of 32
int pdn_gemm_outres_tn_plan(int N, int K, int* nw_out, int* k_per_split_out) {
  static const int nw_env = getenv("PDN_OUTRES_NW") ? atoi(getenv("PDN_OUTRES_NW")) : 0;
  const int nw = nw_env ? nw_env : 8;
  const int col_wgs = (N / 32 + nw - 1) / nw;
  int splits = (nw == 8 ? 256 : 512) / (col_wgs > 0 ? col_wgs : 1);
  if (splits < 1) splits = 1;
  if (splits >= 16 && col_wgs > 1) splits &= ~7;     // whole octets of K ranges: one per XCD (see the kernel's block mapping)
  const int pieces = K / OR_KP;
  if (splits > pieces) splits = pieces > 0 ? pieces : 1;
  const int kps = ((pieces + splits - 1) / splits) * OR_KP;
  if (nw_out) *nw_out = nw;
  if (k_per_split_out) *k_per_split_out = kps;
  return (K + kps - 1) / kps;
}

// C slabs: slab s (rows of `ldc` floats) at C + s * slab receives the partial product of split s
static int outres_tn_launch(OutResTnParams& p, int nw, bool ce, void* stream) {
  const int splits = (p.K + p.k_per_split - 1) / p.k_per_split;
  const dim3 grid((p.N / 32 + nw - 1) / nw, splits);
  static const int swz_env = getenv("PDN_OUTRES_TN_SWIZZLE") ? atoi(getenv("PDN_OUTRES_TN_SWIZZLE")) : 1;
  p.xcd_swizzle = (swz_env && splits % 8 == 0 && grid.x > 1) ? 1 : 0;
  const size_t shm = (size_t)(2 * OR_KP * OR_N + nw * 2 * OR_KP * 32 + 4 * 32) * sizeof(float);
  static const int wide = getenv("PDN_OUTRES_TN_WIDE") ? atoi(getenv("PDN_OUTRES_TN_WIDE")) : 1;
  static bool attr_set = false;
#define TN_EACH(X) X(8, false, false) X(4, false, false) X(8, true, false) X(4, true, false) X(8, false, true) X(4, false, true) X(8, true, true) X(4, true, true)
  if (!attr_set) {
#define TN_ATTR(NW_, CE_, W_) \
    PDN_HIP(hipFuncSetAttribute((const void*)gemm_outres_tn_kernel<NW_, CE_, W_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    TN_EACH(TN_ATTR)
#undef TN_ATTR
    attr_set = true;
  }
  hipStream_t st = (hipStream_t)stream;
#define TN_GO(NW_, CE_, W_) \
  if (nw == NW_ && ce == CE_ && (wide != 0) == W_) hipLaunchKernelGGL((gemm_outres_tn_kernel<NW_, CE_, W_>), grid, dim3(NW_ * 64), shm, st, p);
  TN_EACH(TN_GO)
#undef TN_GO
#undef TN_EACH
  pdn_count(ce ? PDN_CNT_CE_DW : PDN_CNT_OUTRES_TN);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

// C slabs: slab s (rows of `ldc` floats) at C + s * slab receives the partial product of split s
int pdn_gemm_outres_tn_launch(const float* X, const float* G, float* C, int N, int K, int64_t ldx, int64_t ldg,
                              int64_t ldc, int64_t slab, int nw, int k_per_split, void* stream) {
  OutResTnParams p{X, G, C, N, K, ldx, ldg, ldc, slab, k_per_split, nullptr, nullptr, nullptr, 0.f, nullptr, N, 0};
  return outres_tn_launch(p, nw, false, stream);
}

// G (K x N) as `N / nb_cols` column blocks whose products go to separate outputs: slab s of block b at
// C + (b * splits + s) * 288 * nb_cols, rows of nb_cols floats (the batched layout of gemm_splitk_reduce_kernel)
int pdn_gemm_outres_tn_blocks_launch(const float* X, const float* G, float* C, int N, int K, int64_t ldx, int64_t ldg,
                                     int nb_cols, int nw, int k_per_split, void* stream) {
  const int splits = (K + k_per_split - 1) / k_per_split;
  OutResTnParams p{X, G, C, N, K, ldx, ldg, nb_cols, (int64_t)288 * nb_cols, k_per_split, nullptr, nullptr, nullptr, 0.f,
                   nullptr, nb_cols, (int64_t)splits * 288 * nb_cols};
  return outres_tn_launch(p, nw, false, stream);
}

// the same with G = logits turned into the cross-entropy gradient on the fly; colsum: [splits][N] or null
int pdn_outres_ce_dw_launch(const float* X, const float* logits, float* C, int N, int K, int64_t ldx, int64_t ldg,
                            int64_t ldc, int64_t slab, int nw, int k_per_split, const float* lse,
                            const int64_t* targets, float gscale, const float* gdev, float* colsum, void* stream) {
  OutResTnParams p{X, logits, C, N, K, ldx, ldg, ldc, slab, k_per_split, lse, targets, gdev, gscale, colsum, N, 0};
  return outres_tn_launch(p, nw, true, stream);
}
