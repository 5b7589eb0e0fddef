// Tile-piece row-resident fp32 GEMM (gfx950 only, round 5): the projections with a contraction of exactly the model
// width (288) whose epilogue work used to be EXPOSED in the chunk kernel of csrc/gemm_rowres.hip.
//
//   C (M x N) = A (M x 288, rows contiguous) * B (288 x N)   (+ bias[N])  + the fused epilogues of gemm_rowres.hip
//
// Replaces `x @ W` / `grad @ W^T` of pydynet/core/tensor.py:657-676 for x Wq|Wk|Wv (+ RoPE, llm/llama/model.py:23-44,
// 93-104), x Wg|Wu (+ SwiGLU, model.py:56-58), dy W_down^T (+ SwiGLU backward) and h W_out (+ the row maxima of
// nn/functional.py:364-381).
//
// What round 4's profile said about the chunk kernel (one 32 x 96 accumulator block per wave and 96 x 96 B pieces):
// every wave of the chip finishes its chunk at the same moment, so (i) the stores of a chunk leave as one chip-wide
// burst (25-37 MB) and the next piece's staging loads -- vmcnt retires in order -- wait behind the acknowledgement of
// every one of them (tools/rowres_ablate.py: 32-62 us per launch with nothing else changed), (ii) whatever the store
// phase has to READ first (the saved gate | up rows of the SwiGLU backward, the RoPE table) is a chain of exposed
// round trips: 160 us of a 397 us launch for the SwiGLU backward.
//
// Here a piece of B is ONE 32-column tile over the WHOLE contraction (288 x 32 floats = 36 KiB, double buffered):
//   * a wave still owns 32 rows of A in 144 VGPRs (loaded once) and issues 144 MFMAs per piece between two barriers,
//     but a finished 32 x 32 accumulator tile is only 16 registers, so the accumulators ROTATE (two sets; three for the
//     gate / up pairs of SwiGLU): while piece t + 1 multiplies, tile t leaves -- one row step (one or two 128-byte row
//     segments per half-wave) every second k-group, i.e. the stores of the whole chip are spread evenly over time;
//   * what a row step has to read (saved gate / up values, RoPE factors, the bias) is requested 16-20 k-groups (3-4 us)
//     ahead into an eight-row ring of registers, behind which nothing waits;
//   * the staging loads of the next piece are consumed 16 k-groups after they were issued, so the in-order counter
//     never makes the MFMA stream wait for a store acknowledgement;
//   * nothing in the steady-state loop is conditional (the first drain of a workgroup writes the not-yet-computed
//     registers to the place of its LAST tile, which the final drain overwrites): the compiler's wait counts stay exact.
// The k-order of every output element is that of csrc/gemm.hip and csrc/gemm_rowres.hip (k = 8g + 4h + q inside a
// k-group, groups ascending): results are bit-identical to both.
#include "common.h"
#include "gemm_rowtile.h"
#include <stdlib.h>
#include <type_traits>

// Timing-ablation switches (tools/*_probe.py): read ONCE per process; a non-zero value makes kernels skip work and return
// WRONG results, so it is announced on stderr instead of taking effect silently.
static int pdn_ablation_switch(const char* name) {
  const char* e = getenv(name);
  const int v = e ? atoi(e) : 0;
  if (v) fprintf(stderr, "[pdnhip] WARNING: %s=%d -- timing ablation active, results of the affected kernels are WRONG\n", name, v);
  return v;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define RT_KG 36                  // k-groups of 8 (K = 288)
#define RT_PIECE (288 * 32)       // floats per piece

struct RowTileParams {
  const float* A;
  const float* B;
  float* C;
  const float* bias;              // always a readable address (B when there is no bias: has_bias = 0)
  const float* residual;          // EPI 6: (M x N), leading dimension ldc, added in the drain
  int M, N;
  int64_t lda, ldb, ldc;
  int pieces, ppw;                // pieces in total / per workgroup (grid.y)
  int tpb;                        // pieces per column block of B
  unsigned tpb_magic;             // ceil(2^32 / tpb)
  int64_t b_bstride;
  int has_bias;
  float* H;
  const float* GU;
  const float2* rope;
  int64_t ldh;
  int F, L, hd, rope_tiles;
  unsigned g_off, u_off;
  float* lse;
  // NORM: A holds the rows BEFORE an RMSNorm (nn/modules/norm.py:221-248); the wave normalises its rows in registers
  // (y = x / sqrt(mean(x^2) + eps) * w), multiplies THOSE, and leaves y and the rows' rms for the backward
  const float* norm_w;
  float* xn;
  float* rms;
  int64_t ldxn;
  float norm_eps;
  int ablate;                     // timing experiments (PDN_ROWTILE_ABLATE; 0 in the library): 1 = no barriers after the first,
                                  // 2 = the A rows are loaded from ONE cache line per lane (no 75 MB burst), 4 = no final drain
};

template <int V> using rt_ic = std::integral_constant<int, V>;

__device__ __forceinline__ float rt_sigmoid(float g) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * g));
}
// the neighbour lane's value (lane ^ 1): DPP quad_perm [1, 0, 3, 2]
__device__ __forceinline__ float rt_pair(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ void rt_barrier() {      // bare: __syncthreads() would drain vmcnt (loads AND stores in flight)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// BT: B is the row-major (N x 288) matrix whose transpose is meant.  EPI: 0 bias, 1 SwiGLU forward (NN), 2 SwiGLU
// backward (NT), 3 RoPE (NN), 5 row maxima (NN).  GUARD: M is not a multiple of 256 (row tests in every drain step).
// NORM: the RMSNorm in front of the projection is applied to the A rows in registers (its own pass over x is gone).
template <bool BT, int EPI, bool GUARD, bool NORM = false>
__global__ __launch_bounds__(512, 1) void gemm_rowtile_kernel(RowTileParams p) {
  static_assert(EPI != 1 || !BT, "SwiGLU forward: NN form");
  static_assert(EPI != 2 || BT, "SwiGLU backward: NT form");
  static_assert((EPI != 3 && EPI != 5) || !BT, "RoPE / row maxima: NN form");
  // (EPI 6 = EPI 0 + a residual read eight row steps ahead of its add, like the saved gate / up rows of EPI 2)
  constexpr int NSET = EPI == 1 ? 3 : 2;
  __shared__ __attribute__((aligned(16))) float smem[2 * RT_PIECE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int m0 = (blockIdx.x * 8 + wave) * 32;
  const int P_begin = blockIdx.y * p.ppw;
  const int P_end = min(p.pieces, P_begin + p.ppw);
  if (P_begin >= P_end) return;                     // (the whole workgroup)
  const unsigned ldb = (unsigned)p.ldb, ldc = (unsigned)p.ldc;

  // piece P: its 32 columns of B (NN) / rows of B^T (NT), and its first column in C
  auto piece_b = [&](int P) -> const float* {
    if (EPI == 1) return p.B + ((P & 1) ? p.u_off : p.g_off) + 32 * (P >> 1);
    const int blk = (int)__umulhi((unsigned)P, p.tpb_magic), tin = P - blk * p.tpb;
    return p.B + (int64_t)blk * p.b_bstride + (BT ? (int64_t)(tin * 32) * p.ldb : (int64_t)(tin * 32));
  };

  // staging plan: a piece is 36 wave instructions of 1 KiB; instruction I covers the 16-byte units 64 I .. 64 I + 63 of
  // the LDS image (linear in the lane: the park writes are conflict free); wave w moves I = w, w + 8, ...
  // NN image [k][32]:  unit U = row k = U / 8, column unit U % 8.
  // NT image [n][288]: unit U = row n = U / 72, k unit (U % 72) ^ ((n >> 1) & 7) -- the swizzle of gemm_rowres.hip on the
  // source side, so that one ds_read_b128 per four MFMAs is conflict free without padding.
  unsigned soff[BT ? 5 : 1];                        // BYTES: a 32-bit lane offset beside a wave-uniform base
  if (BT) {
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      const int I = min(q * 8 + wave, 35), U = 64 * I + lane;
      const int n = U / 72, cu = U - 72 * n;
      soff[BT ? q : 0] = 4u * ((unsigned)n * ldb + 4u * (unsigned)(cu ^ ((n >> 1) & 7)));
    }
  } else {
    // row k = 8 I + (lane >> 3): the 8 I rows go to the (uniform) base, one lane offset serves every instruction
    soff[0] = 4u * ((unsigned)(lane >> 3) * ldb + 4u * (unsigned)(lane & 7));
  }
  // source of instruction q of a piece that starts at `base`: uniform pointer + lane offset in bytes
  auto stage_src = [&](const float* base, int q) __attribute__((always_inline)) -> const float4* {
    unsigned o = soff[BT ? q : 0];
    asm volatile("" : "+v"(o));                     // (widened HERE, next to the load: the saddr + 32-bit voffset form)
    const float* b = BT ? base : base + (int64_t)(8 * min(q * 8 + wave, 35)) * p.ldb;
    return reinterpret_cast<const float4*>(reinterpret_cast<const char*>(b) + o);
  };
  float4 rb[3];
  auto issue_one = [&](const float* base, int q) __attribute__((always_inline)) {
    const float4 v = *stage_src(base, q);
    rb[q % 3].x = v.x; rb[q % 3].y = v.y; rb[q % 3].z = v.z; rb[q % 3].w = v.w;
  };
  // registers -> LDS.  The image is [n][288] with the 16-byte units of row n XOR-swizzled by (n >> 1) & 7 in BOTH forms, so
  // that the MFMA B fragments of a k-group are ONE conflict-free ds_read_b128 per lane.  NT: a staged float4 is such a unit
  // (the swizzle was applied on the source side: the write is linear in the lane).  NN: a staged float4 holds row k of
  // columns 4u .. 4u + 3, i.e. one float of four image rows: four ds_write_b32 (two lanes per bank -- the price of not
  // keeping a transposed copy of the weights; the chunk kernel read NN fragments as four ds_read_b32 per k-group instead).
  const int nn_pbase = (4 * (lane & 7)) * 288 + ((lane >> 3) & 3), nn_sw0 = (2 * (lane & 7)) & 7;
  auto park_v = [&](int buf, int q, const float4& v) __attribute__((always_inline)) {
    const int I = min(q * 8 + wave, 35);
    if (BT) {
      *reinterpret_cast<float4*>(smem + buf * RT_PIECE + I * 256 + 4 * lane) = v;
    } else {
      const int t0 = (2 * I + lh) ^ nn_sw0;         // k >> 2 = 2 I + (lane >> 5); columns 4u, 4u + 1 share (n >> 1) & 7
      float* d01 = smem + buf * RT_PIECE + nn_pbase + 4 * t0;
      float* d23 = smem + buf * RT_PIECE + nn_pbase + 2 * 288 + 4 * (t0 ^ 1);
      d01[0] = v.x; d01[288] = v.y; d23[0] = v.z; d23[288] = v.w;
    }
  };
  auto park_one = [&](int buf, int q) __attribute__((always_inline)) { park_v(buf, q, rb[q % 3]); };

  // ---- prologue: first piece into buffer 0, the wave's 32 rows of A into registers (requested behind the piece, which
  // is parked while they are still on their way) ------------------------------------------------------------------
  float4 t5[5];
  {
    const float* b0p = piece_b(P_begin);
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      const float4 v = *stage_src(b0p, q);
      t5[q].x = v.x; t5[q].y = v.y; t5[q].z = v.z; t5[q].w = v.w;
    }
  }
  float4 a[RT_KG];
  {
    const int arow_i = min(m0 + li, p.M - 1);
    const float* arow = p.A + (int64_t)arow_i * p.lda + 4 * lh;
    const int astep = (p.ablate & 2) ? 0 : 8;
#pragma unroll
    for (int t = 0; t < RT_KG; ++t) {
      const float4 v = *reinterpret_cast<const float4*>(arow + astep * t);
      a[t].x = v.x; a[t].y = v.y; a[t].z = v.z; a[t].w = v.w;
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    park_v(0, q, t5[q]);
  }
  if (NORM) {
    // a lane holds half of row m0 + li (k = 8 t + 4 lh .. + 3): sum of squares in-lane, the other half one shuffle away
    float ss = 0.f;
#pragma unroll
    for (int t = 0; t < RT_KG; ++t) ss += (a[t].x * a[t].x + a[t].y * a[t].y) + (a[t].z * a[t].z + a[t].w * a[t].w);
    ss += __shfl_xor(ss, 32, 64);
    // x * (1 / r) * w, sum of squares in-lane + one shuffle: the standalone pdn_rmsnorm_fwd_f32 (and the reference,
    // norm.py:221-248) compute x / r * w with a wave reduction, so xn and rms of the folded form agree with the unfolded
    // one to fp32 round-off, NOT bit for bit (tests/test_fused_epilogues.py: folded vs separate norm, 5e-6 relative,
    // full and ragged row blocks); the backward reuses the xn / rms stored HERE, so a step is self-consistent.
    const float r = sqrtf(ss / 288.f + p.norm_eps);
    const float inv = 1.f / r;
    const bool row_ok = !GUARD || m0 + li < p.M;
    if (lh == 0 && row_ok) p.rms[m0 + li] = r;
    const float* wrow = p.norm_w + 4 * lh;
    float* xrow = p.xn + (int64_t)min(m0 + li, p.M - 1) * p.ldxn + 4 * lh;
#pragma unroll
    for (int t0 = 0; t0 < RT_KG; t0 += 6) {         // the norm weights six k-groups at a time (register budget)
      float4 w6[6];
#pragma unroll
      for (int e = 0; e < 6; ++e) {
        const float4 v = *reinterpret_cast<const float4*>(wrow + 8 * (t0 + e));
        w6[e].x = v.x; w6[e].y = v.y; w6[e].z = v.z; w6[e].w = v.w;
      }
#pragma unroll
      for (int e = 0; e < 6; ++e) {
        const int t = t0 + e;
        a[t].x = a[t].x * inv * w6[e].x; a[t].y = a[t].y * inv * w6[e].y;
        a[t].z = a[t].z * inv * w6[e].z; a[t].w = a[t].w * inv * w6[e].w;
        if (row_ok) *reinterpret_cast<float4*>(xrow + 8 * t) = a[t];
      }
    }
  }

  f32x16 acc[NSET];
#pragma unroll
  for (int s = 0; s < NSET; ++s)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[s][r] = (EPI == 5 && s == 1) ? -INFINITY : 0.f;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // B fragment addressing (floats from smem), both forms: row li of the image, 16-byte unit (2 g + lh) ^ ((li >> 1) & 7)
  const int xl = (li >> 1) & 7;
  int bq[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bq[j] = li * 288 + 4 * ((2 * j + lh) ^ xl);

  // ---- drain state -----------------------------------------------------------------------------------------
  const bool wave_on = !GUARD || m0 < p.M;
  const int mrem = p.M - m0 - 4 * lh;               // GUARD: row step r exists when (r & 3) + 8 (r >> 2) < mrem
  float ring0[8], ring1[8];                         // EPI 2: saved gate / up of the row steps ahead; EPI 3: (cos, -+sin)
#pragma unroll
  for (int r = 0; r < 8; ++r) { ring0[r] = 0.f; ring1[r] = 0.f; }
  float mx[16];                                     // EPI 5: running maximum of register row r over this lane's columns
#pragma unroll
  for (int r = 0; r < 16; ++r) mx[r] = -INFINITY;
  float bvd = 0.f, bvn = 0.f;                       // bias of the draining / the multiplying tile's column
  unsigned colh_d = 0, colh_n = 0;                  // EPI 3: this lane's column inside its head (draining / multiplying tile)
  const float2* tabw = nullptr;
  if (EPI == 3) {
    const unsigned hd = (unsigned)p.hd;
    const unsigned hm = (unsigned)(((1ull << 32) + hd - 1) / hd);
    auto colh_of = [&](int P) -> unsigned { const unsigned x = 32u * (unsigned)P + (unsigned)li; return x - hd * __umulhi(x, hm); };
    colh_d = colh_of(P_end - 1);
    colh_n = colh_of(P_begin);
    tabw = p.rope + (int64_t)(m0 % p.L) * p.hd;     // (the 32 rows of a wave lie in one sequence)
  }
  // row offsets (in rows) of drain step r relative to the wave's first row, GUARD form: clamped into the matrix
  auto row_of = [&](int r) -> int { return (r & 3) + 8 * (r >> 2) + 4 * lh; };

  // first ring fill: rows 0..7 of the first (dummy) drain -- the place of the workgroup's last tile
  if (EPI == 2 && wave_on) {
    const float* Gd = p.GU + (int64_t)m0 * p.ldc + 32 * (P_end - 1);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const unsigned o = (unsigned)(GUARD ? min(row_of(r), p.M - 1 - m0) : row_of(r)) * (4u * ldc) + 4u * li;
      ring0[r] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(Gd) + o);
      ring1[r] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(Gd) + o + 4u * (unsigned)p.F);
    }
  }
  if (EPI == 6 && wave_on) {
    const float* Rd = p.residual + (int64_t)m0 * p.ldc + 32 * (P_end - 1);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const unsigned o = (unsigned)(GUARD ? min(row_of(r), p.M - 1 - m0) : row_of(r)) * (4u * ldc) + 4u * li;
      ring0[r] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(Rd) + o);
    }
  }
  if (EPI == 3 && wave_on) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float2 t = tabw[(unsigned)row_of(r) * (unsigned)p.hd + colh_d];
      ring0[r] = t.x; ring1[r] = t.y;
    }
  }

  // One row step of a drain.  AD / AU: the accumulator set(s) that leave (EPI 1: gate and up of a pair).
  // `od` / `oh`: running row offsets (C / H leading dimensions), advanced here.  Pd: the tile (EPI 1: the pair) that
  // leaves; Pn: the tile whose rows 0..7 are requested by steps 8..15 (none when LAST).
  // MODE (EPI 1 only): 0 = gate rows of a pair as they are, 1 = up rows + h.
  // byte-offset accessors: wave-uniform base + unsigned 32-bit lane offset in BYTES (the saddr form of global_load /
  // global_store; an offset in floats would have to be widened before the shift and costs a 64-bit address per row)
  auto ldf = [](const void* base, unsigned ob) __attribute__((always_inline)) -> float {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + ob);
  };
  auto stf = [](void* base, unsigned ob, float v) __attribute__((always_inline)) {
    *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + ob) = v;
  };
  const unsigned ldcb = 4u * ldc, ldhb = 4u * (unsigned)p.ldh, hdb = 8u * (unsigned)p.hd, Fb = 4u * (unsigned)p.F;
  auto drain_step = [&](const f32x16& AD, const f32x16& AU, const int r, auto modec, auto lastc, float* Cd, float* Hd,
                        const float* Gd, const float* Gn, bool rot_d, unsigned& od, unsigned& oh, unsigned& ot)
                        __attribute__((always_inline)) {
    constexpr int MODE = decltype(modec)::value;
    constexpr bool LAST = decltype(lastc)::value != 0;
    const bool ok = !GUARD || (r & 3) + 8 * (r >> 2) < mrem;
    // the step's inputs pass through an (empty) volatile asm: nothing of it can be computed ahead of its slot -- the
    // optimiser otherwise forms all 16 rows' values and addresses at the top of the piece and keeps them in registers
    float x = AD[r];
    asm volatile("" : "+v"(x), "+v"(od));
    if (EPI == 0 || EPI == 5) {
      const float v = x + bvd;
      if (EPI == 5) mx[r] = fmaxf(mx[r], v);
      if (ok) stf(Cd, od, v);
    } else if (EPI == 6) {
      if (ok) stf(Cd, od, (x + bvd) + ring0[r & 7]);
      if (r < 8 || !LAST) {
        const float* G = r < 8 ? Gd : Gn;           // (Gd / Gn: the residual rows of the leaving / the next tile)
        unsigned o;
        if (GUARD) o = (unsigned)min(row_of(r < 8 ? r + 8 : r - 8), p.M - 1 - m0) * ldcb + 4u * li;
        else o = r < 8 ? od + 16u * ldcb : od - 16u * ldcb;
        ring0[r & 7] = ldf(G, o);
      }
    } else if (EPI == 1) {
      if (MODE == 0) {
        if (ok) stf(Cd, od, x);
      } else {
        float u = AU[r];
        asm volatile("" : "+v"(u), "+v"(oh));
        if (ok) { stf(Cd, od, u); stf(Hd, oh, x * rt_sigmoid(x) * u); }
      }
    } else if (EPI == 2) {
      const float g = ring0[r & 7], u = ring1[r & 7], dh = x;
      const float s = rt_sigmoid(g), sl = g * s;
      const float dsl = fmaf(sl, 1.f - s, s);       // silu'(g) = s (1 + g (1 - s))
      if (ok) { stf(Cd, od, dh * u * dsl); stf(Cd, od + Fb, dh * sl); }
      if (r < 8 || !LAST) {
        // rows r + 8 of this tile / rows r - 8 of the next one (16 rows = +-16 ldc away; GUARD: clamped into the matrix)
        const float* G = r < 8 ? Gd : Gn;
        unsigned o;
        if (GUARD) o = (unsigned)min(row_of(r < 8 ? r + 8 : r - 8), p.M - 1 - m0) * ldcb + 4u * li;
        else o = r < 8 ? od + 16u * ldcb : od - 16u * ldcb;
        ring0[r & 7] = ldf(G, o); ring1[r & 7] = ldf(G, o + Fb);
      }
    } else if (EPI == 3) {
      asm volatile("" : "+v"(ot));
      const float w = fmaf(x, ring0[r & 7], rt_pair(x) * ring1[r & 7]);
      if (ok) stf(Cd, od, rot_d ? w : x);
      if (r < 8 || !LAST) {
        const unsigned o = r < 8 ? ot + 16u * hdb : ot - 16u * hdb - 8u * colh_d + 8u * colh_n;
        const float2 t = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(tabw) + o);
        ring0[r & 7] = t.x; ring1[r & 7] = t.y;
      }
    }
    const bool jump = (r & 3) == 3;
    od += jump ? 5u * ldcb : ldcb;
    if (EPI == 1) oh += jump ? 5u * ldhb : ldhb;
    if (EPI == 3) ot += jump ? 5u * hdb : hdb;
  };

  // ---- one piece: 36 k-groups of 4 MFMAs on accumulator set U % NSET out of LDS buffer U & 1; in their shadow the next
  // piece is staged and the previous tile(s) leave ---------------------------------------------------------------
  auto run_piece = [&](auto uc, int P) __attribute__((always_inline)) {
    constexpr int U = decltype(uc)::value;
    constexpr int S = U % NSET, BUF = U & 1;
    if (!(p.ablate & 1) || P == P_begin) rt_barrier();
    const int Pn = min(P + 1, P_end - 1);           // (after the last piece: a redundant fetch into the idle buffer)
    const float* nb = piece_b(Pn);
    // what leaves during this piece
    const bool first = P == P_begin;
    float* Cd = nullptr; float* Hd = nullptr; const float* Gd = nullptr; const float* Gn = nullptr;
    bool rot_d = false;
    if (EPI == 1) {
      if (U & 1) {                                  // gate rows of this pair (computed by the piece before)
        Cd = p.C + (int64_t)m0 * p.ldc + 32 * (P >> 1);
      } else {                                      // up rows + h of the pair before (first: the place of the last pair)
        const int ad = first ? (P_end >> 1) - 1 : (P >> 1) - 1;
        Cd = p.C + (int64_t)m0 * p.ldc + p.F + 32 * ad;
        Hd = p.H + (int64_t)m0 * p.ldh + 32 * ad;
      }
    } else {
      const int Pd = first ? P_end - 1 : P - 1;
      Cd = p.C + (int64_t)m0 * p.ldc + 32 * Pd;
      if (EPI == 2) { Gd = p.GU + (int64_t)m0 * p.ldc + 32 * Pd; Gn = p.GU + (int64_t)m0 * p.ldc + 32 * P; }
      if (EPI == 6) { Gd = p.residual + (int64_t)m0 * p.ldc + 32 * Pd; Gn = p.residual + (int64_t)m0 * p.ldc + 32 * P; }
      if (EPI == 3) rot_d = Pd < p.rope_tiles;
    }
    unsigned od = (unsigned)(4 * lh) * ldcb + 4u * li, oh = (unsigned)(4 * lh) * ldhb + 4u * li;
    unsigned ot = (unsigned)(4 * lh) * hdb + 8u * colh_d;

    // (fragment bases opaque per piece: the k-group addresses of one buffer are common subexpressions of every second
    //  piece body, and the optimiser keeps them alive -- in scratch -- rather than adding a constant again)
    asm volatile("" : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]));
    float bf[2][4];
#define RT_LOADB(X, G)                                                                          \
  {                                                                                             \
    const float4 v = *reinterpret_cast<const float4*>(                                          \
        __builtin_assume_aligned(smem + bq[(G) & 3] + BUF * RT_PIECE + 32 * ((G) >> 2), 16));   \
    bf[X][0] = v.x; bf[X][1] = v.y; bf[X][2] = v.z; bf[X][3] = v.w;                             \
  }
    RT_LOADB(0, 0)
#pragma unroll
    for (int g = 0; g < RT_KG; ++g) {
      if (g + 1 < RT_KG) { RT_LOADB((g + 1) & 1, g + 1) }
      __builtin_amdgcn_sched_barrier(0);
      {
        const float4 av = a[g];
        acc[S] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bf[g & 1][0], g == 0 ? zero16 : acc[S], 0, 0, 0);
        acc[S] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bf[g & 1][1], acc[S], 0, 0, 0);
        acc[S] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bf[g & 1][2], acc[S], 0, 0, 0);
        acc[S] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bf[g & 1][3], acc[S], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- slot g ----
      if (g < 3) issue_one(nb, g);                  // staging, first half: requested in k-groups 0..2, parked in 16
      if (g == 16) { park_one(BUF ^ 1, 0); park_one(BUF ^ 1, 1); park_one(BUF ^ 1, 2); }
      if (g == 17 || g == 18) issue_one(nb, g - 14);   // second half: requested in 17, 18, parked in 34
      if (g == 34) { park_one(BUF ^ 1, 3); park_one(BUF ^ 1, 4); }
      if ((EPI == 0 || EPI == 5 || EPI == 6) && g == 20) {      // bias of THIS tile's column (it leaves during the next piece)
        const float b = ldf(p.bias + 32 * P, 4u * li);
        bvn = p.has_bias ? b : 0.f;
      }
      if ((g & 1) && g >= 3 && g <= 33 && wave_on) {
        const int r = (g - 3) >> 1;                 // (a constant after unrolling)
        if (EPI == 1) {
          if (U & 1) drain_step(acc[(S + 2) % NSET], acc[(S + 2) % NSET], r, rt_ic<0>{}, rt_ic<0>{}, Cd, Hd, Gd, Gn, rot_d, od, oh, ot);
          else drain_step(acc[(S + 1) % NSET], acc[(S + 2) % NSET], r, rt_ic<1>{}, rt_ic<0>{}, Cd, Hd, Gd, Gn, rot_d, od, oh, ot);
        } else {
          drain_step(acc[(S + 1) % NSET], acc[(S + 1) % NSET], r, rt_ic<0>{}, rt_ic<0>{}, Cd, Hd, Gd, Gn, rot_d, od, oh, ot);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#undef RT_LOADB
    // the tile that was multiplied leaves next
    bvd = bvn;
    if (EPI == 3) {
      colh_d = colh_n;
      const unsigned c = colh_n + 32u;
      colh_n = c >= (unsigned)p.hd ? c - (unsigned)p.hd : c;
    }
  };

  // the last tile (EPI 1: up + h of the last pair) leaves with nothing to hide behind
  auto final_drain = [&](auto sc) __attribute__((always_inline)) {
    constexpr int SD = decltype(sc)::value;         // the set of the last piece (EPI 1: unused)
    if (!wave_on || (p.ablate & 4)) return;
    float* Cd; float* Hd = nullptr; const float* Gd = nullptr;
    bool rot_d = false;
    if (EPI == 1) {
      const int ad = (P_end >> 1) - 1;
      Cd = p.C + (int64_t)m0 * p.ldc + p.F + 32 * ad;
      Hd = p.H + (int64_t)m0 * p.ldh + 32 * ad;
    } else {
      Cd = p.C + (int64_t)m0 * p.ldc + 32 * (P_end - 1);
      if (EPI == 2) Gd = p.GU + (int64_t)m0 * p.ldc + 32 * (P_end - 1);
      if (EPI == 6) Gd = p.residual + (int64_t)m0 * p.ldc + 32 * (P_end - 1);
      if (EPI == 3) rot_d = P_end - 1 < p.rope_tiles;
    }
    unsigned od = (unsigned)(4 * lh) * ldcb + 4u * li, oh = (unsigned)(4 * lh) * ldhb + 4u * li;
    unsigned ot = (unsigned)(4 * lh) * hdb + 8u * colh_d;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (EPI == 1) drain_step(acc[1], acc[2 % NSET], r, rt_ic<1>{}, rt_ic<1>{}, Cd, Hd, Gd, Gd, rot_d, od, oh, ot);
      else drain_step(acc[SD % NSET], acc[SD % NSET], r, rt_ic<0>{}, rt_ic<1>{}, Cd, Hd, Gd, Gd, rot_d, od, oh, ot);
    }
  };

  if (EPI == 1) {
    // pieces per workgroup: a multiple of 6 (gate / up alternate, three accumulator sets rotate)
    for (int P = P_begin; P < P_end; P += 6) {
      run_piece(rt_ic<0>{}, P); run_piece(rt_ic<1>{}, P + 1); run_piece(rt_ic<2>{}, P + 2);
      run_piece(rt_ic<3>{}, P + 3); run_piece(rt_ic<4>{}, P + 4); run_piece(rt_ic<5>{}, P + 5);
    }
    final_drain(rt_ic<0>{});
  } else {
    int P = P_begin;
    for (; P + 1 < P_end; P += 2) { run_piece(rt_ic<0>{}, P); run_piece(rt_ic<1>{}, P + 1); }
    if (P < P_end) { run_piece(rt_ic<0>{}, P); final_drain(rt_ic<0>{}); }
    else final_drain(rt_ic<1>{});
  }

  if (EPI == 5) {
    // 16 x 32 candidates -> 16 row maxima by a halving butterfly (csrc/gemm_rowres.hip, rr_rowmax): at lane distance 16
    // a lane keeps 8 of its rows and hands 8 to its partner, then 4, 2, 1; lane l ends with register row l >> 1.
    float v[8];
    {
      const bool up = (li & 16) != 0;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float x = mx[r], y = mx[r + 8];
        v[r] = fmaxf(up ? y : x, __shfl_xor(up ? x : y, 16, 64));
      }
    }
#define RT_BFLY(D)                                                                   \
  {                                                                                  \
    const bool up = (li & (2 * (D))) != 0;                                           \
    _Pragma("unroll") for (int r = 0; r < (D); ++r) {                                \
      const float send = up ? v[r] : v[r + (D)], keep = up ? v[r + (D)] : v[r];      \
      v[r] = fmaxf(keep, __shfl_xor(send, 2 * (D), 64));                             \
    }                                                                                \
  }
    RT_BFLY(4) RT_BFLY(2) RT_BFLY(1)
#undef RT_BFLY
    const float m = fmaxf(v[0], __shfl_xor(v[0], 1, 64));
    const int r = li >> 1, row = m0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (!(li & 1) && row < p.M) p.lse[(int64_t)blockIdx.y * p.M + row] = m;
  }
}

// ---- host side -----------------------------------------------------------------------------------------------
static void rowtile_plan(const RowTileArgs& a, int* pieces, int* ppw, int* nsplit) {
  const int np = a.epi == 1 ? 2 * (a.F / 32) : a.N / 32;
  const int row_blocks = (a.M + 255) / 256;
  int ns = 1;
  while (row_blocks * ns < 256 && ns < np) ++ns;
  int per = (np + ns - 1) / ns;
  if (a.epi == 1) per = (per + 5) / 6 * 6;
  *pieces = np; *ppw = per; *nsplit = (np + per - 1) / per;
}

// 0 = never (the chunk kernel of gemm_rowres.hip runs), 1 = when every CU gets an 8-wave workgroup (default), 2 = whenever
// the shape is valid (tests: small and ragged shapes on this kernel).  PDN_ROWTILE sets the initial value.
static int g_rowtile_mode = getenv("PDN_ROWTILE") ? atoi(getenv("PDN_ROWTILE")) : 1;
extern "C" int pdn_gemm_rowtile_mode(int mode) {
  const int prev = g_rowtile_mode;
  if (mode >= 0 && mode <= 2) g_rowtile_mode = mode;
  return prev;
}

int pdn_rowtile_takes(const RowTileArgs& a) {
  if (g_rowtile_mode == 0) return 0;
  if (a.M < 1 || a.N < 64 || a.N % 32 != 0) return 0;
  if (a.nblocks < 1 || (a.N / a.nblocks) % 32 != 0 || (a.N / a.nblocks) * a.nblocks != a.N) return 0;
  if (a.epi == 1 && (a.F % 96 != 0 || a.b_trans)) return 0;
  if (a.epi == 2 && !a.b_trans) return 0;
  if ((a.epi == 3 || a.epi == 5) && a.b_trans) return 0;
  if (a.epi == 4) return 0;
  if (a.residual && a.epi != 0) return 0;
  const int np = a.epi == 1 ? 2 * (a.F / 32) : a.N / 32;
  const int row_blocks = (a.M + 255) / 256;
  // one 8-wave workgroup per CU must have work (rows, or rows x column ranges); below that the 4-wave chunk kernel runs
  if (g_rowtile_mode != 2 && !(row_blocks >= 192 || (int64_t)row_blocks * np >= 256)) return 0;
  if ((int64_t)287 * a.ldb + 4096 >= (1ll << 30) || (int64_t)48 * a.ldc + 2 * (int64_t)a.F + 64 >= (1ll << 30)) return 0;
  return 1;
}

int pdn_rowtile_launch(const RowTileArgs& a, void* stream) {
  int pieces, ppw, nsplit;
  rowtile_plan(a, &pieces, &ppw, &nsplit);
  if (a.epi == 5 && !a.lse) { if (a.parts) *a.parts = nsplit; return PDN_OK; }
  RowTileParams p;
  memset(&p, 0, sizeof(p));
  p.A = a.A; p.B = a.B; p.C = a.C;
  p.bias = a.bias ? a.bias : a.B; p.has_bias = a.bias ? 1 : 0;
  p.residual = a.residual;
  p.M = a.M; p.N = a.N; p.lda = a.lda; p.ldb = a.ldb; p.ldc = a.ldc;
  p.pieces = pieces; p.ppw = ppw;
  p.tpb = a.epi == 1 ? pieces : (a.N / a.nblocks) / 32;
  p.tpb_magic = (unsigned)(((1ull << 32) + (unsigned)p.tpb - 1) / (unsigned)p.tpb);
  p.b_bstride = a.nblocks > 1 ? a.b_block_stride : 0;
  p.H = a.H; p.GU = a.GU; p.rope = reinterpret_cast<const float2*>(a.rope); p.ldh = a.ldh;
  p.F = a.F; p.L = a.L > 0 ? a.L : 1; p.hd = a.hd > 0 ? a.hd : 32; p.rope_tiles = a.rope_cols / 32;
  p.g_off = a.g_off; p.u_off = a.u_off; p.lse = a.lse;
  static const int s_pdn_rowtile_ablate = pdn_ablation_switch("PDN_ROWTILE_ABLATE");
  p.ablate = s_pdn_rowtile_ablate;
  p.norm_w = a.norm_w; p.xn = a.xn; p.rms = a.rms; p.ldxn = a.ldxn; p.norm_eps = a.norm_eps;
  const bool norm = a.norm_w != nullptr;
  if (norm && !(a.epi == 1 || a.epi == 3)) { pdn_set_error("pdn_rowtile_launch: the RMSNorm fold exists for the gate | up and q | k | v projections"); return PDN_EINVAL; }
  if (norm && !(a.xn && a.rms && (a.ldxn & 3) == 0 && (((uintptr_t)a.xn | (uintptr_t)a.norm_w) & 15) == 0)) { pdn_set_error("pdn_rowtile_launch: RMSNorm fold needs xn / rms outputs, 16-byte aligned"); return PDN_EINVAL; }
  const dim3 grid((a.M + 255) / 256, nsplit), block(512);
  hipStream_t st = (hipStream_t)stream;
  const bool guard = a.M % 256 != 0;
#define RT_LAUNCH(BT_, EPI_)                                                                          \
  if (guard) hipLaunchKernelGGL((gemm_rowtile_kernel<BT_, EPI_, true>), grid, block, 0, st, p);        \
  else hipLaunchKernelGGL((gemm_rowtile_kernel<BT_, EPI_, false>), grid, block, 0, st, p)
#define RT_LAUNCH_N(BT_, EPI_)                                                                        \
  if (guard) hipLaunchKernelGGL((gemm_rowtile_kernel<BT_, EPI_, true, true>), grid, block, 0, st, p);  \
  else hipLaunchKernelGGL((gemm_rowtile_kernel<BT_, EPI_, false, true>), grid, block, 0, st, p)
  switch (a.epi) {
    case 0:
      if (a.residual) { if (a.b_trans) { RT_LAUNCH(true, 6); } else { RT_LAUNCH(false, 6); } }
      else if (a.b_trans) { RT_LAUNCH(true, 0); } else { RT_LAUNCH(false, 0); }
      break;
    case 1: if (norm) { RT_LAUNCH_N(false, 1); } else { RT_LAUNCH(false, 1); } break;
    case 2: RT_LAUNCH(true, 2); break;
    case 3: if (norm) { RT_LAUNCH_N(false, 3); } else { RT_LAUNCH(false, 3); } break;
    case 5: RT_LAUNCH(false, 5); break;
    default: pdn_set_error("pdn_rowtile_launch: epilogue %d", a.epi); return PDN_EINVAL;
  }
#undef RT_LAUNCH
#undef RT_LAUNCH_N
  pdn_count(a.epi == 0 ? PDN_CNT_ROWTILE_PLAIN : a.epi == 5 ? PDN_CNT_ROWTILE_ROWMAX : PDN_CNT_ROWTILE_PLAIN + a.epi);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}
