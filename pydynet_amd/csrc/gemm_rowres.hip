// Row-resident fp32 GEMM for the layer projections (short contraction, gfx950 only).
//
//   C (M x N) = A (M x K, rows contiguous) * B (K x N)  (+ bias[N]) (+ residual[M x N])
//
// The products around a transformer block -- x Wq|Wk|Wv, x Wg|Wu, h W_out (llm/llama/model.py:93-121,
// 56-58, 179: `x @ W` through pydynet/core/tensor.py:657-676) and the input gradients dY W^T that contract
// over 288 features -- have a contraction of a few hundred and a tall A: a tiled kernel spends its
// time in per-tile prologues, a barrier every k-tile and LDS traffic for both operands.  Here
//   * a wave owns 32 rows of A and keeps ALL of their K values in registers as MFMA A operands
//     (lane (i, h) holds A[row i][8t + 4h .. +3] for t < K/8: K/2 VGPRs, loaded once);
//   * B is streamed through LDS in 96 (k) x 96 (n) pieces by LDS-DMA (`global_load_lds_dwordx4`: no VGPR
//     staging, no ds_write), double buffered, shared by the four waves of a workgroup: one barrier
//     per 144 MFMAs per wave.  The nine DMA instructions a wave contributes to the next piece are
//     spread over the MFMA stream (issued as one burst after the barrier they back up the address path
//     and hold all four waves in VMEM issue: 82 -> 86 % of the matrix peak on 65536 x 32064 x 288);
//   * B row-major [k][n] (forward, `x @ W`): LDS image [k][96], fragments are conflict-free ds_read_b32;
//     B^T row-major [n][k] (`grad @ W^T`, tensor.py:670): LDS image [n][96] with the 16-byte units of
//     row n XOR-swizzled by (n >> 1) & 7 -- applied on the SOURCE side of the DMA, whose LDS side is
//     linear in the lane -- so one ds_read_b128 per four MFMAs is conflict free without padding;
//   * the accumulator (lane = output column, registers = rows) is stored straight from registers:
//     every store instruction writes two full 128-byte row segments.
// A never passes through LDS and there is no k-tile prologue: between barriers a wave issues an
// unbroken MFMA stream.  Two workgroups (2 x 4 waves) per CU run out of phase.
// The k-order of every output element is the same as in csrc/gemm.hip (k = 8t + 4h + j inside an MFMA
// group, groups ascending), so both kernels give bit-identical results.
#include "common.h"
#include "gemm_rowtile.h"
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

struct RowResParams {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  const float* residual;
  int M, N;
  int64_t lda, ldb, ldc;
  int chunks, chunks_per_wg;      // 96-column chunks of N in total / per workgroup (grid.y)
  int cpb;                        // chunks per column block of B (B may come as several equally spaced
  int64_t b_bstride;              // matrices side by side: Wq | Wk | Wv); `chunks` when B is one matrix
  // ---- fused epilogues (EPI != 0, see below) ----
  float* H;                       // EPI 1: silu(gate) * up, (M x F), leading dimension ldh
  const float* GU;                // EPI 2: the saved [gate | up] rows (M x 2F, leading dimension ldc) of the forward
  const float2* rope;             // EPI 3: (L x hd) table of (cos, -+sin) per column of a head (rr_rope_table)
  int64_t ldh;
  int F;                          // EPI 1 / 2: FFN width (columns per half of the packed buffer)
  int L, hd, rope_chunks;         // EPI 3: positions per sequence, head dim, chunks (of 96 columns) that are rotated
  unsigned hd_magic;              //        2^32 / hd rounded up (x / hd = umulhi(x, magic) for x < 2^16)
  unsigned g_off, u_off;          // EPI 1: first float of the gate / up matrix relative to B
  float* lse;                     // EPI 4: log-sum-exp of every output row (M); EPI 5: their maxima
  int epi_ablate;                 // timing experiments (PDN_ROWRES_EPI_ABLATE; 0 in the library): 2 = EPI 5 without its maximum
};

__device__ __forceinline__ void rr_glds16(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

#define RR_NC 96          // columns per chunk (three 32-wide MFMA tiles)
#define RR_KP 96          // contraction rows per piece (twelve 8-row groups)

// Accumulator block (three 32 x 32 tiles: lane = column, register r = row (r & 3) + 8 (r >> 2) + 4 h) ->
// C rows, + bias, + residual.  Wave-uniform 64-bit bases + one running 32-bit lane offset (opaque to the
// optimiser: hoisting 48 loop-invariant addresses out of the chunk loop costs more registers than exist).
// `nt` (1..3): tiles of this chunk that lie inside N.
__device__ __forceinline__ void rr_store(const RowResParams& p, f32x16 (&acc)[3], int m0, int c, int li, int lh,
                                         bool full, int nt) {
  float* __restrict__ Cw = p.C + (int64_t)m0 * p.ldc + c * RR_NC;
  const float* __restrict__ Rw = p.residual ? p.residual + (int64_t)m0 * p.ldc + c * RR_NC : nullptr;
  const unsigned ldc = (unsigned)p.ldc;
  unsigned o0 = (unsigned)(4 * lh) * ldc + li;
  asm volatile("" : "+v"(o0));
  float bv[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) bv[j] = (p.bias && j < nt) ? p.bias[c * RR_NC + 32 * j + li] : 0.f;
  const int mrem = p.M - m0 - 4 * lh;             // rows rr < mrem exist
#define RR_ROWS(GUARD, RES)                                                      \
  {                                                                              \
    unsigned o = o0;                                                             \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                              \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) {                            \
        const int r = 4 * q + e;                                                 \
        if (!GUARD || 8 * q + e < mrem) {                                        \
          _Pragma("unroll") for (int j = 0; j < 3; ++j) {                        \
            if (!GUARD || j < nt) {                                              \
              float v = acc[j][r] + bv[j];                                       \
              if (RES) v += Rw[o + 32 * j];                                      \
              Cw[o + 32 * j] = v;                                                \
            }                                                                    \
          }                                                                      \
        }                                                                        \
        o += (e == 3) ? 5 * ldc : ldc;                                           \
      }                                                                          \
    }                                                                            \
  }
  if (p.epi_ablate & 4) {
  } else if (full && nt == 3) {
    if (Rw) RR_ROWS(false, true) else RR_ROWS(false, false)
  } else {
    if (Rw) RR_ROWS(true, true) else RR_ROWS(true, false)
  }
#undef RR_ROWS
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
}


// ---------------------------------------------------------------------------------------------------------
// Fused epilogues (round 4): the bandwidth passes the reference composes as separate nodes next to a projection
// ride in the store of the accumulator block instead of being a read + write of the whole activation.
//   EPI 1  gate | up projection with SwiGLU (llm/llama/model.py:56-58, nn/functional.py:39-40): the 96-column chunks
//          of a workgroup walk the two weight matrices in TILE pairs -- chunk 2p = (gate a, up a, gate b), chunk
//          2p + 1 = (up b, gate c, up c) with a, b, c the three 32-column tiles 3p .. 3p + 2 of the FFN width -- so
//          that gate and up of the same columns meet in the registers of one lane: h = silu(g) u leaves with the
//          packed [gate | up] rows (kept for the backward); only tile `gate b` waits one chunk, parked in 4 KiB of
//          LDS per wave.  NN form, 8-wave workgroups.
//   EPI 2  dh = dy W_down^T with the SwiGLU gradient: the accumulator block is dh for 96 hidden units; the saved
//          gate / up values of the same positions are read (lane = column: 128-byte row segments) and
//          d[gate | up] is what leaves -- dh never exists in memory.  NT form.
//   EPI 3  q | k | v projection with RoPE (model.py:23-44) on the q and k column blocks: lane = column, so the
//          interleaved pair (2i, 2i + 1) is the neighbour lane (one DPP quad permute); (cos, -+sin) of the lane's
//          column come from an expanded (L x hd) table.  The attention kernels then read rotated rows as they are.
__device__ __forceinline__ float rr_sigmoid(float g) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * g));
}

template <bool GUARD>
__device__ __forceinline__ void rr_store_swiglu(const RowResParams& p, f32x16 (&acc)[3], int m0, int c, int li, int lh,
                                                float* __restrict__ park) {
  const int pr = c >> 1, odd = c & 1, F = p.F;
  float* __restrict__ Cw = p.C + (int64_t)m0 * p.ldc + 96 * pr;
  float* __restrict__ Hw = p.H + (int64_t)m0 * p.ldh + 96 * pr;
  const unsigned ldc = (unsigned)p.ldc, ldh = (unsigned)p.ldh;
  unsigned o = (unsigned)(4 * lh) * ldc + li, oh = (unsigned)(4 * lh) * ldh + li;
  asm volatile("" : "+v"(o), "+v"(oh));
  const int mrem = p.M - m0 - 4 * lh;
  if (!odd) {                                       // tiles: gate a | up a | gate b
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * q + e;
        const float g = acc[0][r], u = acc[1][r], g2 = acc[2][r];
        park[r * 64] = g2;
        if (!GUARD || 8 * q + e < mrem) {
          Cw[o] = g; Cw[o + F] = u; Cw[o + 32] = g2;
          Hw[oh] = g * rr_sigmoid(g) * u;
        }
        o += (e == 3) ? 5 * ldc : ldc; oh += (e == 3) ? 5 * ldh : ldh;
      }
    }
  } else {                                          // tiles: up b | gate c | up c
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * q + e;
        const float ub = acc[0][r], g = acc[1][r], u = acc[2][r];
        const float gb = park[r * 64];
        if (!GUARD || 8 * q + e < mrem) {
          Cw[o + F + 32] = ub; Cw[o + 64] = g; Cw[o + F + 64] = u;
          Hw[oh + 32] = gb * rr_sigmoid(gb) * ub;
          Hw[oh + 64] = g * rr_sigmoid(g) * u;
        }
        o += (e == 3) ? 5 * ldc : ldc; oh += (e == 3) ? 5 * ldh : ldh;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
}

template <bool GUARD>
__device__ __forceinline__ void rr_store_swiglu_bwd(const RowResParams& p, f32x16 (&acc)[3], int m0, int c, int li, int lh) {
  const int F = p.F;
  float* __restrict__ Cw = p.C + (int64_t)m0 * p.ldc + c * RR_NC;
  const float* __restrict__ Gw = p.GU + (int64_t)m0 * p.ldc + c * RR_NC;
  const unsigned ldc = (unsigned)p.ldc;
  const int mrem = p.M - m0 - 4 * lh;
  // half a tile (8 rows of 32 columns) at a time: 16 loads in flight, then 16 stores -- what the registers beside
  // the A block and the accumulators allow
#pragma unroll
  for (int j = 0; j < 3; ++j) {
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      unsigned o = (unsigned)(4 * lh + 16 * hf) * ldc + li + 32 * j;
      asm volatile("" : "+v"(o));
      float gv[8], uv[8];
      {
        unsigned oo = o;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const bool ok = !GUARD || 16 * hf + 8 * (r >> 2) + (r & 3) < mrem;
          gv[r] = ok ? Gw[oo] : 0.f;
          uv[r] = ok ? Gw[oo + F] : 0.f;
          oo += ((r & 3) == 3) ? 5 * ldc : ldc;
        }
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float g = gv[r], s = rr_sigmoid(g), sl = g * s, dh = acc[j][8 * hf + r];
        const float dsl = fmaf(sl, 1.f - s, s);       // silu'(g) = s (1 + g (1 - s))
        if (!GUARD || 16 * hf + 8 * (r >> 2) + (r & 3) < mrem) {
          Cw[o] = dh * uv[r] * dsl;
          Cw[o + F] = dh * sl;
        }
        o += ((r & 3) == 3) ? 5 * ldc : ldc;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
}

// the neighbour lane's value (lane ^ 1): DPP quad_perm [1, 0, 3, 2]
__device__ __forceinline__ float rr_pair(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

template <bool GUARD>
__device__ __forceinline__ void rr_store_rope(const RowResParams& p, f32x16 (&acc)[3], int m0, int c, int li, int lh) {
  float* __restrict__ Cw = p.C + (int64_t)m0 * p.ldc + c * RR_NC;
  const unsigned ldc = (unsigned)p.ldc, hd = (unsigned)p.hd;
  const int mrem = p.M - m0 - 4 * lh;
  const bool rot = c < p.rope_chunks;               // (uniform)
  const float2* __restrict__ tab = p.rope + (int64_t)(m0 % p.L) * p.hd;      // the 32 rows of a wave lie in one sequence
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    if (rot) {
      // column inside its head: (96 c + 32 j + li) mod hd with the uniform part reduced by a multiply-high (hd >= 32)
      const unsigned x0 = (unsigned)__builtin_amdgcn_readfirstlane(c * RR_NC + 32 * j);
      const unsigned b0 = x0 - hd * __umulhi(x0, p.hd_magic);
      unsigned colh = b0 + (unsigned)li;
      colh = colh >= hd ? colh - hd : colh;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {              // half a tile at a time: 8 table entries (16 registers) in flight
        unsigned o = (unsigned)(4 * lh + 16 * hf) * ldc + li + 32 * j;
        unsigned ot = (unsigned)(4 * lh + 16 * hf) * hd + colh;
        asm volatile("" : "+v"(o), "+v"(ot));
        float2 cs[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const float2 t = tab[ot];
          cs[r].x = t.x; cs[r].y = t.y;
          ot += ((r & 3) == 3) ? 5 * hd : hd;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const float v = acc[j][8 * hf + r];
          const float w = fmaf(v, cs[r].x, rr_pair(v) * cs[r].y);
          if (!GUARD || 16 * hf + 8 * (r >> 2) + (r & 3) < mrem) Cw[o] = w;
          o += ((r & 3) == 3) ? 5 * ldc : ldc;
        }
      }
    } else {
      unsigned o = (unsigned)(4 * lh) * ldc + li + 32 * j;
      asm volatile("" : "+v"(o));
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (!GUARD || 8 * (r >> 2) + (r & 3) < mrem) Cw[o] = acc[j][r];
        o += ((r & 3) == 3) ? 5 * ldc : ldc;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
}

//   EPI 4  vocabulary projection with the ROW STATISTICS of cross entropy (llm/llama/model.py:179 feeding
//          nn/functional.py:364-381): the MFMA operands are swapped -- D^T[vocabulary][token] = W^T x^T, the S^T = K Q^T
//          trick of the attention kernels -- so that a LANE owns a token and the registers run over the vocabulary:
//          the running row maximum and sum of exponentials are in-lane arithmetic plus ONE cross-half shuffle per
//          96-column chunk (with lane = column they were five cross-lane steps per row and statistic: tried in round
//          3, +1.95 ms).  The logits leave as 16-byte pieces of the lane's own row (the two half-waves complete a
//          128-byte line between them); the separate statistics pass over the logits (8.4 GB read, 1.5 ms) is gone.
//   EPI 5  the row MAXIMUM only, in the plain orientation (lane = column) and with the plain store.  Measured on the 9.0 ms
//          product (tools/lmhead_probe.py): the transposed form costs +0.6 ms even without a single exponential --
//          0.5 ms of it the store, 64 pieces of 16 bytes in 32 rows per instruction instead of two 128-byte row segments
//          -- and all of it exposed, because the waves of a workgroup finish their chunks together.  Here the 16 x 64
//          candidates of a chunk (16 register rows per lane, after the in-lane maximum over the three tiles) shrink by a
//          HALVING butterfly: at lane distance 16 each lane keeps 8 of its rows and hands the other 8 to its partner, then
//          4, 2, 1 -- 16 cross-lane moves per chunk instead of 80 -- and lane l ends with the maximum of register row
//          l >> 1.  The sum of exponentials is a by-product of the input-gradient product, which forms exp(logit - max)
//          anyway (csrc/gemm_outres.hip, CE == 2).
__device__ __forceinline__ void rr_rowmax(const RowResParams& p, const f32x16 (&acc)[3], int c, int li, int nt, float& m_run) {
  float bv[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) bv[j] = (p.bias && j < nt) ? p.bias[c * RR_NC + 32 * j + li] : 0.f;
  auto row3 = [&](int r) {                          // in-lane maximum of register row r over the chunk's tiles
    float m = acc[0][r] + bv[0];
    if (nt > 1) m = fmaxf(m, acc[1][r] + bv[1]);
    if (nt > 2) m = fmaxf(m, acc[2][r] + bv[2]);
    return m;
  };
  float v[8];
  {
    const bool up = (li & 16) != 0;                 // lane distance 16: keep 8 rows, pass 8 rows on
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float a = row3(r), b = row3(r + 8);
      v[r] = fmaxf(up ? b : a, __shfl_xor(up ? a : b, 16, 64));
    }
  }
  // lane distance 2 D: keep D rows, pass D rows on (D a literal: with a loop over D the compiler indexed v[] dynamically
  // -- select chains and scratch -- and the epilogue measured 1.3 ms on the 9 ms product)
#define RR_BFLY(D)                                                                   \
  {                                                                                  \
    const bool up = (li & (2 * (D))) != 0;                                           \
    _Pragma("unroll") for (int r = 0; r < (D); ++r) {                                \
      const float send = up ? v[r] : v[r + (D)], keep = up ? v[r + (D)] : v[r];      \
      v[r] = fmaxf(keep, __shfl_xor(send, 2 * (D), 64));                             \
    }                                                                                \
  }
  RR_BFLY(4) RR_BFLY(2) RR_BFLY(1)
#undef RR_BFLY
  m_run = fmaxf(m_run, fmaxf(v[0], __shfl_xor(v[0], 1, 64)));
}

template <bool GUARD>
__device__ __forceinline__ void rr_store_lse(const RowResParams& p, f32x16 (&acc)[3], int m0, int c, int li, int lh, int nt,
                                             float& m_run, float& z_run) {
  const bool ok = !GUARD || m0 + li < p.M;
  float* __restrict__ Cw = p.C + (int64_t)(m0 + (ok ? li : 0)) * p.ldc + c * RR_NC + 4 * lh;
  const float* __restrict__ Bw = p.bias ? p.bias + c * RR_NC + 4 * lh : nullptr;
  // bias of the lane's vocabulary entries (register group g of tile j: columns 32 j + 8 g + 4 h + 0..3), then the chunk's maximum
  float mc = -INFINITY;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    if (j < nt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (Bw) {
          const float4 b = *reinterpret_cast<const float4*>(Bw + 32 * j + 8 * g);
          acc[j][4 * g] += b.x; acc[j][4 * g + 1] += b.y; acc[j][4 * g + 2] += b.z; acc[j][4 * g + 3] += b.w;
        }
        mc = fmaxf(mc, fmaxf(fmaxf(acc[j][4 * g], acc[j][4 * g + 1]), fmaxf(acc[j][4 * g + 2], acc[j][4 * g + 3])));
        if (ok) *reinterpret_cast<float4*>(Cw + 32 * j + 8 * g) =
            make_float4(acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]);
      }
    }
  }
  mc = fmaxf(mc, __shfl_xor(mc, 32, 64));           // the row's other 48 columns of this chunk
  const float mn = fmaxf(m_run, mc);
  const float L2E = 1.4426950408889634f, c2 = -mn * L2E;
  float zc = 0.f;
#pragma unroll
  for (int j = 0; j < 3; ++j)
    if (j < nt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) zc += __builtin_amdgcn_exp2f(fmaf(acc[j][r], L2E, c2));
    }
  z_run = z_run * __builtin_amdgcn_exp2f((m_run - mn) * L2E) + zc;     // (first chunk: exp2(-inf) = 0)
  m_run = mn;
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
}

// KG = K / 8.  BT: B is given as the row-major (N x K) matrix whose transpose is meant.
// ABLATE (timing experiments only, 0 in the library): 2 = no B DMA after the first two pieces,
// 32 = the DMA of a piece issued as one burst.
// NW: waves per workgroup (4: two workgroups per CU; 8: one -- half the DMA instructions per wave).
// STAGE: the B piece reaches LDS by LDS-DMA (0) or through registers, global_load_dwordx4 + ds_write_b128 (1):
// an LDS-DMA instruction holds up the issuing wave's MFMA stream for 50-170 cycles, a plain load far less.
// EPI: fused epilogue (0 none, 1 SwiGLU forward, 2 SwiGLU backward, 3 RoPE on the leading chunks), see above.
template <int KG, bool BT, int NW, int STAGE, int ABLATE = 0, int EPI = 0>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void gemm_rowres_kernel(RowResParams p) {
  static_assert(EPI != 1 || (!BT && NW == 8), "SwiGLU forward epilogue: NN form, one 8-wave workgroup per CU");
  static_assert(EPI != 2 || BT, "SwiGLU backward epilogue: NT form");
  static_assert((EPI != 4 && EPI != 5) || !BT, "row-statistics epilogues: NN form");
  constexpr int NQ = (36 + NW - 1) / NW;          // DMA instructions per wave and piece
  constexpr int NPK = KG / 12;                    // pieces along K
  constexpr int PIECE = RR_KP * RR_NC;            // floats
  static_assert(KG % 12 == 0, "K must be a multiple of 96");
  __shared__ __attribute__((aligned(16))) float smem[2 * PIECE + (EPI == 1 ? NW * 1024 : 0)];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int m0 = (blockIdx.x * NW + wave) * 32;
  const int c_begin = blockIdx.y * p.chunks_per_wg;
  const int c_end = min(p.chunks, c_begin + p.chunks_per_wg);
  const int nloc = c_end - c_begin;
  // (walking the chunks in a different order on every XCD, so that 512 workgroups in lockstep do not ask
  //  for the same 36 KB piece at the same moment, measured neutral to slower: the order stays 0, 1, 2 ...)
  auto chunk_of = [&](int ci) { return c_begin + ci; };
  // first element of chunk c in B: column block c / cpb, 96-column (NN) or 96-row (NT) slice c % cpb of it
  auto chunk_base = [&](int c) -> const float* {
    if (EPI == 1) return p.B;                       // (tile bases: see issue_one)
    const int blk = c / p.cpb, cin = c - blk * p.cpb;
    return p.B + blk * p.b_bstride + (BT ? (int64_t)(cin * RR_NC) * p.ldb : (int64_t)(cin * RR_NC));
  };
  const int c_tail = (p.N % RR_NC) ? p.chunks - 1 : -1;      // the chunk that sticks out of N, if any

  // DMA plan: a piece is 36 wave instructions of 1 KiB; instruction I covers the 16-byte units
  // 64 I .. 64 I + 63 of the piece = units u = lane + 64 (I % 3) of 8-row group I / 3: row u / 24, unit
  // u % 24 of that row (wave-uniform 64-bit base + unsigned 32-bit lane offset).  Wave w issues I = w, w + NW, ...
  // NN: rows are k, units run along n.   NT: rows are n, units run along k, and LDS unit cu' of row n
  // receives source unit cu' ^ ((n >> 1) & 7).
  // The last chunk of an N that is not a multiple of 96 (`c_tail`, nt = 1 or 2 tiles inside N) reads
  // sources folded back into the matrix -- NN: column units modulo 8 nt (`offt`), NT: row groups modulo
  // 4 nt -- the tiles outside N are computed on repeated data and never stored.  No branches in the
  // DMA path: it is issued between MFMAs.
  const int nt_tail = c_tail >= 0 ? (p.N - c_tail * RR_NC) / 32 : 3;
  // (the lane offset is recomputed per instruction -- ten VALU operations in the shadow of the MFMAs --
  //  rather than kept in registers: the A block leaves none to spare)
  const unsigned ldb = (unsigned)p.ldb;
  // (nt_tail is 1 or 2 when there is a tail chunk, so both folds are masks: no division, no branch)
  auto lane_off = [&](int j, int gk, int cmask) -> unsigned {
    int ln = lane;
    asm volatile("" : "+v"(ln));                  // recomputed at every use: hoisting these out of the loops spills
    const int u = ln + 64 * j, row = u / 24;
    int cu = u - 24 * row;
    if (BT) cu ^= (4 * (gk & 1) + (row >> 1)) & 7;
    else cu &= cmask;
    return (unsigned)row * ldb + 4u * (unsigned)cu;
  };
  // EPI 1: the three 32-column tiles of chunk c come from different places -- chunk 2p: gate tile 3p, up tile 3p,
  // gate tile 3p + 1; chunk 2p + 1: up tile 3p + 1, gate tile 3p + 2, up tile 3p + 2 (up = gate + b_bstride floats)
  auto lane_off_tiles = [&](int j, int c) -> unsigned {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int u = ln + 64 * j, row = u / 24, cu = u - 24 * row, t = cu >> 3;
    // (gate / up offsets from p.B = the lower of the two addresses: the matrices may sit in either order)
    const unsigned gt = p.g_off + 96u * (unsigned)(c >> 1), ut = p.u_off + 96u * (unsigned)(c >> 1);
    const unsigned tb0 = (c & 1) ? ut + 32u : gt, tb1 = (c & 1) ? gt + 64u : ut, tb2 = (c & 1) ? ut + 64u : gt + 32u;
    const unsigned tb = t == 0 ? tb0 : (t == 1 ? tb1 : tb2);
    return (unsigned)row * ldb + tb + 4u * (unsigned)(cu & 7);
  };
  // instruction q (0 .. NQ-1) of this wave's share of piece (chunk c, k-piece kc) into buffer `buf`
  // (I clamped: a repeated instruction is harmless)
  // STAGE 1: the wave's share of the next piece on its way to LDS, in two halves (fetched in the first / second
  // half of the current piece and parked in its middle / at its end): half the staging registers
  constexpr int NPH = BT ? 2 : 3;                  // staging phases per piece (NN: three, to fit the register budget)
  constexpr int WIN = 12 / NPH;                    // slots (k-groups) per phase
  constexpr int RBN = STAGE ? (NQ + NPH - 1) / NPH : 1;
  float4 rb[RBN];
  auto issue_one = [&](int buf, int c, const float* cb, int kc, int q) {
    const bool tail = c == c_tail;
    const int I = min(q * NW + wave, 35), gk = I / 3, j = I - 3 * gk;
    const int gks = BT ? (gk & (tail ? 4 * nt_tail - 1 : 15)) : gk;
    float* dst = smem + buf * PIECE + I * 256;
    const float* src = BT ? cb + (int64_t)(gks * 8) * p.ldb + kc * RR_KP
                          : cb + (int64_t)(kc * RR_KP + gk * 8) * p.ldb;
    const unsigned loff = EPI == 1 ? lane_off_tiles(j, c) : lane_off(j, gk, tail ? 8 * nt_tail - 1 : 31);
    if (STAGE) {
      const float4 v = *reinterpret_cast<const float4*>(src + loff);
      rb[q % RBN].x = v.x; rb[q % RBN].y = v.y; rb[q % RBN].z = v.z; rb[q % RBN].w = v.w;
    } else {
      rr_glds16(src + loff, dst);
    }
  };
  auto park = [&](int buf, int half) {            // STAGE 1: registers -> LDS (unit 64 I + lane, linear: no conflicts)
#pragma unroll
    for (int e = 0; e < RBN; ++e) {
      const int q = half * RBN + e;
      if (q < NQ) {
        const int I = min(q * NW + wave, 35);
        *reinterpret_cast<float4*>(smem + buf * PIECE + I * 256 + 4 * lane) = rb[e];
      }
    }
  };

  // A rows of this wave -> registers.  The first piece only needs a[0..11]: they and the first B
  // piece are requested first, so the MFMAs start when a third of the block has arrived.
  float4 a[KG];
  const int arow_i = min(m0 + li, p.M - 1);
  const float* arow = p.A + (int64_t)arow_i * p.lda + 4 * lh;
#define RR_LOADA(T0, T1)                                                             \
  _Pragma("unroll") for (int t = (T0); t < (T1); ++t) {                              \
    const float4 v = *reinterpret_cast<const float4*>(arow + 8 * t);                 \
    a[t].x = v.x; a[t].y = v.y; a[t].z = v.z; a[t].w = v.w;                          \
  }
  RR_LOADA(0, 12)
  if (nloc > 0) {
    if (STAGE) {
#pragma unroll
      for (int h = 0; h < NPH; ++h) {
#pragma unroll
        for (int e = 0; e < RBN; ++e)
          if (h * RBN + e < NQ) issue_one(0, chunk_of(0), chunk_base(chunk_of(0)), 0, h * RBN + e);
        park(0, h);
      }
    } else {
#pragma unroll
      for (int q = 0; q < NQ; ++q) issue_one(0, chunk_of(0), chunk_base(chunk_of(0)), 0, q);
    }
  }
  RR_LOADA(12, KG)
#undef RR_LOADA
  const bool full = m0 + 32 <= p.M;

  f32x16 acc[3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // B fragment addressing (floats, relative to the piece)
  const int xl = (li >> 1) & 7;
  int bq[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) bq[q] = BT ? li * RR_KP + 4 * ((2 * q + lh) ^ xl) : 0;
  const int bnn = (4 * lh) * RR_NC + li;

  float m_run = -INFINITY, z_run = 0.f;             // EPI 4: running maximum / sum of exponentials of the lane's row
  int s = 0;
  for (int ci = 0; ci < nloc; ++ci) {
    const int c = chunk_of(ci);
    const int c_nxt = ci + 1 < nloc ? chunk_of(ci + 1) : c;    // after the last chunk: a redundant fetch into the idle buffer
    const float* cb = chunk_base(c);
    const float* cb_nxt = chunk_base(c_nxt);
#pragma unroll
    for (int kc = 0; kc < NPK; ++kc, ++s) {
      const int nb = (s + 1) & 1, nc = kc + 1 < NPK ? c : c_nxt, nk = kc + 1 < NPK ? kc + 1 : 0;
      const float* ncb = kc + 1 < NPK ? cb : cb_nxt;
      if (STAGE) {
        // a bare barrier: __syncthreads() carries a fence that drains vmcnt, i.e. would wait for loads in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      } else {
        if (s == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KG - 12) : "memory");   // a[12..] may still be in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                 // (bare: __syncthreads() would drain vmcnt to 0 again)
        asm volatile("" ::: "memory");
      }
      const bool more = !(ABLATE & 2) || s < 1;
      if (more && (ABLATE & 32)) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) issue_one(nb, nc, ncb, nk, q);
      }
      const float* Bp = smem + (s & 1) * PIECE;
      // B fragments one 8-row group ahead of the MFMAs that use them (two named sets, pinned by the
      // scheduling barriers: the compiler would otherwise hoist several groups and spill)
      float b0[3][4], b1[3][4];
#define RR_LOADB(BX, G)                                                                    \
  if (BT) {                                                                                \
    _Pragma("unroll") for (int j = 0; j < 3; ++j) {                                        \
      const float4 v = *reinterpret_cast<const float4*>(Bp + bq[(G) & 3] + 32 * ((G) >> 2) + j * (32 * RR_KP)); \
      BX[j][0] = v.x; BX[j][1] = v.y; BX[j][2] = v.z; BX[j][3] = v.w;                      \
    }                                                                                      \
  } else {                                                                                 \
    _Pragma("unroll") for (int j = 0; j < 3; ++j)                                          \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) BX[j][q] = Bp[bnn + (8 * (G) + q) * RR_NC + 32 * j]; \
  }
#define RR_MFMA(BX, G)                                                               \
  {                                                                                  \
    const float4 av = a[kc * 12 + (G)];                                              \
    _Pragma("unroll") for (int j = 0; j < 3; ++j) acc[j] = EPI == 4 ? __builtin_amdgcn_mfma_f32_32x32x2f32(BX[j][0], av.x, acc[j], 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, BX[j][0], acc[j], 0, 0, 0); \
    _Pragma("unroll") for (int j = 0; j < 3; ++j) acc[j] = EPI == 4 ? __builtin_amdgcn_mfma_f32_32x32x2f32(BX[j][1], av.y, acc[j], 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, BX[j][1], acc[j], 0, 0, 0); \
    _Pragma("unroll") for (int j = 0; j < 3; ++j) acc[j] = EPI == 4 ? __builtin_amdgcn_mfma_f32_32x32x2f32(BX[j][2], av.z, acc[j], 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, BX[j][2], acc[j], 0, 0, 0); \
    _Pragma("unroll") for (int j = 0; j < 3; ++j) acc[j] = EPI == 4 ? __builtin_amdgcn_mfma_f32_32x32x2f32(BX[j][3], av.w, acc[j], 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, BX[j][3], acc[j], 0, 0, 0); \
  }
      // slot t (one per 12 MFMAs) issues DMA instructions PER t .. PER t + PER - 1 of the next piece
      constexpr int PER = NW == 4 ? 2 : 1;
#define RR_SLOT(T)                                                                   \
  if (STAGE) {                                                                       \
    if ((T) % WIN < RBN && ((T) / WIN) * RBN + (T) % WIN < NQ) issue_one(nb, nc, ncb, nk, ((T) / WIN) * RBN + (T) % WIN); \
    if ((T) % WIN == WIN - 1 && (T) / WIN < NPH - 1) park(nb, (T) / WIN);            \
  } else if (!(ABLATE & 34) || (more && !(ABLATE & 32))) {                           \
    _Pragma("unroll") for (int e = 0; e < PER; ++e)                                  \
      if (PER * (T) + e < NQ) issue_one(nb, nc, ncb, nk, PER * (T) + e);             \
  }
      if (STAGE && !BT) {
        // (NN, register-staged: one fragment set -- the partner wave of the SIMD covers the LDS latency; the second
        //  set is what the staging registers are paid with)
#pragma unroll
        for (int g = 0; g < 12; ++g) {
          RR_LOADB(b0, g)
          __builtin_amdgcn_sched_barrier(0);
          RR_MFMA(b0, g)
          __builtin_amdgcn_sched_barrier(0);
          RR_SLOT(g)
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        RR_LOADB(b0, 0)
#pragma unroll
        for (int g = 0; g < 12; g += 2) {
          RR_LOADB(b1, g + 1)
          __builtin_amdgcn_sched_barrier(0);
          RR_MFMA(b0, g)
          __builtin_amdgcn_sched_barrier(0);
          RR_SLOT(g)
          __builtin_amdgcn_sched_barrier(0);
          if (g + 2 < 12) { RR_LOADB(b0, g + 2) }
          __builtin_amdgcn_sched_barrier(0);
          RR_MFMA(b1, g + 1)
          __builtin_amdgcn_sched_barrier(0);
          RR_SLOT(g + 1)
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#undef RR_LOADB
#undef RR_SLOT
#undef RR_MFMA
      if (STAGE) park(nb, NPH - 1);                 // every wave is past this piece's barrier: that buffer is idle
    }
    // ---- store the finished 32 x 96 block of this wave -------------------------------------
    if constexpr (EPI == 1) {
      float* park = smem + 2 * PIECE + wave * 1024 + lane;
      if (full) rr_store_swiglu<false>(p, acc, m0, c, li, lh, park); else rr_store_swiglu<true>(p, acc, m0, c, li, lh, park);
    } else if constexpr (EPI == 2) {
      if (p.epi_ablate & 4) { for (int j = 0; j < 3; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f; }
      else if (full) rr_store_swiglu_bwd<false>(p, acc, m0, c, li, lh); else rr_store_swiglu_bwd<true>(p, acc, m0, c, li, lh);
    } else if constexpr (EPI == 3) {
      if (full) rr_store_rope<false>(p, acc, m0, c, li, lh); else rr_store_rope<true>(p, acc, m0, c, li, lh);
    } else if constexpr (EPI == 4) {
      const int nt = c == c_tail ? nt_tail : 3;
      if (full) rr_store_lse<false>(p, acc, m0, c, li, lh, nt, m_run, z_run);
      else rr_store_lse<true>(p, acc, m0, c, li, lh, nt, m_run, z_run);
    } else if constexpr (EPI == 5) {
      const int nt = c == c_tail ? nt_tail : 3;
      if (!(p.epi_ablate & 2)) rr_rowmax(p, acc, c, li, nt, m_run);
      rr_store(p, acc, m0, c, li, lh, full, nt);
    } else {
      rr_store(p, acc, m0, c, li, lh, full, c == c_tail ? nt_tail : 3);
    }
  }
  if constexpr (EPI == 4) {
    const float z = z_run + __shfl_xor(z_run, 32, 64);
    if (lh == 0 && m0 + li < p.M) p.lse[m0 + li] = m_run + logf(z);
  }
  if constexpr (EPI == 5) {                          // lane l: register row l >> 1 = row (r & 3) + 8 (r >> 2) + 4 h of the block
    // (chunks split over grid.y when the rows alone do not fill the chip: one vector of maxima per range, part y at
    //  lse + y * M -- the consumer takes the maximum over the parts)
    const int r = li >> 1, row = m0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (!(li & 1) && row < p.M) p.lse[(int64_t)blockIdx.y * p.M + row] = m_run;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the redundant last fetch must not outlive the workgroup's LDS
}

// K = 288 (the model width of the benchmarked Llama), N a multiple of 32 and >= 96, A rows contiguous,
// B rows contiguous (either orientation), 16-byte aligned rows.
// 1 when the tile-piece kernel would take a plain (M x N) projection (pdn_gemm_f32 then also sends it the 288-wide
// products and the ones with a residual, which the chunk kernel loses to the tiled kernel)
int pdn_rowtile_plain_ok(int M, int N, int b_trans) {
  RowTileArgs ta;
  memset(&ta, 0, sizeof(ta));
  ta.M = M; ta.N = N; ta.lda = 288; ta.ldb = b_trans ? 288 : N; ta.ldc = N; ta.b_trans = b_trans; ta.nblocks = 1;
  return pdn_rowtile_takes(ta);
}
extern "C" int pdn_gemm_rowres_supported(int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int b_trans) {
  return K == 288 && N % 32 == 0 && N >= RR_NC && M >= 1 && lda % 4 == 0 && ldb % 4 == 0 && lda >= K &&
         ldb >= (b_trans ? K : N) && ldc >= N && (int64_t)32 * ldc < (1ll << 30);
}

// B as `nblocks` matrices side by side (block b at B + b * b_block_stride floats; NN: each (K x N / nblocks),
// NT: each (N / nblocks x K)); nblocks > 1 needs N / nblocks to be a multiple of 96.
// `epi` (may be null): fused epilogue request, EPI of the kernel template + its operands.
int pdn_gemm_prof_begin(int family, double flops, double bytes, void* stream);    // csrc/gemm.hip: bench.py's per-family timing
void pdn_gemm_prof_end(int token, void* stream);
struct RowResEpi {
  int kind;                 // 1 SwiGLU forward, 2 SwiGLU backward, 3 RoPE
  float* H; int64_t ldh;    // 1
  const float* GU;          // 2
  int F;                    // 1, 2
  const float* rope; int L, hd, rope_cols;   // 3
  unsigned g_off = 0, u_off = 0;              // 1
  float* lse = nullptr;                       // 4, 5
  int parts = 0;                              // 5, plan only (lse == nullptr): the number of chunk ranges = vectors of maxima
  // RMSNorm folded into the A load (kinds 1 and 3, tile-piece kernel only): norm weight, eps, outputs y (M x 288) and rms (M)
  const float* norm_w = nullptr; float* xn = nullptr; float* rms = nullptr; int64_t ldxn = 0; float norm_eps = 0.f;
};

static int rowres_launch(const float* A, const float* B, float* C, const float* bias, const float* residual, int M,
                         int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int b_trans, int nblocks,
                         int64_t b_block_stride, void* stream, const RowResEpi* epi = nullptr) {
  if (M == 0 || N == 0) return PDN_OK;
  PDN_CHECK_ARG(A && B && C, "pdn_gemm_rowres_f32: null operand");
  const int nper = nblocks > 0 ? N / nblocks : 0;
  if (nblocks < 1 || nper * nblocks != N || !pdn_gemm_rowres_supported(M, nper, K, lda, ldb, ldc, b_trans) ||
      ldc < N || (nblocks > 1 && (nper % RR_NC != 0 || b_block_stride % 4 != 0))) {
    pdn_set_error("pdn_gemm_rowres_f32: unsupported shape M=%d N=%d K=%d blocks=%d (K 288, N a multiple of 32 >= 96 "
                  "-- of 96 per block when B comes in blocks --, leading dimensions multiples of 4)", M, N, K, nblocks);
    return PDN_EUNSUPPORTED;
  }
  PDN_CHECK_ARG(((((uintptr_t)A | (uintptr_t)B) & 15) == 0), "pdn_gemm_rowres_f32: 16-byte alignment required");
  // round 5: the tile-piece kernel (csrc/gemm_rowtile.hip) takes every shape that gives each CU an 8-wave workgroup --
  // its stores and epilogue reads leave under the next tile's MFMAs instead of in a store phase of their own
  // (a residual that ALIASES C stays on the chunk kernel: the tile-piece kernel's first, dummy drain writes the place of the
  //  workgroup's last tile before that tile's residual rows have been read)
  const bool res_alias = residual && (const float*)C < residual + (int64_t)M * ldc && residual < (const float*)C + (int64_t)M * ldc;
  // (the same for a SwiGLU backward run IN PLACE, d[gate | up] over the saved [gate | up]: the chunk kernel reads a
  //  position before it writes it; the tile-piece kernel's dummy first drain would not)
  const bool gu_alias = epi && epi->kind == 2 && epi->GU && (const float*)C < epi->GU + (int64_t)M * ldc &&
                        epi->GU < (const float*)C + (int64_t)M * ldc;
  if (epi && epi->norm_w && epi->xn && A < epi->xn + (int64_t)M * epi->ldxn && epi->xn < A + (int64_t)M * lda) {
    pdn_set_error("row-resident projection with the RMSNorm folded in: xn must not overlap x (other workgroups read the same rows)");
    return PDN_EINVAL;
  }
  if ((!residual || !epi) && !res_alias && !gu_alias) {
    RowTileArgs ta;
    memset(&ta, 0, sizeof(ta));
    ta.A = A; ta.B = B; ta.C = C; ta.bias = bias; ta.residual = residual; ta.M = M; ta.N = N; ta.lda = lda; ta.ldb = ldb; ta.ldc = ldc;
    ta.b_trans = b_trans; ta.nblocks = nblocks; ta.b_block_stride = b_block_stride; ta.epi = epi ? epi->kind : 0;
    if (epi) {
      ta.H = epi->H; ta.ldh = epi->ldh; ta.GU = epi->GU; ta.F = epi->F; ta.rope = epi->rope; ta.L = epi->L; ta.hd = epi->hd;
      ta.rope_cols = epi->rope_cols; ta.g_off = epi->g_off; ta.u_off = epi->u_off; ta.lse = epi->lse;
      ta.parts = &const_cast<RowResEpi*>(epi)->parts;
      ta.norm_w = epi->norm_w; ta.xn = epi->xn; ta.rms = epi->rms; ta.ldxn = epi->ldxn; ta.norm_eps = epi->norm_eps;
    }
    if (pdn_rowtile_takes(ta)) return pdn_rowtile_launch(ta, stream);
  }
  if (epi && epi->norm_w) {
    pdn_set_error("row-resident projection with the RMSNorm folded in: only the tile-piece kernel has it (M=%d too small)", M);
    return PDN_EUNSUPPORTED;
  }
  RowResParams p{A, B, C, bias, residual, M, N, lda, ldb, ldc, (N + RR_NC - 1) / RR_NC, 0, 0, b_block_stride};
  p.cpb = nblocks > 1 ? nper / RR_NC : p.chunks;
  p.H = nullptr; p.GU = nullptr; p.rope = nullptr; p.ldh = 0; p.F = 0; p.L = 1; p.hd = 1; p.rope_chunks = 0; p.hd_magic = 0; p.g_off = 0; p.u_off = 0; p.lse = nullptr;
  const int kind = epi ? epi->kind : 0;
  if (kind) {
    p.H = epi->H; p.ldh = epi->ldh; p.GU = epi->GU; p.F = epi->F;
    p.rope = reinterpret_cast<const float2*>(epi->rope); p.L = epi->L; p.hd = epi->hd;
    p.rope_chunks = epi->rope_cols / RR_NC;
    p.hd_magic = (unsigned)(((1ull << 32) + (unsigned)epi->hd - 1) / (unsigned)epi->hd);
    p.g_off = epi->g_off; p.u_off = epi->u_off;
    p.lse = epi->lse;
  }
  static const int s_pdn_rowres_epi_ablate = pdn_ablation_switch("PDN_ROWRES_EPI_ABLATE");
  p.epi_ablate = s_pdn_rowres_epi_ablate;
  hipStream_t st = (hipStream_t)stream;
  static const int ablate = getenv("PDN_ROWRES_ABLATE") ? atoi(getenv("PDN_ROWRES_ABLATE")) : 0;
  static const int nw_env = getenv("PDN_ROWRES_NW") ? atoi(getenv("PDN_ROWRES_NW")) : 0;
  static const int stage_env = getenv("PDN_ROWRES_STAGE") ? atoi(getenv("PDN_ROWRES_STAGE")) : -1;
  // register-staged B in 8-wave workgroups: NT 72.9 vs 68.6 % at N = 768, 87.9 vs 75.8 % at N = 32000; NN (staged in
  // three phases with a single fragment set, which is what fits in 256 registers) 85.7 vs 84.5 % at N = 32000, 76.1 vs
  // 74.8 % at 1536, 73.3 vs 71.7 % at 864.  The 4-wave form (fewer rows than fill the chip) keeps the LDS-DMA.
  // one 8-wave workgroup per CU (half the fetch instructions per wave) as long as that fills the chip -- with the
  // column chunks split over grid.y if need be (round 3: at 8192 / 16384 / 32768 rows the 8-wave form measured
  // 74 / 83 / 85 % on the lm_head forward against 70 / 77 / 79 % for two 4-wave workgroups per CU, 1-4 points
  // on the layer projections: the threshold of round 2, 49152 rows, only looked at the row blocks)
  const int rb8 = (M + 255) / 256;
  const int nw = kind ? 8 : nw_env ? nw_env : (rb8 >= 192 || (int64_t)rb8 * p.chunks >= 256) ? 8 : 4;
  const int stage = kind ? 1 : stage_env >= 0 ? stage_env : (nw == 8 ? 1 : 0);
  const int row_blocks = (M + 32 * nw - 1) / (32 * nw), target = nw == 4 ? 512 : 256;
  // fill every CU (two 4-wave or one 8-wave workgroup each): split the chunks over grid.y
  int nsplit = 1;
  while (row_blocks * nsplit < target && nsplit < p.chunks) ++nsplit;
  p.chunks_per_wg = (p.chunks + nsplit - 1) / nsplit;
  if (kind == 1) p.chunks_per_wg += p.chunks_per_wg & 1;       // gate / up tiles meet inside a PAIR of chunks
  if (kind == 4) p.chunks_per_wg = p.chunks;                   // a row's statistics are one workgroup's
  nsplit = (p.chunks + p.chunks_per_wg - 1) / p.chunks_per_wg;
  if (kind == 5 && !epi->lse) { const_cast<RowResEpi*>(epi)->parts = nsplit; return PDN_OK; }
  const dim3 grid(row_blocks, nsplit);
#define RR_LAUNCH(BT_, NW_, AB_) if (stage) hipLaunchKernelGGL((gemm_rowres_kernel<36, BT_, NW_, 1, 0>), grid, dim3(NW_ * 64), 0, st, p); else hipLaunchKernelGGL((gemm_rowres_kernel<36, BT_, NW_, 0, AB_>), grid, dim3(NW_ * 64), 0, st, p)
  if (kind == 1) {
    hipLaunchKernelGGL((gemm_rowres_kernel<36, false, 8, 1, 0, 1>), grid, dim3(512), 0, st, p);
  } else if (kind == 2) {
    hipLaunchKernelGGL((gemm_rowres_kernel<36, true, 8, 1, 0, 2>), grid, dim3(512), 0, st, p);
  } else if (kind == 3) {
    hipLaunchKernelGGL((gemm_rowres_kernel<36, false, 8, 1, 0, 3>), grid, dim3(512), 0, st, p);
  } else if (kind == 4) {
    hipLaunchKernelGGL((gemm_rowres_kernel<36, false, 8, 1, 0, 4>), grid, dim3(512), 0, st, p);
  } else if (kind == 5) {
    hipLaunchKernelGGL((gemm_rowres_kernel<36, false, 8, 1, 0, 5>), grid, dim3(512), 0, st, p);
  } else if (nw == 8) {
    if (b_trans) RR_LAUNCH(true, 8, 0); else RR_LAUNCH(false, 8, 0);
  } else if (b_trans) {
    RR_LAUNCH(true, 4, 0);
  } else if (ablate == 2) {
    RR_LAUNCH(false, 4, 2);
  } else if (ablate == 32) {
    RR_LAUNCH(false, 4, 32);
  } else {
    RR_LAUNCH(false, 4, 0);
  }
#undef RR_LAUNCH
  pdn_count(PDN_CNT_ROWRES_CHUNK);
  if (kind) pdn_count(PDN_CNT_ROWRES_CHUNK_EPI);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

// ---- fused-epilogue entry points (include/pdn_hip.h) ------------------------------------------------------------
// gu (M x 2F) = x (M x 288) [Wg | Wu]  and  h (M x F) = silu(gate) * up in the same launch.  Wg, Wu: (288 x F) row-major,
// `w_stride` floats apart.  F a multiple of 96.
extern "C" int pdn_gateup_swiglu_supported(int M, int F, int K) {
  return (K == 288 && F % RR_NC == 0 && F >= RR_NC && M >= 1 && (int64_t)64 * F < (1ll << 29)) ? 1 : 0;
}
extern "C" int pdn_gateup_swiglu_fwd_f32(const float* x, const float* w_gate, int64_t w_stride, float* gu, float* h, int M,
                                         int F, int K, int64_t ldx, void* stream) {
  if (M == 0 || F == 0) return PDN_OK;
  PDN_CHECK_ARG(x && w_gate && gu && h, "pdn_gateup_swiglu_fwd_f32: null operand");
  const int64_t ws = w_stride < 0 ? -w_stride : w_stride;      // (the up matrix may sit below the gate matrix)
  if (!pdn_gateup_swiglu_supported(M, F, K) || ws % 4 != 0 || ws < (int64_t)K * F || ws + (int64_t)K * F >= (1ll << 30)) {
    pdn_set_error("pdn_gateup_swiglu_fwd_f32: unsupported shape M=%d F=%d K=%d (K 288, F a multiple of 96)", M, F, K);
    return PDN_EUNSUPPORTED;
  }
  RowResEpi e{1, h, F, nullptr, F, nullptr, 1, 1, 0};
  e.g_off = w_stride < 0 ? (unsigned)ws : 0u;
  e.u_off = w_stride < 0 ? 0u : (unsigned)ws;
  const float* wbase = w_stride < 0 ? w_gate + w_stride : w_gate;
  const int tk = pdn_gemm_prof_begin(5, 2.0 * M * (2.0 * F) * K, 4.0 * ((double)M * K + 2.0 * K * F + 3.0 * M * F), stream);
  const int rc = rowres_launch(x, wbase, gu, nullptr, nullptr, M, 2 * F, K, ldx, F, 2 * F, 0, 2, ws, stream, &e);
  pdn_gemm_prof_end(tk, stream);
  return rc;
}
// The same with the RMSNorm in front of the projection folded into the A load (nn/modules/norm.py:221-248 feeding
// llm/llama/model.py:56-58): x = the rows BEFORE the norm; xn (M x K) = x / sqrt(mean(x^2) + eps) * norm_w and rms (M) are left
// for the backward (the weight gradient contracts xn, the norm's backward wants x and rms).  Tile-piece kernel only.
static int rowtile_would_take(int M, int N, int epi, int F) {
  RowTileArgs ta;
  memset(&ta, 0, sizeof(ta));
  ta.M = M; ta.N = N; ta.lda = 288; ta.ldb = N; ta.ldc = N; ta.nblocks = 1; ta.epi = epi; ta.F = F;
  return pdn_rowtile_takes(ta);
}
extern "C" int pdn_gateup_swiglu_norm_supported(int M, int F, int K) {
  return (pdn_gateup_swiglu_supported(M, F, K) && rowtile_would_take(M, 2 * F, 1, F)) ? 1 : 0;
}
extern "C" int pdn_gateup_swiglu_norm_fwd_f32(const float* x, const float* norm_w, float eps, float* xn, float* rms,
                                              const float* w_gate, int64_t w_stride, float* gu, float* h, int M, int F,
                                              int K, int64_t ldx, void* stream) {
  if (M == 0 || F == 0) return PDN_OK;
  PDN_CHECK_ARG(x && norm_w && xn && rms && w_gate && gu && h, "pdn_gateup_swiglu_norm_fwd_f32: null operand");
  const int64_t ws = w_stride < 0 ? -w_stride : w_stride;
  if (!pdn_gateup_swiglu_norm_supported(M, F, K) || ws % 4 != 0 || ws < (int64_t)K * F || ws + (int64_t)K * F >= (1ll << 30)) {
    pdn_set_error("pdn_gateup_swiglu_norm_fwd_f32: unsupported shape M=%d F=%d K=%d", M, F, K);
    return PDN_EUNSUPPORTED;
  }
  RowResEpi e{1, h, F, nullptr, F, nullptr, 1, 1, 0};
  e.g_off = w_stride < 0 ? (unsigned)ws : 0u;
  e.u_off = w_stride < 0 ? 0u : (unsigned)ws;
  e.norm_w = norm_w; e.xn = xn; e.rms = rms; e.ldxn = K; e.norm_eps = eps;
  const float* wbase = w_stride < 0 ? w_gate + w_stride : w_gate;
  const int tk = pdn_gemm_prof_begin(5, 2.0 * M * (2.0 * F) * K, 4.0 * (2.0 * M * K + 2.0 * K * F + 3.0 * M * F), stream);
  const int rc = rowres_launch(x, wbase, gu, nullptr, nullptr, M, 2 * F, K, ldx, F, 2 * F, 0, 2, ws, stream, &e);
  pdn_gemm_prof_end(tk, stream);
  return rc;
}
// dgu (M x 2F) = SwiGLU'(gu) applied to dh = dy (M x 288) W_down^T, W_down (F x 288) row-major; dh is never written.
extern "C" int pdn_swiglu_bwd_gemm_f32(const float* dy, const float* w_down, const float* gu, float* dgu, int M, int F, int K,
                                       int64_t ldy, void* stream) {
  if (M == 0 || F == 0) return PDN_OK;
  PDN_CHECK_ARG(dy && w_down && gu && dgu, "pdn_swiglu_bwd_gemm_f32: null operand");
  if (!pdn_gateup_swiglu_supported(M, F, K)) {
    pdn_set_error("pdn_swiglu_bwd_gemm_f32: unsupported shape M=%d F=%d K=%d (K 288, F a multiple of 96)", M, F, K);
    return PDN_EUNSUPPORTED;
  }
  RowResEpi e{2, nullptr, 0, gu, F, nullptr, 1, 1, 0};
  const int tk = pdn_gemm_prof_begin(5, 2.0 * M * (double)F * K, 4.0 * ((double)M * K + (double)K * F + 4.0 * M * F), stream);
  const int rc = rowres_launch(dy, w_down, dgu, nullptr, nullptr, M, F, K, ldy, K, 2 * F, 1, 1, 0, stream, &e);
  pdn_gemm_prof_end(tk, stream);
  return rc;
}
// qkv (M x 3D) = x (M x 288) [Wq | Wk | Wv] with RoPE applied to the q and k column blocks: row m is position m % L,
// `rope` the (L x hd x 2) table of pdn_rope_table_f32.  D a multiple of 96, hd even, L a multiple of 32.
extern "C" int pdn_qkv_rope_supported(int M, int D, int K, int L, int hd) {
  return (K == 288 && D % RR_NC == 0 && hd >= 32 && hd % 2 == 0 && D % hd == 0 && 3 * D < 65536 && L % 32 == 0 && L > 0 && M % L == 0) ? 1 : 0;
}
extern "C" int pdn_qkv_rope_fwd_f32(const float* x, const float* wq, int64_t w_stride, float* qkv, const float* rope, int M,
                                    int D, int K, int L, int hd, int64_t ldx, void* stream) {
  if (M == 0 || D == 0) return PDN_OK;
  PDN_CHECK_ARG(x && wq && qkv && rope && (((uintptr_t)rope & 7) == 0), "pdn_qkv_rope_fwd_f32: null / misaligned operand");
  if (!pdn_qkv_rope_supported(M, D, K, L, hd) || w_stride % 4 != 0) {
    pdn_set_error("pdn_qkv_rope_fwd_f32: unsupported shape M=%d D=%d K=%d L=%d hd=%d", M, D, K, L, hd);
    return PDN_EUNSUPPORTED;
  }
  RowResEpi e{3, nullptr, 0, nullptr, 0, rope, L, hd, 2 * D};
  const int tk = pdn_gemm_prof_begin(5, 2.0 * M * (3.0 * D) * K, 4.0 * ((double)M * K + 3.0 * K * D + 3.0 * M * D), stream);
  const int rc = rowres_launch(x, wq, qkv, nullptr, nullptr, M, 3 * D, K, ldx, D, 3 * D, 0, 3, w_stride, stream, &e);
  pdn_gemm_prof_end(tk, stream);
  return rc;
}

// q | k | v + RoPE with the RMSNorm in front folded into the A load (see pdn_gateup_swiglu_norm_fwd_f32)
extern "C" int pdn_qkv_rope_norm_supported(int M, int D, int K, int L, int hd) {
  return (pdn_qkv_rope_supported(M, D, K, L, hd) && rowtile_would_take(M, 3 * D, 3, 0)) ? 1 : 0;
}
extern "C" int pdn_qkv_rope_norm_fwd_f32(const float* x, const float* norm_w, float eps, float* xn, float* rms, const float* wq,
                                         int64_t w_stride, float* qkv, const float* rope, int M, int D, int K, int L, int hd,
                                         int64_t ldx, void* stream) {
  if (M == 0 || D == 0) return PDN_OK;
  PDN_CHECK_ARG(x && norm_w && xn && rms && wq && qkv && rope && (((uintptr_t)rope & 7) == 0),
                "pdn_qkv_rope_norm_fwd_f32: null / misaligned operand");
  if (!pdn_qkv_rope_norm_supported(M, D, K, L, hd) || w_stride % 4 != 0) {
    pdn_set_error("pdn_qkv_rope_norm_fwd_f32: unsupported shape M=%d D=%d K=%d L=%d hd=%d", M, D, K, L, hd);
    return PDN_EUNSUPPORTED;
  }
  RowResEpi e{3, nullptr, 0, nullptr, 0, rope, L, hd, 2 * D};
  e.norm_w = norm_w; e.xn = xn; e.rms = rms; e.ldxn = K; e.norm_eps = eps;
  const int tk = pdn_gemm_prof_begin(5, 2.0 * M * (3.0 * D) * K, 4.0 * (2.0 * M * K + 3.0 * K * D + 3.0 * M * D), stream);
  const int rc = rowres_launch(x, wq, qkv, nullptr, nullptr, M, 3 * D, K, ldx, D, 3 * D, 0, 3, w_stride, stream, &e);
  pdn_gemm_prof_end(tk, stream);
  return rc;
}

// logits (M x V) = x (M x 288) W (288 x V) + bias and lse[m] = log sum_v exp(logits[m][v]) in ONE launch (EPI 4).
// V a multiple of 32 (>= 96); enough rows for one 8-wave workgroup per CU to own whole rows (M >= 49152), else
// PDN_EUNSUPPORTED and the caller runs the product and pdn_cross_entropy_fwd_f32.
extern "C" int pdn_linear_lse_supported(int64_t M, int V, int K) {
  return (K == 288 && V % 32 == 0 && V >= RR_NC && M >= 49152 && M < (1ll << 31)) ? 1 : 0;
}
extern "C" int pdn_linear_lse_fwd_f32(const float* x, const float* w, const float* bias, float* logits, float* lse, int M,
                                      int V, int K, int64_t ldx, int64_t ldw, int64_t ldl, void* stream) {
  if (M == 0 || V == 0) return PDN_OK;
  PDN_CHECK_ARG(x && w && logits && lse, "pdn_linear_lse_fwd_f32: null operand");
  if (!pdn_linear_lse_supported(M, V, K) || (ldl & 3) || ((uintptr_t)logits & 15) || ((uintptr_t)bias & 15)) {
    pdn_set_error("pdn_linear_lse_fwd_f32: unsupported shape M=%d V=%d K=%d", M, V, K);
    return PDN_EUNSUPPORTED;
  }
  RowResEpi e{4, nullptr, 0, nullptr, 0, nullptr, 1, 1, 0};
  e.lse = lse;
  const int tk = pdn_gemm_prof_begin(2, 2.0 * M * (double)V * K, 0.0, stream);
  const int rc = rowres_launch(x, w, logits, bias, nullptr, M, V, K, ldx, ldw, ldl, 0, 1, 0, stream, &e);
  pdn_gemm_prof_end(tk, stream);
  return rc;
}

// The same product leaving the row maxima of the logits only (EPI 5): the first half of a cross entropy whose sum of
// exponentials comes out of the input-gradient product (pdn_linear_ce_dx_deferred_f32).  With few rows the column chunks
// are split into `pdn_linear_rowmax_parts` ranges over the grid: `rowmax` holds that many vectors of M maxima (part p at
// rowmax + p * M), the maximum of a row is the maximum over the parts.
extern "C" int pdn_linear_rowmax_supported(int64_t M, int V, int K) {
  return (K == 288 && V % 32 == 0 && V >= RR_NC && M >= 1 && M < (1ll << 31)) ? 1 : 0;
}
extern "C" int pdn_linear_rowmax_parts(int64_t M, int V, int K) {
  if (!pdn_linear_rowmax_supported(M, V, K)) return 0;
  RowResEpi e{5, nullptr, 0, nullptr, 0, nullptr, 1, 1, 0};
  alignas(16) static float dummy[4];                 // (plan only: nothing is launched, the operands are never touched)
  if (rowres_launch(dummy, dummy, dummy, nullptr, nullptr, (int)M, V, K, K, V, V, 0, 1, 0, nullptr, &e) != PDN_OK) return 0;
  return e.parts;
}
extern "C" int pdn_linear_rowmax_fwd_f32(const float* x, const float* w, const float* bias, float* logits, float* rowmax,
                                         int M, int V, int K, int64_t ldx, int64_t ldw, int64_t ldl, void* stream) {
  if (M == 0 || V == 0) return PDN_OK;
  PDN_CHECK_ARG(x && w && logits && rowmax, "pdn_linear_rowmax_fwd_f32: null operand");
  if (!pdn_linear_rowmax_supported(M, V, K) || (ldl & 3) || ((uintptr_t)logits & 15) || ((uintptr_t)bias & 15)) {
    pdn_set_error("pdn_linear_rowmax_fwd_f32: unsupported shape M=%d V=%d K=%d", M, V, K);
    return PDN_EUNSUPPORTED;
  }
  RowResEpi e{5, nullptr, 0, nullptr, 0, nullptr, 1, 1, 0};
  e.lse = rowmax;
  const int tk = pdn_gemm_prof_begin(2, 2.0 * M * (double)V * K, 0.0, stream);
  const int rc = rowres_launch(x, w, logits, bias, nullptr, M, V, K, ldx, ldw, ldl, 0, 1, 0, stream, &e);
  pdn_gemm_prof_end(tk, stream);
  return rc;
}

// (L x hd x 2) table for the RoPE epilogue from the reference's (L x hd/2) cos / sin tables (llm/llama/model.py:13-20):
// entry (pos, col) = (cos[pos][col / 2], col odd ? sin : -sin), so that out = v cos + pair(v) * entry.y
__global__ void rr_rope_table_kernel(const float* __restrict__ cs, const float* __restrict__ sn, float2* __restrict__ out,
                                     int total, int hd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int pos = i / hd, col = i - pos * hd;
  const float c = cs[pos * (hd / 2) + (col >> 1)], s = sn[pos * (hd / 2) + (col >> 1)];
  out[i] = make_float2(c, (col & 1) ? s : -s);
}
extern "C" int pdn_rope_table_f32(const float* cos_t, const float* sin_t, float* out, int L, int hd, void* stream) {
  if (L == 0 || hd == 0) return PDN_OK;
  PDN_CHECK_ARG(cos_t && sin_t && out && hd % 2 == 0 && (((uintptr_t)out & 7) == 0), "pdn_rope_table_f32: bad arguments");
  hipLaunchKernelGGL(rr_rope_table_kernel, dim3((L * hd + 255) / 256), dim3(256), 0, (hipStream_t)stream, cos_t, sin_t,
                     reinterpret_cast<float2*>(out), L * hd, hd);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

extern "C" int pdn_gemm_rowres_f32(const float* A, const float* B, float* C, const float* bias,
                                   const float* residual, int M, int N, int K, int64_t lda, int64_t ldb,
                                   int64_t ldc, int b_trans, void* stream) {
  return rowres_launch(A, B, C, bias, residual, M, N, K, lda, ldb, ldc, b_trans, 1, 0, stream);
}

// used by pdn_gemm_f32 for a batch that is really one product: the same A against `nblocks` equally spaced
// weight matrices, the results side by side in one packed buffer (fused QKV, gate | up)
int pdn_gemm_rowres_blocks(const float* A, const float* B, float* C, int M, int N, int K, int64_t lda, int64_t ldb,
                           int64_t ldc, int b_trans, int nblocks, int64_t b_block_stride, void* stream) {
  return rowres_launch(A, B, C, nullptr, nullptr, M, N, K, lda, ldb, ldc, b_trans, nblocks, b_block_stride, stream);
}
