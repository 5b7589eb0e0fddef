// Feed-forward half of a decode layer in ONE launch (llm/llama/model.py:47-58 FeedForward.forward on one new token,
// called from TransformerBlock.forward, model.py:118-121):
//     h   = x + attention output            <- arrives as base + per-head records (decode_stage.h)
//     out = h + down(silu(gate(n)) * up(n)),  n = RMSNorm(h)
// The down projection is a sum over the hidden units, so it splits by hidden unit exactly like gate / up do by
// column: workgroup j owns 32 hidden units -- their 32 gate and 32 up columns (a 64-column skinny product over the
// staged row), SwiGLU on those 32 values, then THEIR 32 rows of the down matrix -- and leaves its contribution to the
// output row as a plain record.  The next kernel (the following layer's q|k|v projection, or the vocabulary
// projection) adds the F / 32 records to h in a fixed order while it stages its input: pdn_decode_gemv_sum_f32.
// Per layer that is three launches (q|k|v, attention + output projection, this one) instead of five; a batch-1 token
// is launch latency, not bytes (60 MB of weights that sit in the Infinity Cache).
#include "common.h"
#include "decode_stage.h"

#define DEC_MAX_B 8
#define MLP_SL 32                          // hidden units per workgroup (32 columns = one 128-byte line per matrix row)

template <int NB, int PD, bool EXACT>
__global__ __launch_bounds__(256) void decode_mlp_kernel(const float* __restrict__ base, const float* __restrict__ recs, int D,
                                                         int R, const float* __restrict__ norm_w,
                                                         const float* __restrict__ Wg, const float* __restrict__ Wu,
                                                         const float* __restrict__ Wd,
                                                         // ^ 14 dwords: in SGPRs at dispatch (kernarg preload)
                                                         int w_rs, int wd_rs, int ns, int H, float eps, float* x_out,
                                                         float* __restrict__ parts, int base_rs, int recs_rs, int x_out_rs,
                                                         int parts_rs, int B) {
  const DecSum sum{base, recs, x_out, base_rs, recs_rs, x_out_rs, R, R > 0 ? 4 : 0, ns, H};
  extern __shared__ __attribute__((aligned(16))) float xs[];        // [B][D] staged rows, then scratch
  __shared__ float ssq[4 * DEC_MAX_B];
  __shared__ float4 part[4][16][NB];
  __shared__ float hb[DEC_MAX_B][MLP_SL];
  constexpr int Q = 16, S = 16, P = 18;    // (EXACT: D needs exactly P k-steps and PD rows of Wd per thread: no conditional loads)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int quad = tid % Q, slice = tid / Q, j = blockIdx.x;
  DEC_T_BEGIN(3);

  // ---- the row (base + records) first, then every weight this workgroup needs: a wave's loads return in issue
  //      order, the row is what the chain waits for, and the weights do not depend on it.  No load sits behind a
  //      branch (rows past the end re-read a valid row and are masked where they are used) ----
  DecStage stg;
  dec_stage_issue(sum, D, 0, norm_w, stg);
  const float* wp = (quad < 8 ? Wg : Wu) + (unsigned)(j * MLP_SL + 4 * (quad & 7));
  // (a load instruction costs its 16 clocks of the CU's address path whether its lanes are useful or not, so steps
  //  past the end are skipped by uniform branches; they are the LAST loads of their group)
  const int nsteps = (D + S - 1) / S;
  float4 wreg[P];
#pragma unroll
  for (int i = 0; i < P; ++i) {
    const int k = slice + i * S;
    if (!EXACT) wreg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (EXACT || i < nsteps) wreg[i] = *reinterpret_cast<const float4*>(wp + (unsigned)((k < D ? k : slice) * w_rs));
  }
  const int nq = D >> 2, G = dec_div(256, nq), dsl = dec_div(tid, nq), dq = tid - dsl * nq;
  const float* wdp = Wd + (unsigned)(j * MLP_SL * wd_rs + 4 * dq);
  const int ndsteps = __builtin_amdgcn_readfirstlane((MLP_SL + G - 1) / G);
  float4 wd[PD];
#pragma unroll
  for (int i = 0; i < PD; ++i) {
    const int r = dsl + i * G;
    if (!EXACT) wd[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (EXACT || i < ndsteps) wd[i] = *reinterpret_cast<const float4*>(wdp + (unsigned)((r < MLP_SL ? r : 0) * wd_rs));
  }
  DEC_T(1);

  // ---- h = base + records (the first workgroup leaves it in x_out), n = RMSNorm(h) ----
  float* scratch = xs + B * D;
  dec_stage_row(sum, D, 0, stg, xs, scratch, ssq, j == 0, true);          // (row 0 outside the loop: exact load waits)
  for (int b = 1; b < B; ++b) {
    dec_stage_issue(sum, D, b, norm_w, stg);
    dec_stage_row(sum, D, b, stg, xs, scratch, ssq, j == 0, true);
  }
  DEC_T(2);
  DEC_T(3);

  // ---- [gate | up] columns of this slice, SwiGLU (functional.py:39-40: x / (1 + exp(-x)); model.py:56-58) ----
  for (int b0 = 0; b0 < B; b0 += NB) {
    float4 acc[NB];
#pragma unroll
    for (int r = 0; r < NB; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const int k = slice + i * S, kc = k < D ? k : slice;
      const float4 w = wreg[i];            // (zeros where the step was skipped)
#pragma unroll
      for (int r = 0; r < NB; ++r) {
        float a = xs[min(b0 + r, B - 1) * D + kc];
        a = (k < D && b0 + r < B) ? a : 0.f;
        acc[r].x = fmaf(a, w.x, acc[r].x); acc[r].y = fmaf(a, w.y, acc[r].y);
        acc[r].z = fmaf(a, w.z, acc[r].z); acc[r].w = fmaf(a, w.w, acc[r].w);
      }
    }
#pragma unroll 4
    for (int k = slice + P * S; k < D; k += S) {
      const float4 w = *reinterpret_cast<const float4*>(wp + (unsigned)(k * w_rs));
#pragma unroll
      for (int r = 0; r < NB; ++r) {
        const float a = (b0 + r < B) ? xs[(b0 + r) * D + k] : 0.f;
        acc[r].x = fmaf(a, w.x, acc[r].x); acc[r].y = fmaf(a, w.y, acc[r].y);
        acc[r].z = fmaf(a, w.z, acc[r].z); acc[r].w = fmaf(a, w.w, acc[r].w);
      }
    }
#pragma unroll
    for (int r = 0; r < NB; ++r) {
#pragma unroll
      for (int o = 32; o >= Q; o >>= 1) {
        acc[r].x += __shfl_xor(acc[r].x, o, 64); acc[r].y += __shfl_xor(acc[r].y, o, 64);
        acc[r].z += __shfl_xor(acc[r].z, o, 64); acc[r].w += __shfl_xor(acc[r].w, o, 64);
      }
    }
    DEC_T(4);
    lds_barrier();
    if (lane < Q) {
#pragma unroll
      for (int r = 0; r < NB; ++r) part[wave][lane][r] = acc[r];
    }
    lds_barrier();
    if (tid < MLP_SL * NB) {
      const int i = tid % MLP_SL, r = tid / MLP_SL, b = b0 + r;
      if (b < B) {
        float g = 0.f, u = 0.f;
        for (int wv = 0; wv < 4; ++wv) {               // fixed order over the four waves
          g += reinterpret_cast<const float*>(&part[wv][i >> 2][r])[i & 3];
          u += reinterpret_cast<const float*>(&part[wv][8 + (i >> 2)][r])[i & 3];
        }
        const float sc_ = dec_norm_scale(ssq, b, D, eps);           // (RMSNorm's scalar, applied to the products)
        g *= sc_; u *= sc_;
        hb[b][i] = g / (1.f + expf(-g)) * u;
      }
    }
  }
  lds_barrier();
  DEC_T(5);

  // ---- this slice's rows of the down projection: thread = (row slice, column quad) ----
  float4* comb = reinterpret_cast<float4*>(scratch);
  for (int b = 0; b < B; ++b) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (dsl < G) {
#pragma unroll
      for (int i = 0; i < PD; ++i) {
        const int r = dsl + i * G;
        float a = hb[b][r < MLP_SL ? r : 0];
        a = r < MLP_SL ? a : 0.f;            // (wd: zeros where the step was skipped)
        acc.x = fmaf(a, wd[i].x, acc.x); acc.y = fmaf(a, wd[i].y, acc.y);
        acc.z = fmaf(a, wd[i].z, acc.z); acc.w = fmaf(a, wd[i].w, acc.w);
      }
      for (int r = dsl + PD * G; r < MLP_SL; r += G) {
        const float4 w = *reinterpret_cast<const float4*>(wdp + (unsigned)(r * wd_rs));
        const float a = hb[b][r];
        acc.x = fmaf(a, w.x, acc.x); acc.y = fmaf(a, w.y, acc.y); acc.z = fmaf(a, w.z, acc.z); acc.w = fmaf(a, w.w, acc.w);
      }
      comb[dsl * nq + dq] = acc;
    }
    lds_barrier();
    if (tid < nq) {
      float4 r = comb[tid];
      for (int g = 1; g < G; ++g) { const float4 t = comb[g * nq + tid]; r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w; }
      *reinterpret_cast<float4*>(parts + (unsigned)(b * parts_rs + j * D + 4 * tid)) = r;
    }
    lds_barrier();
  }
  DEC_T(6);
  DEC_T_END();
}
DEC_TRACE_DUMP(pdn_dec_trace_dump_layer)

// Number of partial rows pdn_decode_mlp_f32 leaves per sequence (0: this F is not taken).
extern "C" int pdn_decode_mlp_slices(int F) { return F > 0 && F % MLP_SL == 0 ? F / MLP_SL : 0; }

// base (B, D) rows; records (optional): (B, n_splits, H, 4 + D) from pdn_decode_attention_oproj_f32, rows
// records_row_stride floats apart; x_out (optional) receives h.  Wg / Wu: (D, F) gate / up matrices as nn.Linear
// stores them (in, out), rows w_row_stride floats apart; Wd: (F, D), rows wd_row_stride apart.  parts: (B, F / 32, D)
// partial rows, sequence b at parts + b * parts_row_stride.
extern "C" int pdn_decode_mlp_f32(const float* base, int64_t base_row_stride, const float* records,
                                  int64_t records_row_stride, int n_splits, int H, float* x_out, int64_t x_out_row_stride,
                                  const float* norm_w, float eps, const float* Wg, const float* Wu, int64_t w_row_stride,
                                  const float* Wd, int64_t wd_row_stride, float* parts, int64_t parts_row_stride, int B,
                                  int D, int F, void* stream) {
  if (B == 0) return PDN_OK;
  PDN_CHECK_ARG(base && norm_w && Wg && Wu && Wd && parts && D > 0, "pdn_decode_mlp_f32: bad arguments");
  PDN_CHECK_ARG(B <= DEC_MAX_B && D % 4 == 0 && D <= 1024 && F % MLP_SL == 0 && F > 0,
                "pdn_decode_mlp_f32: B <= %d, D %% 4 == 0, D <= 1024, F %% %d == 0 (B %d, D %d, F %d)", DEC_MAX_B, MLP_SL, B, D, F);
  PDN_CHECK_ARG(base_row_stride % 4 == 0 && records_row_stride % 4 == 0 && x_out_row_stride % 4 == 0 &&
                    w_row_stride % 4 == 0 && wd_row_stride % 4 == 0 && parts_row_stride % 4 == 0 &&
                    ((((uintptr_t)base | (uintptr_t)records | (uintptr_t)x_out | (uintptr_t)Wg | (uintptr_t)Wu |
                       (uintptr_t)Wd | (uintptr_t)parts | (uintptr_t)norm_w) & 15) == 0),
                "pdn_decode_mlp_f32: 16-byte aligned rows");
  PDN_CHECK_ARG(!records || (n_splits > 0 && H > 0 && n_splits * H <= 256), "pdn_decode_mlp_f32: n_splits * H <= 256");
  const int R = records ? n_splits * H : 0, G = 256 / (D / 4);
  const int64_t lim = (int64_t)1 << 31;  // (the kernel does its row arithmetic in 32 bits)
  PDN_CHECK_ARG(base_row_stride >= 0 && records_row_stride >= 0 && x_out_row_stride >= 0 && w_row_stride >= 0 &&
                    wd_row_stride >= 0 && parts_row_stride >= 0 && (int64_t)B * base_row_stride < lim &&
                    (int64_t)B * records_row_stride + (int64_t)R * (4 + D) < lim && (int64_t)B * x_out_row_stride < lim &&
                    (int64_t)D * w_row_stride < lim && (int64_t)F * wd_row_stride < lim &&
                    (int64_t)B * parts_row_stride + (int64_t)F / MLP_SL * D < lim,
                "pdn_decode_mlp_f32: strides out of the 32-bit range of the kernel");

  const int scratch = records ? dec_sum_scratch(D, R, 4) : 0;
  const size_t shm = sizeof(float) * ((size_t)B * D + (scratch > G * D ? scratch : G * D));
  PDN_CHECK_ARG(shm <= 64 * 1024, "pdn_decode_mlp_f32: rows do not fit in LDS");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(F / MLP_SL);
#define MLP_GO(NB, PD, EX)                                                                                              \
  hipLaunchKernelGGL((decode_mlp_kernel<NB, PD, EX>), grid, dim3(256), shm, st, base, records, D, R, norm_w, Wg, Wu, Wd, \
                     (int)w_row_stride, (int)wd_row_stride, n_splits, H, eps, x_out, parts, (int)base_row_stride,       \
                     (int)records_row_stride, (int)x_out_row_stride, (int)parts_row_stride, B)
  const bool exact = (D + 15) / 16 == 18 && (MLP_SL + G - 1) / G == 11;      // (D = 288: 18 k-steps, 11 rows of Wd)
  if (exact) { if (B == 1) MLP_GO(1, 11, true); else if (B == 2) MLP_GO(2, 11, true); else MLP_GO(4, 11, true); }
  else { if (B == 1) MLP_GO(1, 16, false); else if (B == 2) MLP_GO(2, 16, false); else MLP_GO(4, 16, false); }
#undef MLP_GO
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}
