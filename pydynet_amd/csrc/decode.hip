// Greedy KV-cache decode step (llm/llama/model.py:105-121 in eval mode + `generate`, model.py:254-269;
// the loop the reference times in llm/llama/infer.py:46-63), one new token per sequence.
//
// At batch 1 a step is ~60 MB of weight traffic (the whole model sits in the 256 MiB Infinity Cache after the first
// token) and round 2 spent 430 us on it: ~77 launches issued from Python, each a generic kernel.  These kernels are
// built so that ONE hipGraph holds the whole step and can be replayed for every token:
//   * nothing depends on a host value that changes between tokens -- the position is read from DEVICE memory
//     (`pos`, advanced by the last kernel of the step), the picked token's embedding row is written where the next
//     step starts, the token itself goes to a history slot that may be mapped host memory;
//   * few launches per transformer block: this file holds the skinny product with its fused stagings (RMSNorm /
//     SwiGLU / merge of attention partials / sum of hand-off records, decode_stage.h), the attention over the cache with
//     the optional per-head output projection, and the pick; decode_layer.hip the feed-forward half of a block in one
//     launch, decode_block.hip the attention half (2 launches per block; 3 and 5 are the fallbacks).
// What these kernels are tuned for is the latency chain, not bandwidth: DESIGN.md 4.10 lists what was measured from
// inside them (tools/decode_trace.sh).
//
// Skinny product y (B x N) = a (B x K) @ W (K x N), B <= a few rows, W row-major (in, out) as the reference
// stores it (nn/modules/linear.py:26-27).  HBM/L2-bound on W and latency-bound at these sizes, so the design
// goal is "every CU has a slice and every load is 16 bytes": a workgroup owns TN columns (16 / 32 / 64, chosen by
// the host so that small N still spreads over >= ~50 CUs), thread = (k-slice, column quad); the k-slices are
// summed by wave shuffles and then across the four waves through LDS in a FIXED order (no atomics: the result
// does not depend on timing, and greedy decoding must give the same tokens every run).
#include "common.h"
#include "decode_stage.h"

#define DEC_MAX_B 8

template <int TN, int NB, int P, bool EXACT>
__global__ __launch_bounds__(256) void decode_gemv_kernel(
    const float* __restrict__ x, const float* __restrict__ recs, int K, int R, const float* __restrict__ norm_w,
    const float* __restrict__ W, int64_t w_bs, int w_rs, int N,
    // ^ 14 dwords: in SGPRs at dispatch (kernarg preload)
    int blk_cols, int x_rs, float eps, const float* __restrict__ bias, const float* residual, int r_rs,
    float* y, int y_rs, int B, int act, int act_ns, int act_hd,
    float* __restrict__ blk_max, int* __restrict__ blk_arg, float* x_out, int recs_rs, int x_out_rs, int staged_) {
  // x is also the base row of the hand-off (decode_stage.h); staged_: the row goes through dec_stage_* (K % 4 == 0, K <= 1024)
  const DecSum sum{staged_ ? x : nullptr, recs, x_out, x_rs, recs_rs, x_out_rs, R, 0, 0, 0};
  extern __shared__ __attribute__((aligned(16))) float xs[];        // [B][K] staged (normalised / gated) input rows
  __shared__ float red[16];
  __shared__ float ssq[4 * DEC_MAX_B];
  __shared__ float4 part[4][TN / 4][NB];
  __shared__ float cand_v[NB][TN / 4];
  __shared__ int cand_i[NB][TN / 4];
  constexpr int Q = TN / 4;                // column quads per workgroup
  constexpr int S = 256 / Q;               // k-slices per workgroup
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  const int quad = tid % Q, slice = tid / Q;
  DEC_T_BEGIN(TN == 64 ? 4 : 0);
  const int n = blockIdx.x * TN + 4 * quad;                         // first of this thread's four columns
  const bool live = n < N;
  const int blk = live ? n / blk_cols : 0, col = live ? n - blk * blk_cols : 0;
  const float* wp = W + (int64_t)blk * w_bs + col;                  // (threads past N read column 0: loaded, never stored)

  // ---- the input row first (a wave's loads return in issue order and the row is what the chain waits for), then,
  //      without waiting, the first P k-steps of this thread's column quad: the weights do not depend on the
  //      activation, so they travel while the row is summed and normalised (these kernels are latency chains:
  //      a token is ~20 dependent launches of a few microseconds each).  No load sits behind a branch: rows past K
  //      re-read row `slice` and are masked where they are used (32-bit offsets, two instructions per load) ----
  DecStage stg;
  const bool staged = sum.base != nullptr;
  if (staged) dec_stage_issue(sum, K, 0, norm_w, stg);
  // (a load instruction costs its 16 clocks of the CU's address path whether its lanes are useful or not -- four
  //  waves x a few dozen loads is most of a microsecond -- so steps past K are skipped by a uniform branch; they are
  //  the LAST loads, where the compiler's then conservative wait counts cost nothing)
  const int nsteps = (K + S - 1) / S;
  float4 wreg[P];
#pragma unroll
  for (int i = 0; i < P; ++i) {
    const int k = slice + i * S;
    if (!EXACT) wreg[i] = make_float4(0.f, 0.f, 0.f, 0.f);        // (EXACT: K needs exactly P k-steps)
    if (EXACT || i < nsteps) wreg[i] = *reinterpret_cast<const float4*>(wp + (unsigned)((k < K ? k : slice) * w_rs));
  }
  // epilogue operands of the finishing threads, likewise
  const int fq = tid % Q, fj = tid / Q, fn = blockIdx.x * TN + 4 * fq;
  const bool fin = tid < Q * NB && fn < N;
  float4 fbias = make_float4(0.f, 0.f, 0.f, 0.f);
  if (fin && bias) fbias = *reinterpret_cast<const float4*>(bias + fn);
  float4 fres = make_float4(0.f, 0.f, 0.f, 0.f);          // (row fj of the first pass; later passes load in place)
  if (fin && residual && fj < B) fres = *reinterpret_cast<const float4*>(residual + (unsigned)(fj * r_rs + fn));

  DEC_T(1);
  // ---- stage the input rows: RMSNorm (norm.py:245-248: x / sqrt(mean(x^2) + eps) * w) or SwiGLU
  //      (functional.py:39-40, model.py:56-58: g / (1 + exp(-g)) * u on packed [gate | up] rows) on the way ----
  if (staged) {
    // the rows arrive as base (+ records of the previous kernel's workgroups, decode_stage.h), then RMSNorm
    // (row 0 outside the loop: merged with the later rows' re-issue, the compiler's wait for the first use of the
    //  row would be "all but the last eight loads", i.e. nearly all the weights)
    dec_stage_row(sum, K, 0, stg, xs, xs + B * K, ssq, blockIdx.x == 0, norm_w != nullptr);
    for (int b = 1; b < B; ++b) {
      dec_stage_issue(sum, K, b, norm_w, stg);
      dec_stage_row(sum, K, b, stg, xs, xs + B * K, ssq, blockIdx.x == 0, norm_w != nullptr);
    }
  } else
  for (int b = 0; b < B; ++b) {
    const float* xr = x + (unsigned)(b * x_rs);
    if (act == 2) {
      // x row b = the NS partial results of pdn_decode_attention_f32 (key ranges of one query):
      // per (split, head) [m, l, pad, pad | o[hd]]; the row of the product is their softmax-weighted merge
      //   att[h] = sum_s exp(m_s - M) o_s / sum_s exp(m_s - M) l_s,  M = max_s m_s   (splits with no keys: l = 0)
      // The records come in with ONE round of independent loads (into LDS), then the merge reads LDS.
      const int NS = act_ns, hd = act_hd, rec = 4 + hd, H = K / hd, tot = NS * H * rec;
      float* raw = xs + B * K;
      lds_barrier();                     // (the previous row's merge is done with `raw`)
      for (int i = 4 * tid; i < tot; i += 1024)
        *reinterpret_cast<float4*>(raw + i) = *reinterpret_cast<const float4*>(xr + i);
      lds_barrier();
      // one thread per head turns (m, l) of its NS ranges into merge weights w_s = exp(m_s - M) / sum_s exp(m_s - M) l_s
      // (left in the record's pad slot), then every element is NS independent multiply-adds
      if (tid < H) {
        float M = -INFINITY;
        for (int sp = 0; sp < NS; ++sp) {
          const float* r = raw + (sp * H + tid) * rec;
          if (r[1] > 0.f) M = fmaxf(M, r[0]);
        }
        float den = 0.f;
        for (int sp = 0; sp < NS; ++sp) {
          float* r = raw + (sp * H + tid) * rec;
          const float w = r[1] > 0.f ? expf(r[0] - M) : 0.f;
          r[2] = w;
          den += w * r[1];
        }
        const float inv = 1.f / den;
        for (int sp = 0; sp < NS; ++sp) raw[(sp * H + tid) * rec + 2] *= inv;
      }
      lds_barrier();
      for (int k = tid; k < K; k += 256) {
        const int h = k / hd, d = k - h * hd;
        float num = 0.f;
        for (int sp = 0; sp < NS; ++sp) {
          const float* r = raw + (sp * H + h) * rec;
          num = fmaf(r[2], r[4 + d], num);
        }
        xs[b * K + k] = num;
      }
    } else if (act) {
      for (int k = tid; k < K; k += 256) {
        const float g = xr[k], u = xr[K + k];
        xs[b * K + k] = g / (1.f + expf(-g)) * u;
      }
    } else if (norm_w) {
      // K <= 512: one element pair per thread stays in registers between the two passes
      float v0 = 0.f, v1 = 0.f, w0 = 0.f, w1 = 0.f;
      const bool small = K <= 512;
      float ss = 0.f;
      if (small) {
        if (tid < K) { v0 = xr[tid]; w0 = norm_w[tid]; }
        if (tid + 256 < K) { v1 = xr[tid + 256]; w1 = norm_w[tid + 256]; }
        ss = v0 * v0 + v1 * v1;
      } else {
        for (int k = tid; k < K; k += 256) { const float v = xr[k]; ss += v * v; }
      }
      ss = block_sum_lds(ss, red);
      const float scale = 1.f / sqrtf(ss / (float)K + eps);
      if (small) {
        if (tid < K) xs[b * K + tid] = v0 * scale * w0;
        if (tid + 256 < K) xs[b * K + tid + 256] = v1 * scale * w1;
      } else {
        for (int k = tid; k < K; k += 256) xs[b * K + k] = xr[k] * scale * norm_w[k];
      }
      lds_barrier();                     // `red` is reused by the next row
    } else {
      for (int k = tid; k < K; k += 256) xs[b * K + k] = xr[k];
    }
  }
  lds_barrier();
  DEC_T(2);

  for (int b0 = 0; b0 < B; b0 += NB) {
    float4 acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    {
#pragma unroll
      for (int i = 0; i < P; ++i) {
        const int k = slice + i * S, kc = k < K ? k : slice;
        const float4 w = wreg[i];          // (zeros where the step was skipped)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          float a = xs[min(b0 + j, B - 1) * K + kc];
          a = (k < K && b0 + j < B) ? a : 0.f;
          acc[j].x = fmaf(a, w.x, acc[j].x); acc[j].y = fmaf(a, w.y, acc[j].y);
          acc[j].z = fmaf(a, w.z, acc[j].z); acc[j].w = fmaf(a, w.w, acc[j].w);
        }
      }
#pragma unroll 4
      for (int k = slice + P * S; k < K; k += S) {
        const float4 w = *reinterpret_cast<const float4*>(wp + (unsigned)(k * w_rs));
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const float a = (b0 + j < B) ? xs[(b0 + j) * K + k] : 0.f;
          acc[j].x = fmaf(a, w.x, acc[j].x); acc[j].y = fmaf(a, w.y, acc[j].y);
          acc[j].z = fmaf(a, w.z, acc[j].z); acc[j].w = fmaf(a, w.w, acc[j].w);
        }
      }
    }
    // k-slices of one wave: lanes with equal `quad` are Q apart
#pragma unroll
    for (int j = 0; j < NB; ++j) {
#pragma unroll
      for (int o = 32; o >= Q; o >>= 1) {
        acc[j].x += __shfl_xor(acc[j].x, o, 64); acc[j].y += __shfl_xor(acc[j].y, o, 64);
        acc[j].z += __shfl_xor(acc[j].z, o, 64); acc[j].w += __shfl_xor(acc[j].w, o, 64);
      }
    }
    DEC_T(3);
    lds_barrier();
    if (lane < Q) {
#pragma unroll
      for (int j = 0; j < NB; ++j) part[wave][lane][j] = acc[j];
    }
    lds_barrier();
    const int b = b0 + fj;
    const bool mine = fin && b < B;
    if (mine) {
      float4 r = part[0][fq][fj];
#pragma unroll
      for (int wv = 1; wv < 4; ++wv) {
        const float4 t = part[wv][fq][fj];
        r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w;
      }
      if (staged && norm_w) {              // (the row was staged without RMSNorm's scalar: decode_stage.h)
        const float sc_ = dec_norm_scale(ssq, b, K, eps);
        r.x *= sc_; r.y *= sc_; r.z *= sc_; r.w *= sc_;
      }
      r.x += fbias.x; r.y += fbias.y; r.z += fbias.z; r.w += fbias.w;
      if (residual) {
        const float4 t = b0 == 0 ? fres : *reinterpret_cast<const float4*>(residual + (unsigned)(b * r_rs + fn));
        r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w;
      }
      *reinterpret_cast<float4*>(y + (unsigned)(b * y_rs + fn)) = r;
      if (blk_max) {                       // first maximum of this thread's four columns
        float bv = r.x; int bi = fn;
        if (r.y > bv) { bv = r.y; bi = fn + 1; }
        if (r.z > bv) { bv = r.z; bi = fn + 2; }
        if (r.w > bv) { bv = r.w; bi = fn + 3; }
        cand_v[fj][fq] = bv; cand_i[fj][fq] = bi;
      }
    }
    if (blk_max) {
      // per row: first maximum over this workgroup's columns (quads ascend with the column index)
      lds_barrier();
      if (tid < NB && b0 + tid < B) {
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int q = 0; q < Q; ++q) {
          if (blockIdx.x * TN + 4 * q >= N) break;
          const float v = cand_v[tid][q];
          if (v > bv || bi == 0x7fffffff) { bv = v; bi = cand_i[tid][q]; }
        }
        blk_max[(int64_t)(b0 + tid) * gridDim.x + blockIdx.x] = bv;
        blk_arg[(int64_t)(b0 + tid) * gridDim.x + blockIdx.x] = bi;
      }
    }
  }
  DEC_T(4);
  DEC_T_END();
}

template <int TN, int P>
static int launch_gemv(int nb, dim3 grid, size_t shm, hipStream_t st, const float* x, int64_t x_rs, const float* norm_w,
                       float eps, const float* W, int64_t w_rs, int blk_cols, int64_t w_bs, const float* bias,
                       const float* residual, int64_t r_rs, float* y, int64_t y_rs, int B, int K, int N, int act,
                       int act_ns, int act_hd, float* blk_max, int* blk_arg, const DecSum& sum) {
  constexpr int S = 256 / (TN / 4);
  const bool exact = (K + S - 1) / S == P;     // (no conditional weight loads: see decode_block.hip)
#define DEC_GO(NB, EX)                                                                                                 \
  hipLaunchKernelGGL((decode_gemv_kernel<TN, NB, P, EX>), grid, dim3(256), shm, st, x, sum.recs, K, sum.R, norm_w, W,    \
                     w_bs, (int)w_rs, N, blk_cols, (int)x_rs, eps, bias, residual, (int)r_rs, y, (int)y_rs, B, act,     \
                     act_ns, act_hd, blk_max, blk_arg, sum.x_out, sum.recs_rs, sum.x_out_rs, sum.base != nullptr ? 1 : 0)
  if (exact) { if (nb == 1) DEC_GO(1, true); else if (nb == 2) DEC_GO(2, true); else DEC_GO(4, true); }
  else { if (nb == 1) DEC_GO(1, false); else if (nb == 2) DEC_GO(2, false); else DEC_GO(4, false); }
#undef DEC_GO
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

// y (B, N) = f(x) (B, K) @ W + bias + residual.   f = RMSNorm(norm_w, eps) when norm_w is given; act = 1: x rows are
// packed [gate | up] of width 2 K and f = silu(gate) * up; act = 2: x rows are the `act_ns` key-range partials of
// pdn_decode_attention_f32 ((act_ns, K / act_hd, 4 + act_hd) floats per row) and f merges them.  W: `N / blk_cols` column blocks of `blk_cols` columns,
// block j at W + j * w_bs, rows w_rs floats apart (one (K, N) matrix: blk_cols = N).  y must not alias x;
// residual may alias y.  blk_max / blk_arg (optional, B x pdn_decode_gemv_blocks(N) each): per row and per
// workgroup the first maximum of the workgroup's output columns and its column index -- the first half of a greedy
// pick over a wide vocabulary projection (pdn_decode_pick_tick_f32 finishes it).
extern "C" int pdn_decode_gemv_blocks(int N) { return N <= 4096 ? (N + 15) / 16 : (N <= 16384 ? (N + 31) / 32 : (N + 63) / 64); }

static int decode_gemv_impl(const float* x, int64_t x_row_stride, const float* norm_w, float eps, const float* W,
                            int64_t w_row_stride, int blk_cols, int64_t w_block_stride, const float* bias,
                            const float* residual, int64_t res_row_stride, float* y, int64_t y_row_stride, int B, int K,
                            int N, int act, int act_ns, int act_hd, float* blk_max, int* blk_arg, const DecSum& sum_in,
                            void* stream) {
  if (B == 0 || N == 0) return PDN_OK;
  PDN_CHECK_ARG((blk_max == nullptr) == (blk_arg == nullptr), "pdn_decode_gemv_f32: blk_max and blk_arg go together");
  PDN_CHECK_ARG(x && W && y && K > 0 && blk_cols > 0 && N % blk_cols == 0, "pdn_decode_gemv_f32: bad arguments");
  PDN_CHECK_ARG(B <= DEC_MAX_B && (size_t)B * K * 4 <= 64 * 1024, "pdn_decode_gemv_f32: B = %d rows of K = %d do not fit (B <= %d, B * K <= 16384)", B, K, DEC_MAX_B);
  PDN_CHECK_ARG(blk_cols % 4 == 0 && w_row_stride % 4 == 0 && w_block_stride % 4 == 0 && y_row_stride % 4 == 0 &&
                    res_row_stride % 4 == 0 &&
                    ((((uintptr_t)W | (uintptr_t)y | (uintptr_t)bias | (uintptr_t)residual) & 15) == 0),
                "pdn_decode_gemv_f32: columns in multiples of 4, 16-byte aligned operands");
  PDN_CHECK_ARG(act == 0 || norm_w == nullptr, "pdn_decode_gemv_f32: act and norm are exclusive");
  const int64_t lim = (int64_t)1 << 31;  // (the kernels do their row arithmetic in 32 bits)
  PDN_CHECK_ARG(w_row_stride >= 0 && (int64_t)K * w_row_stride < lim && (int64_t)B * (x_row_stride < 0 ? -x_row_stride : x_row_stride) * 2 < lim &&
                    x_row_stride >= 0 && (int64_t)B * y_row_stride < lim && y_row_stride >= 0 && res_row_stride >= 0 &&
                    (int64_t)B * res_row_stride < lim,
                "pdn_decode_gemv_f32: strides out of the 32-bit range of the kernel");
  PDN_CHECK_ARG(act != 2 || (act_ns > 0 && act_hd > 0 && K % act_hd == 0), "pdn_decode_gemv_f32: act 2 needs splits / head_dim");
  const int nb = B == 1 ? 1 : (B == 2 ? 2 : 4);
  DecSum sum = sum_in;
  if (act == 0 && !sum.base && K % 4 == 0 && K <= 1024 && x_row_stride % 4 == 0 && (((uintptr_t)x | (uintptr_t)norm_w) & 15) == 0)
    sum.base = x, sum.base_rs = (int)x_row_stride;       // (the staged path with no records)
  const size_t shm = ((size_t)B * K + (act == 2 ? (size_t)act_ns * (K / act_hd) * (4 + act_hd) : 0) +
                      (sum.R > 0 ? (size_t)dec_sum_scratch(K, sum.R, sum.hdr) : 0)) * sizeof(float);
  PDN_CHECK_ARG(shm <= 64 * 1024 && (act != 2 || (x_row_stride % 4 == 0 && act_hd % 4 == 0 && ((uintptr_t)x & 15) == 0)),
                "pdn_decode_gemv_f32: act 2 staging does not fit / is not 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  // narrow column tiles while N is small: a workgroup streams K * TN * 4 bytes, the chip has 256 CUs
  // (prefetch depth P: 12 k-steps of 64 slices cover K <= 768, 18 of 16 slices cover K <= 288 -- the Llama shapes)
  if (N <= 4096 && K <= 320)             // (five k-steps of 64 slices: the 288-wide Llama)
    return launch_gemv<16, 5>(nb, dim3((N + 15) / 16), shm, st, x, x_row_stride, norm_w, eps, W, w_row_stride, blk_cols,
                              w_block_stride, bias, residual, res_row_stride, y, y_row_stride, B, K, N, act, act_ns, act_hd, blk_max, blk_arg, sum);
  if (N <= 4096)
    return launch_gemv<16, 12>(nb, dim3((N + 15) / 16), shm, st, x, x_row_stride, norm_w, eps, W, w_row_stride, blk_cols,
                               w_block_stride, bias, residual, res_row_stride, y, y_row_stride, B, K, N, act, act_ns, act_hd, blk_max, blk_arg, sum);
  if (N <= 16384)
    return launch_gemv<32, 12>(nb, dim3((N + 31) / 32), shm, st, x, x_row_stride, norm_w, eps, W, w_row_stride, blk_cols,
                               w_block_stride, bias, residual, res_row_stride, y, y_row_stride, B, K, N, act, act_ns, act_hd, blk_max, blk_arg, sum);
  return launch_gemv<64, 18>(nb, dim3((N + 63) / 64), shm, st, x, x_row_stride, norm_w, eps, W, w_row_stride, blk_cols,
                             w_block_stride, bias, residual, res_row_stride, y, y_row_stride, B, K, N, act, act_ns, act_hd, blk_max, blk_arg, sum);
}

extern "C" int pdn_decode_gemv_f32(const float* x, int64_t x_row_stride, const float* norm_w, float eps, const float* W,
                                   int64_t w_row_stride, int blk_cols, int64_t w_block_stride, const float* bias,
                                   const float* residual, int64_t res_row_stride, float* y, int64_t y_row_stride,
                                   int B, int K, int N, int act, int act_ns, int act_hd, float* blk_max, int* blk_arg,
                                   void* stream) {
  DecSum none{};
  return decode_gemv_impl(x, x_row_stride, norm_w, eps, W, w_row_stride, blk_cols, w_block_stride, bias, residual,
                          res_row_stride, y, y_row_stride, B, K, N, act, act_ns, act_hd, blk_max, blk_arg, none, stream);
}

// The same product with its input rows handed over as  base + sum of `n_parts` partial rows  (decode_stage.h; the
// records pdn_decode_mlp_f32 leaves: row b's j-th partial at parts + b * parts_row_stride + j * K), then RMSNorm.
// x_out (optional): the summed rows, written once -- the base of the next hand-off.
extern "C" int pdn_decode_gemv_sum_f32(const float* base, int64_t base_row_stride, const float* parts, int n_parts,
                                       int64_t parts_row_stride, float* x_out, int64_t x_out_row_stride,
                                       const float* norm_w, float eps, const float* W, int64_t w_row_stride, int blk_cols,
                                       int64_t w_block_stride, const float* bias, float* y, int64_t y_row_stride, int B,
                                       int K, int N, float* blk_max, int* blk_arg, void* stream) {
  PDN_CHECK_ARG(base && parts && n_parts > 0 && K % 4 == 0 && K <= 1024 && base_row_stride % 4 == 0 &&
                    parts_row_stride % 4 == 0 && x_out_row_stride % 4 == 0 &&
                    ((((uintptr_t)base | (uintptr_t)parts | (uintptr_t)x_out | (uintptr_t)norm_w) & 15) == 0),
                "pdn_decode_gemv_sum_f32: K %% 4 == 0, K <= 1024, 16-byte aligned rows");
  const int64_t lim = (int64_t)1 << 31;
  PDN_CHECK_ARG(base_row_stride >= 0 && parts_row_stride >= 0 && x_out_row_stride >= 0 && (int64_t)B * base_row_stride < lim &&
                    (int64_t)B * parts_row_stride + (int64_t)n_parts * K < lim && (int64_t)B * x_out_row_stride < lim,
                "pdn_decode_gemv_sum_f32: strides out of the 32-bit range of the kernel");
  DecSum sum{base, parts, x_out, (int)base_row_stride, (int)parts_row_stride, (int)x_out_row_stride, n_parts, 0, 0, 0};
  return decode_gemv_impl(base, base_row_stride, norm_w, eps, W, w_row_stride, blk_cols, w_block_stride, bias, nullptr, 0,
                          y, y_row_stride, B, K, N, 0, 0, 0, blk_max, blk_arg, sum, stream);
}

// ---- RoPE of the new q / k rows + KV-cache append + decode attention (model.py:23-44, 105-121 with L = 1) ----------
// qkv: (B, 3 D) packed [q | k | v] rows of the fused projection.  A workgroup = (batch, head, key range): one CU
// pulls ~11 bytes per clock, and a head's K / V rows at a few hundred positions are ~100 KB, so the T = *pos + 1 keys
// of a head are cut into NS ranges handled by NS workgroups (flash-decoding).  A kernel of the decode step is a chain
// of memory round trips and what a CU can pull per microsecond (~26 KB): once the scalar load of *pos is back, EVERY
// load of the kernel is issued at once, most urgent first (a wave's loads return in issue order) -- the new q | k | v
// pairs and cos / sin of position *pos, the range's K rows (thread = key), its V rows (thread = (row group, column
// quad)), the head's rows of Wo -- and only rows that exist are fetched.
// Each workgroup rotates its head's q by the angle of position *pos (interleaved pairs (x[2i], x[2i+1])); the one
// whose range holds *pos also rotates k and appends k / v to cache row *pos (kept in LDS: the row is not re-read).
// Result per (range, head): [max score m, sum of exp l, -, - | sum of exp(s - m) v]; the merge over the NS ranges
// happens in the staging phase of the next kernel (pdn_decode_gemv_f32 act = 2, or decode_stage.h) -- no extra launch.
//
// OPROJ: the workgroup also multiplies its (unnormalised) partial result by ITS head's rows of the output projection
// (model.py:116: self.O(output); rows h * hd ... of Wo) and leaves a record [m, l, -, - | D values]
// (decode_stage.h): the projection's sum over heads, the merge of the key ranges and the residual add all happen in
// the staging of the next kernel -- one launch less per layer.  Those rows are 4 hd D bytes for ONE CU, so C
// workgroups share a (range, head): each repeats the attention (the K / V rows come out of L2) and owns D / C columns.
// KPRE / VPRE: float4s of the thread's K row / V rows of the thread held in registers from the start (head_dim 48:
// 12 / 13, 64: 16 / 16 -- a 256-key range completely; 0 / 0: any head_dim, loads where they are used).
template <int KPRE, int VPRE, bool OPROJ>
__global__ __launch_bounds__(256) void decode_attention_kernel(const int* __restrict__ pos_ptr, int H, int hd, int NS, int C,
                                                               const float* __restrict__ qkv, float* __restrict__ kc,
                                                               float* __restrict__ vc, const float* __restrict__ cs,
                                                               // ^ 14 dwords: in SGPRs at dispatch (kernarg preload)
                                                               const float* __restrict__ sn, const float* __restrict__ Wo,
                                                               float* __restrict__ part_out, int64_t qkv_rs, int64_t cbs,
                                                               float inv_sqrt, int wo_rs) {
  extern __shared__ __attribute__((aligned(16))) float sc[];      // [chunk] scores, then [groups + 8][hd] partial sums
  __shared__ __attribute__((aligned(16))) float qs[256], ks[256], vs[256];
  __shared__ float red[16];
  DEC_T_BEGIN(1);
  const int pos = *pos_ptr, T = pos + 1;
  const int ci = blockIdx.x % C, sp = (blockIdx.x / C) % NS, bh = blockIdx.x / (C * NS), b = bh / H, h = bh % H;
  const int tid = threadIdx.x;
  const int chunk = (T + NS - 1) / NS, t0 = sp * chunk, t1 = min(T, t0 + chunk);
  const int D = H * hd, f4 = hd / 4, half = hd / 2, rec = 4 + (OPROJ ? D : hd);
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const int Dc = D / C, nqd = Dc / 4;                              // this workgroup's columns of the projected row
  float* out = part_out + (((int64_t)b * NS + sp) * H + h) * rec;
  if (t0 >= t1) {                        // no keys in this range (uniform over the workgroup)
    if (tid == 0 && ci == 0) { out[0] = -INFINITY; out[1] = 0.f; }
    if (OPROJ) for (int i = tid; i < nqd; i += 256) reinterpret_cast<float4*>(out + 4 + ci * Dc)[i] = z4;
    return;
  }
  float* kb = kc + (int64_t)b * cbs + (int64_t)h * hd;
  float* vb = vc + (int64_t)b * cbs + (int64_t)h * hd;
  const bool owner = pos >= t0 && pos < t1;

  // ---- every load of the kernel, most urgent first; none behind a branch (threads without a row re-read row t0 /
  //      row 0 and the masks are applied where the values are used) ----
  const int hq = min(tid, half - 1);                               // the new token's q / k / v pair of thread tid < hd / 2
  const float* row = qkv + (int64_t)b * qkv_rs + (unsigned)(h * hd + 2 * hq);
  const float2 nq = *reinterpret_cast<const float2*>(row);
  const float rc = cs[(unsigned)(pos * half + hq)], rs = sn[(unsigned)(pos * half + hq)];
  const float2 nk = *reinterpret_cast<const float2*>(row + D);
  const float2 nv = *reinterpret_cast<const float2*>(row + 2 * D);
  float4 kreg[KPRE > 0 ? KPRE : 1];
  const int kt = t0 + tid;                                         // the key whose score this thread computes first
  // (a load instruction costs its 16 clocks of the CU's address path whether its lanes are useful or not -- four
  //  waves x ~45 loads is the "issue" microsecond of these kernels -- so whole-wave skips are uniform branches)
  if (KPRE > 0 && t0 + (tid & ~63) < t1) {                         // (KPRE = hd / 4; this wave has a key)
    const float* kp = kb + (unsigned)(((kt < t1 && kt != pos) ? kt : t0) * D);
#pragma unroll
    for (int c = 0; c < KPRE; ++c) kreg[c] = *reinterpret_cast<const float4*>(kp + 4 * c);
  }
  const int groups = dec_div(256, f4), tg = dec_div(tid, f4), vc4 = tid - tg * f4;
  float4 vreg[VPRE > 0 ? VPRE : 1];
  if (VPRE > 0) {
    const int nv_rows = __builtin_amdgcn_readfirstlane((t1 - t0 + groups - 1) / groups);     // row steps with any key
#pragma unroll
    for (int i = 0; i < VPRE; ++i) {
      const int t = t0 + tg + i * groups;
      if (i < nv_rows)
        vreg[i] = *reinterpret_cast<const float4*>(vb + (unsigned)(((t < t1 && t != pos) ? t : t0) * D + 4 * vc4));
    }
  }
  // this head's rows of the output projection: thread = (row slice, column quad), PW rows each
  constexpr int PW = 16;
  const int G = OPROJ ? dec_div(256, nqd) : 1, osl = dec_div(tid, nqd), oq = tid - osl * nqd;
  float4 wo[OPROJ ? PW : 1];
  if (OPROJ) {
    const float* wop = Wo + (unsigned)(h * hd * wo_rs + ci * Dc + 4 * oq);
    const int nw_rows = __builtin_amdgcn_readfirstlane((hd + G - 1) / G);                    // row steps with any row
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      const int d = osl + i * G;
      if (i < nw_rows) wo[i] = *reinterpret_cast<const float4*>(wop + (unsigned)((d < hd ? d : 0) * wo_rs));
    }
  }
  DEC_T(1);

  if (tid < half) {
    *reinterpret_cast<float2*>(qs + 2 * tid) = make_float2(nq.x * rc - nq.y * rs, nq.x * rs + nq.y * rc);
    if (owner) {
      const float2 kr = make_float2(nk.x * rc - nk.y * rs, nk.x * rs + nk.y * rc);
      *reinterpret_cast<float2*>(ks + 2 * tid) = kr;
      *reinterpret_cast<float2*>(vs + 2 * tid) = nv;
      if (ci == 0) {
        *reinterpret_cast<float2*>(kb + (int64_t)pos * D + 2 * tid) = kr;
        *reinterpret_cast<float2*>(vb + (int64_t)pos * D + 2 * tid) = nv;
      }
    }
  }
  lds_barrier();
  DEC_T(2);
  const float4* q4 = reinterpret_cast<const float4*>(qs);
  float m = -INFINITY;
  for (int t = kt; t < t1; t += 256) {
    float s = 0.f;
    if (KPRE > 0 && t == kt && t != pos) {
#pragma unroll
      for (int c = 0; c < KPRE; ++c) {
        const float4 a = q4[c], k = kreg[c];
        s += (a.x * k.x + a.y * k.y) + (a.z * k.z + a.w * k.w);
      }
    } else {
      const float4* k4 = t == pos ? reinterpret_cast<const float4*>(ks) : reinterpret_cast<const float4*>(kb + (int64_t)t * D);
      for (int c = 0; c < f4; ++c) {
        const float4 a = q4[c], k = k4[c];
        s += (a.x * k.x + a.y * k.y) + (a.z * k.z + a.w * k.w);
      }
    }
    s *= inv_sqrt;
    sc[t - t0] = s;
    m = fmaxf(m, s);
  }
  m = block_max_lds(m, red);
  DEC_T(3);
  float l = 0.f;
  for (int t = kt; t < t1; t += 256) {
    const float pr = expf(sc[t - t0] - m);
    sc[t - t0] = pr;
    l += pr;
  }
  l = block_sum_lds(l, red);                 // (its barriers also publish the probabilities)
  DEC_T(4);
  float4 acc = z4;
  if (tg < groups) {
    if (VPRE > 0) {
#pragma unroll
      for (int i = 0; i < VPRE; ++i) {
        const int t = t0 + tg + i * groups;
        if (t < t1) {
          const float pr = sc[t - t0];
          const float4 v = t == pos ? reinterpret_cast<const float4*>(vs)[vc4] : vreg[i];
          acc.x += pr * v.x; acc.y += pr * v.y; acc.z += pr * v.z; acc.w += pr * v.w;
        }
      }
    }
    for (int t = t0 + tg + VPRE * groups; t < t1; t += groups) {
      const float pr = sc[t - t0];
      const float4 v = t == pos ? reinterpret_cast<const float4*>(vs)[vc4]
                                : *reinterpret_cast<const float4*>(vb + (int64_t)t * D + 4 * vc4);
      acc.x += pr * v.x; acc.y += pr * v.y; acc.z += pr * v.z; acc.w += pr * v.w;
    }
  }
  DEC_T(5);
  lds_barrier();                       // scores are dead: reuse the buffer for the partial sums
  float4* part = reinterpret_cast<float4*>(sc);
  if (tg < groups) part[tg * f4 + vc4] = acc;
  lds_barrier();
  // combine in a fixed order: 8 threads per column quad add every 8th group, then one thread adds those 8
  if (tid < 8 * f4) {
    const int g0 = dec_div(tid, f4);
    float4 r = z4;
    for (int g = g0; g < groups; g += 8) { const float4 t = part[g * f4 + vc4]; r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w; }
    part[groups * f4 + tid] = r;
  }
  lds_barrier();
  if (tid < f4) {
    float4 r = part[groups * f4 + tid];
    for (int g = 1; g < 8; ++g) { const float4 t = part[groups * f4 + g * f4 + tid]; r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w; }
    if (OPROJ) reinterpret_cast<float4*>(qs)[tid] = r;            // (q is dead: the head's hd sums go there)
    else reinterpret_cast<float4*>(out + 4)[tid] = r;
    if (tid == 0 && ci == 0) { out[0] = m; out[1] = l; }
  }
  DEC_T(6);
  if (OPROJ) {
    lds_barrier();
    float4 oacc = z4;
    if (osl < G) {
#pragma unroll
      for (int i = 0; i < PW; ++i) {
        const int d = osl + i * G;
        if (i < (hd + G - 1) / G) {        // (uniform: the same row steps that were loaded)
          float a = qs[d < hd ? d : 0];
          a = d < hd ? a : 0.f;
          oacc.x = fmaf(a, wo[i].x, oacc.x); oacc.y = fmaf(a, wo[i].y, oacc.y);
          oacc.z = fmaf(a, wo[i].z, oacc.z); oacc.w = fmaf(a, wo[i].w, oacc.w);
        }
      }
      for (int d = osl + PW * G; d < hd; d += G) {
        const float4 w = *reinterpret_cast<const float4*>(Wo + (unsigned)((h * hd + d) * wo_rs + ci * Dc + 4 * oq));
        const float a = qs[d];
        oacc.x = fmaf(a, w.x, oacc.x); oacc.y = fmaf(a, w.y, oacc.y); oacc.z = fmaf(a, w.z, oacc.z); oacc.w = fmaf(a, w.w, oacc.w);
      }
      part[osl * nqd + oq] = oacc;       // (the combine above is done with `part`: the barrier before this block)
    }
    lds_barrier();
    if (tid < nqd) {
      float4 r = part[tid];
      for (int g = 1; g < G; ++g) { const float4 t = part[g * nqd + tid]; r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w; }
      reinterpret_cast<float4*>(out + 4 + ci * Dc)[tid] = r;
    }
  }
  DEC_T(7);
  DEC_T_END();
}

// qkv rows `qkv_row_stride` floats apart; partials: (B, n_splits, H, 4 + head_dim) floats; caches hold `max_len` positions per sequence, `cache_batch_stride` floats between sequences; cos / sin
// tables (max_len, head_dim / 2).
static int decode_attention_impl(const float* qkv, int64_t qkv_row_stride, const float* cos_table, const float* sin_table,
                                 float* k_cache, float* v_cache, float* partials, int B, int H, int head_dim, int n_splits,
                                 int64_t cache_batch_stride, const int* pos, int max_len, const float* Wo,
                                 int64_t wo_row_stride, bool oproj, void* stream) {
  if (B == 0 || H == 0) return PDN_OK;
  PDN_CHECK_ARG(qkv && cos_table && sin_table && k_cache && v_cache && partials && pos && max_len > 0,
                "pdn_decode_attention_f32: bad arguments");
  PDN_CHECK_ARG(head_dim % 4 == 0 && head_dim <= 256 && (cache_batch_stride % 4) == 0 && qkv_row_stride % 4 == 0 &&
                    ((((uintptr_t)qkv | (uintptr_t)k_cache | (uintptr_t)v_cache | (uintptr_t)partials) & 15) == 0),
                "pdn_decode_attention_f32: head_dim %% 4, 16-byte alignment required");
  const int NS = n_splits;
  PDN_CHECK_ARG(NS >= 1 && NS <= 64, "pdn_decode_attention_f32: n_splits = %d", NS);
  const int f4 = head_dim / 4, groups = 256 / f4;
  const size_t need = (size_t)(groups + 8) * head_dim, chunk = (size_t)(max_len + NS - 1) / NS;
  const size_t shm = sizeof(float) * (chunk > need ? chunk : need);
  PDN_CHECK_ARG(shm <= 60 * 1024, "pdn_decode_attention_f32: max_len = %d too long", max_len);
  const float inv_sqrt = 1.f / sqrtf((float)head_dim);
  const int D = H * head_dim;
  int C = 1;                             // workgroups sharing a (range, head): D / C columns of the projected row each
  if (oproj) {
    PDN_CHECK_ARG(Wo && D <= 1024 && wo_row_stride % 4 == 0 && (((uintptr_t)Wo) & 15) == 0 && NS * H <= 256,
                  "pdn_decode_attention_oproj_f32: D <= 1024, n_splits * H <= 256, 16-byte aligned Wo rows");
    C = D % 16 == 0 ? 4 : (D % 12 == 0 ? 3 : (D % 8 == 0 ? 2 : 1));
  }
  PDN_CHECK_ARG((int64_t)max_len * D < ((int64_t)1 << 31) && wo_row_stride >= 0 && (int64_t)D * wo_row_stride < ((int64_t)1 << 31),
                "pdn_decode_attention_f32: cache rows / Wo out of the 32-bit range of the kernel");
  const dim3 grid(B * H * NS * C);
  hipStream_t st = (hipStream_t)stream;
#define ATT_GO(KP, VP, OP)                                                                                              \
  hipLaunchKernelGGL((decode_attention_kernel<KP, VP, OP>), grid, dim3(256), shm, st, pos, H, head_dim, NS, C, qkv,      \
                     k_cache, v_cache, cos_table, sin_table, Wo, partials, qkv_row_stride, cache_batch_stride, inv_sqrt, \
                     (int)wo_row_stride)
  if (head_dim == 48) { if (oproj) ATT_GO(12, 13, true); else ATT_GO(12, 13, false); }
  else if (head_dim == 64) { if (oproj) ATT_GO(16, 16, true); else ATT_GO(16, 16, false); }
  else { if (oproj) ATT_GO(0, 0, true); else ATT_GO(0, 0, false); }
#undef ATT_GO
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

extern "C" int pdn_decode_attention_f32(const float* qkv, int64_t qkv_row_stride, const float* cos_table,
                                        const float* sin_table, float* k_cache, float* v_cache, float* partials, int B,
                                        int H, int head_dim, int n_splits, int64_t cache_batch_stride, const int* pos,
                                        int max_len, void* stream) {
  return decode_attention_impl(qkv, qkv_row_stride, cos_table, sin_table, k_cache, v_cache, partials, B, H, head_dim,
                               n_splits, cache_batch_stride, pos, max_len, nullptr, 0, false, stream);
}

// The same with the output projection applied per head: records (B, n_splits, H, 4 + D) of [m, l, -, - | the
// head's unnormalised contribution to the projected row] for the staging of pdn_decode_mlp_f32.  Wo: (D, D) as
// nn.Linear stores it (in, out), rows wo_row_stride floats apart.
extern "C" int pdn_decode_attention_oproj_f32(const float* qkv, int64_t qkv_row_stride, const float* cos_table,
                                              const float* sin_table, float* k_cache, float* v_cache, const float* Wo,
                                              int64_t wo_row_stride, float* records, int B, int H, int head_dim,
                                              int n_splits, int64_t cache_batch_stride, const int* pos, int max_len,
                                              void* stream) {
  return decode_attention_impl(qkv, qkv_row_stride, cos_table, sin_table, k_cache, v_cache, records, B, H, head_dim,
                               n_splits, cache_batch_stride, pos, max_len, Wo, wo_row_stride, true, stream);
}

// ---- greedy pick + position tick (model.py:262-268: logits[:, -1, :].argmax(-1, keepdims=True)) ------------------
// First maximum wins, as numpy.argmax.  Writes the ids where the next step's embedding gather reads them and, from
// block 0, advances *pos: this is the LAST kernel of a step, every reader of *pos in the step ran before it.
__global__ __launch_bounds__(1024) void decode_argmax_tick_kernel(const float* __restrict__ logits, int64_t rs, int V,
                                                                  int64_t* __restrict__ ids, int* __restrict__ pos) {
  __shared__ float bv[16];
  __shared__ int bi[16];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* row = logits + (int64_t)b * rs;
  float best = -INFINITY;
  int idx = 0x7fffffff;
  for (int i = tid; i < V; i += 1024) {
    const float v = row[i];
    if (v > best || (v == best && i < idx)) { best = v; idx = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(idx, o, 64);
    if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
  }
  if (lane == 0) { bv[wave] = best; bi[wave] = idx; }
  lds_barrier();
  if (tid == 0) {
    for (int w = 1; w < 16; ++w)
      if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
    ids[b] = idx == 0x7fffffff ? 0 : idx;            // (a row of NaNs picks 0, as numpy.argmax does)
    if (b == 0 && pos) *pos += 1;
  }
}

extern "C" int pdn_decode_argmax_tick_f32(const float* logits, int64_t row_stride, int B, int V, int64_t* next_ids, int* pos,
                                          void* stream) {
  if (B == 0) return PDN_OK;
  PDN_CHECK_ARG(logits && next_ids && V > 0, "pdn_decode_argmax_tick_f32: bad arguments");
  hipLaunchKernelGGL(decode_argmax_tick_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, logits, row_stride, V, next_ids,
                     pos);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

// ---- second half of the greedy pick over per-workgroup candidates (see pdn_decode_gemv_f32) + position tick -----
// ONE workgroup walks the B rows (B <= 8), so that "read *pos, then advance it" is race free.  Token b goes to ids[b]
// (where the next step's gather reads it) and, when a history is given, to (*hist)[*pos * B + b]: a per-position slot
// the host can fetch -- and hand to the caller as that token's own array -- while later steps already run.
// With an embedding table the picked token's row is copied to x_next[b] right away: the next step then starts at its
// first projection (one launch less per token; model.py:254-256 feeds next_id straight back into the embedding).
__global__ __launch_bounds__(256) void decode_pick_tick_kernel(const float* __restrict__ vals, const int* __restrict__ args,
                                                               int B, int n, int64_t* __restrict__ ids, int* __restrict__ pos,
                                                               int64_t* const* __restrict__ hist,
                                                               const float* __restrict__ emb, int64_t emb_rs, int D,
                                                               float* __restrict__ x_next) {
  __shared__ float bv[4];
  __shared__ int bi[4];
  __shared__ int64_t chosen;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  DEC_T_BEGIN(2);
  const int p = pos ? *pos : 0;
  int64_t* hrow = hist ? *hist + (int64_t)p * B : nullptr;
  for (int b = 0; b < B; ++b) {
    float best = -INFINITY;
    int idx = 0x7fffffff;
    // the first four candidates of the thread with ONE round of loads (n <= 1024: all of them), the rest in a loop
    float cv[4]; int ca[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = min(tid + 256 * u, n - 1);
      cv[u] = vals[(unsigned)(b * n + i)]; ca[u] = args[(unsigned)(b * n + i)];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (tid + 256 * u < n && (cv[u] > best || (cv[u] == best && ca[u] < idx))) { best = cv[u]; idx = ca[u]; }
    for (int i = tid + 1024; i < n; i += 256) {
      const float v = vals[(int64_t)b * n + i];
      const int a = args[(int64_t)b * n + i];
      if (v > best || (v == best && a < idx)) { best = v; idx = a; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(idx, o, 64);
      if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if (lane == 0) { bv[wave] = best; bi[wave] = idx; }
    lds_barrier();
    if (tid == 0) {
      for (int w = 1; w < 4; ++w)
        if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
      const int64_t tok = idx == 0x7fffffff ? 0 : idx;
      ids[b] = tok;
      if (hrow) __hip_atomic_store(hrow + b, tok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (may be host memory)
      chosen = tok;
    }
    lds_barrier();
    DEC_T(1);
    if (emb) {
      const float* row = emb + chosen * emb_rs;
      for (int d = tid; d < D; d += 256) x_next[(int64_t)b * D + d] = row[d];
      lds_barrier();                       // `chosen` is rewritten for the next row
    }
  }
  if (tid == 0 && pos) *pos = p + 1;
  DEC_T(2);
  DEC_T_END();
}
DEC_TRACE_DUMP(pdn_dec_trace_dump_step)

extern "C" int pdn_decode_pick_tick_f32(const float* blk_max, const int* blk_arg, int B, int n_blocks, int64_t* next_ids,
                                        int* pos, int64_t* const* history, const float* emb, int64_t emb_row_stride,
                                        int D, float* x_next, void* stream) {
  if (B == 0) return PDN_OK;
  PDN_CHECK_ARG(blk_max && blk_arg && next_ids && n_blocks > 0, "pdn_decode_pick_tick_f32: bad arguments");
  PDN_CHECK_ARG(!history || pos, "pdn_decode_pick_tick_f32: a history needs the position");
  PDN_CHECK_ARG(!emb || (x_next && D > 0), "pdn_decode_pick_tick_f32: an embedding table needs x_next and D");
  hipLaunchKernelGGL(decode_pick_tick_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, blk_max, blk_arg, B, n_blocks,
                     next_ids, pos, history, emb, emb_row_stride, D, x_next);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}
