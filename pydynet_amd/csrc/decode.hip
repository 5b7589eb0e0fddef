// Greedy KV-cache decode step (llm/llama/model.py:105-121 in eval mode + `generate`, model.py:254-269;
// the loop the reference times in llm/llama/infer.py:46-63), one new token per sequence.
//
// At batch 1 a step is ~40 MB of weight traffic (16 us at the attainable HBM rate; the whole model sits in the
// 256 MiB Infinity Cache after the first token) and round 2 spent 430 us on it: ~77 launches issued from Python,
// each a generic kernel.  These kernels are built so that ONE hipGraph holds the whole step and can be replayed
// for every token:
//   * nothing depends on a host value that changes between tokens -- the position is read from DEVICE memory
//     (`pos`, advanced by the last kernel of the step), the token ids are written where the next step's embedding
//     gather reads them;
//   * six launches per transformer block instead of thirteen: RMSNorm is applied while the activation row is staged
//     into LDS by the projection that consumes it (it is 288 floats: every workgroup redoes it), q | k | v and
//     gate | up are single skinny products over equally spaced weight blocks, RoPE + the KV-cache append are one
//     kernel, SwiGLU is applied in the loads of the down projection, residual adds ride in the epilogues.
//
// Skinny product y (B x N) = a (B x K) @ W (K x N), B <= a few rows, W row-major (in, out) as the reference
// stores it (nn/modules/linear.py:26-27).  HBM/L2-bound on W and latency-bound at these sizes, so the design
// goal is "every CU has a slice and every load is 16 bytes": a workgroup owns TN columns (16 / 32 / 64, chosen by
// the host so that small N still spreads over >= ~50 CUs), thread = (k-slice, column quad); the k-slices are
// summed by wave shuffles and then across the four waves through LDS in a FIXED order (no atomics: the result
// does not depend on timing, and greedy decoding must give the same tokens every run).
#include "common.h"

#define DEC_MAX_B 8

template <int TN, int NB, int P>
__global__ __launch_bounds__(256) void decode_gemv_kernel(
    const float* __restrict__ x, int64_t x_rs, const float* __restrict__ norm_w, float eps,
    const float* __restrict__ W, int64_t w_rs, int blk_cols, int64_t w_bs,
    const float* __restrict__ bias, const float* residual, int64_t r_rs,
    float* y, int64_t y_rs, int B, int K, int N, int act, int act_ns, int act_hd,
    float* __restrict__ blk_max, int* __restrict__ blk_arg) {
  extern __shared__ __attribute__((aligned(16))) float xs[];        // [B][K] staged (normalised / gated) input rows
  __shared__ float red[16];
  __shared__ float4 part[4][TN / 4][NB];
  __shared__ float cand_v[NB][TN / 4];
  __shared__ int cand_i[NB][TN / 4];
  constexpr int Q = TN / 4;                // column quads per workgroup
  constexpr int S = 256 / Q;               // k-slices per workgroup
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  const int quad = tid % Q, slice = tid / Q;
  const int n = blockIdx.x * TN + 4 * quad;                         // first of this thread's four columns
  const bool live = n < N;
  const int blk = live ? n / blk_cols : 0, col = live ? n - blk * blk_cols : 0;
  const float* wp = W + (int64_t)blk * w_bs + col;

  // ---- the weights do not depend on the activation: put the first P k-steps of this thread's column quad in flight
  //      BEFORE the input rows are staged, so the two memory latencies overlap (these kernels are latency chains:
  //      a token is ~35 dependent launches of a few microseconds each) ----
  float4 wreg[P];
#pragma unroll
  for (int i = 0; i < P; ++i) {
    const int k = slice + i * S;
    wreg[i] = (live && k < K) ? *reinterpret_cast<const float4*>(wp + (int64_t)k * w_rs) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // epilogue operands of the finishing threads, likewise
  const int fq = tid % Q, fj = tid / Q, fn = blockIdx.x * TN + 4 * fq;
  const bool fin = tid < Q * NB && fn < N;
  float4 fbias = make_float4(0.f, 0.f, 0.f, 0.f);
  if (fin && bias) fbias = *reinterpret_cast<const float4*>(bias + fn);
  float4 fres = make_float4(0.f, 0.f, 0.f, 0.f);          // (row fj of the first pass; later passes load in place)
  if (fin && residual && fj < B) fres = *reinterpret_cast<const float4*>(residual + (int64_t)fj * r_rs + fn);

  // ---- stage the input rows: RMSNorm (norm.py:245-248: x / sqrt(mean(x^2) + eps) * w) or SwiGLU
  //      (functional.py:39-40, model.py:56-58: g / (1 + exp(-g)) * u on packed [gate | up] rows) on the way ----
  for (int b = 0; b < B; ++b) {
    const float* xr = x + (int64_t)b * x_rs;
    if (act == 2) {
      // x row b = the NS partial results of pdn_decode_attention_f32 (key ranges of one query):
      // per (split, head) [m, l, pad, pad | o[hd]]; the row of the product is their softmax-weighted merge
      //   att[h] = sum_s exp(m_s - M) o_s / sum_s exp(m_s - M) l_s,  M = max_s m_s   (splits with no keys: l = 0)
      // The records come in with ONE round of independent loads (into LDS), then the merge reads LDS.
      const int NS = act_ns, hd = act_hd, rec = 4 + hd, H = K / hd, tot = NS * H * rec;
      float* raw = xs + B * K;
      __syncthreads();                     // (the previous row's merge is done with `raw`)
      for (int i = 4 * tid; i < tot; i += 1024)
        *reinterpret_cast<float4*>(raw + i) = *reinterpret_cast<const float4*>(xr + i);
      __syncthreads();
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
      __syncthreads();
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
      ss = block_sum(ss, red);
      const float scale = 1.f / sqrtf(ss / (float)K + eps);
      if (small) {
        if (tid < K) xs[b * K + tid] = v0 * scale * w0;
        if (tid + 256 < K) xs[b * K + tid + 256] = v1 * scale * w1;
      } else {
        for (int k = tid; k < K; k += 256) xs[b * K + k] = xr[k] * scale * norm_w[k];
      }
      __syncthreads();                     // `red` is reused by the next row
    } else {
      for (int k = tid; k < K; k += 256) xs[b * K + k] = xr[k];
    }
  }
  __syncthreads();

  for (int b0 = 0; b0 < B; b0 += NB) {
    float4 acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
#pragma unroll
      for (int i = 0; i < P; ++i) {
        const int k = slice + i * S;
        if (k < K) {
          const float4 w = wreg[i];
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            const float a = (b0 + j < B) ? xs[(b0 + j) * K + k] : 0.f;
            acc[j].x = fmaf(a, w.x, acc[j].x); acc[j].y = fmaf(a, w.y, acc[j].y);
            acc[j].z = fmaf(a, w.z, acc[j].z); acc[j].w = fmaf(a, w.w, acc[j].w);
          }
        }
      }
#pragma unroll 4
      for (int k = slice + P * S; k < K; k += S) {
        const float4 w = *reinterpret_cast<const float4*>(wp + (int64_t)k * w_rs);
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
    __syncthreads();
    if (lane < Q) {
#pragma unroll
      for (int j = 0; j < NB; ++j) part[wave][lane][j] = acc[j];
    }
    __syncthreads();
    const int b = b0 + fj;
    const bool mine = fin && b < B;
    if (mine) {
      float4 r = part[0][fq][fj];
#pragma unroll
      for (int wv = 1; wv < 4; ++wv) {
        const float4 t = part[wv][fq][fj];
        r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w;
      }
      r.x += fbias.x; r.y += fbias.y; r.z += fbias.z; r.w += fbias.w;
      if (residual) {
        const float4 t = b0 == 0 ? fres : *reinterpret_cast<const float4*>(residual + (int64_t)b * r_rs + fn);
        r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w;
      }
      *reinterpret_cast<float4*>(y + (int64_t)b * y_rs + fn) = r;
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
      __syncthreads();
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
}

template <int TN, int P>
static int launch_gemv(int nb, dim3 grid, size_t shm, hipStream_t st, const float* x, int64_t x_rs, const float* norm_w,
                       float eps, const float* W, int64_t w_rs, int blk_cols, int64_t w_bs, const float* bias,
                       const float* residual, int64_t r_rs, float* y, int64_t y_rs, int B, int K, int N, int act,
                       int act_ns, int act_hd, float* blk_max, int* blk_arg) {
#define DEC_GO(NB)                                                                                                     \
  hipLaunchKernelGGL((decode_gemv_kernel<TN, NB, P>), grid, dim3(256), shm, st, x, x_rs, norm_w, eps, W, w_rs, blk_cols, \
                     w_bs, bias, residual, r_rs, y, y_rs, B, K, N, act, act_ns, act_hd, blk_max, blk_arg)
  if (nb == 1) DEC_GO(1); else if (nb == 2) DEC_GO(2); else DEC_GO(4);
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

extern "C" int pdn_decode_gemv_f32(const float* x, int64_t x_row_stride, const float* norm_w, float eps, const float* W,
                                   int64_t w_row_stride, int blk_cols, int64_t w_block_stride, const float* bias,
                                   const float* residual, int64_t res_row_stride, float* y, int64_t y_row_stride,
                                   int B, int K, int N, int act, int act_ns, int act_hd, float* blk_max, int* blk_arg,
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
  PDN_CHECK_ARG(act != 2 || (act_ns > 0 && act_hd > 0 && K % act_hd == 0), "pdn_decode_gemv_f32: act 2 needs splits / head_dim");
  const int nb = B == 1 ? 1 : (B == 2 ? 2 : 4);
  const size_t shm = ((size_t)B * K + (act == 2 ? (size_t)act_ns * (K / act_hd) * (4 + act_hd) : 0)) * sizeof(float);
  PDN_CHECK_ARG(shm <= 64 * 1024 && (act != 2 || (x_row_stride % 4 == 0 && act_hd % 4 == 0 && ((uintptr_t)x & 15) == 0)),
                "pdn_decode_gemv_f32: act 2 staging does not fit / is not 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  // narrow column tiles while N is small: a workgroup streams K * TN * 4 bytes, the chip has 256 CUs
  // (prefetch depth P: 12 k-steps of 64 slices cover K <= 768, 18 of 16 slices cover K <= 288 -- the Llama shapes)
  if (N <= 4096)
    return launch_gemv<16, 12>(nb, dim3((N + 15) / 16), shm, st, x, x_row_stride, norm_w, eps, W, w_row_stride, blk_cols,
                               w_block_stride, bias, residual, res_row_stride, y, y_row_stride, B, K, N, act, act_ns, act_hd, blk_max, blk_arg);
  if (N <= 16384)
    return launch_gemv<32, 12>(nb, dim3((N + 31) / 32), shm, st, x, x_row_stride, norm_w, eps, W, w_row_stride, blk_cols,
                               w_block_stride, bias, residual, res_row_stride, y, y_row_stride, B, K, N, act, act_ns, act_hd, blk_max, blk_arg);
  return launch_gemv<64, 18>(nb, dim3((N + 63) / 64), shm, st, x, x_row_stride, norm_w, eps, W, w_row_stride, blk_cols,
                             w_block_stride, bias, residual, res_row_stride, y, y_row_stride, B, K, N, act, act_ns, act_hd, blk_max, blk_arg);
}

// ---- RoPE of the new q / k rows + KV-cache append + decode attention (model.py:23-44, 105-121 with L = 1) ----------
// qkv: (B, 3 D) packed [q | k | v] rows of the fused projection.  A workgroup = (batch, head, key range): one CU
// pulls ~11 bytes per clock, and a head's K / V rows at a few hundred positions are ~100 KB, so the T = *pos + 1 keys
// of a head are cut into NS ranges handled by NS workgroups (flash-decoding).  Each rotates its head's q by the angle
// of position *pos (interleaved pairs (x[2i], x[2i+1])); the one whose range holds position *pos also rotates k and
// appends k / v to cache row *pos (kept in LDS: the row this workgroup just stored is not re-read from memory).
// Result per (split, head): [max score m, sum of exp l, -, - | sum of exp(s - m) v]; the merge over the NS ranges
// happens in the staging phase of the output projection (pdn_decode_gemv_f32, act = 2) -- no extra launch.
__global__ __launch_bounds__(256) void decode_attention_kernel(const float* __restrict__ qkv, int64_t qkv_rs,
                                                               const float* __restrict__ cs, const float* __restrict__ sn,
                                                               float* __restrict__ kc, float* __restrict__ vc,
                                                               float* __restrict__ part_out, int H, int hd, int NS,
                                                               int64_t cbs, const int* __restrict__ pos_ptr, float inv_sqrt) {
  extern __shared__ __attribute__((aligned(16))) float sc[];      // [chunk] scores, then [groups + 8][hd] partial sums
  __shared__ __attribute__((aligned(16))) float qs[256], ks[256], vs[256];
  __shared__ float red[16];
  const int pos = *pos_ptr, T = pos + 1;
  const int sp = blockIdx.x % NS, bh = blockIdx.x / NS, b = bh / H, h = bh % H, tid = threadIdx.x;
  const int chunk = (T + NS - 1) / NS, t0 = sp * chunk, t1 = min(T, t0 + chunk);
  const int D = H * hd, f4 = hd / 4, half = hd / 2, rec = 4 + hd;
  float* out = part_out + (((int64_t)b * NS + sp) * H + h) * rec;
  if (t0 >= t1) {                        // no keys in this range (uniform over the workgroup)
    if (tid == 0) { out[0] = -INFINITY; out[1] = 0.f; }
    return;
  }
  float* kb = kc + (int64_t)b * cbs + (int64_t)h * hd;
  float* vb = vc + (int64_t)b * cbs + (int64_t)h * hd;
  const bool owner = pos >= t0 && pos < t1;
  if (tid < half) {
    const float* row = qkv + (int64_t)b * qkv_rs + (int64_t)h * hd + 2 * tid;
    const float c = cs[(int64_t)pos * half + tid], s = sn[(int64_t)pos * half + tid];
    const float2 q = *reinterpret_cast<const float2*>(row);
    *reinterpret_cast<float2*>(qs + 2 * tid) = make_float2(q.x * c - q.y * s, q.x * s + q.y * c);
    if (owner) {
      const float2 k = *reinterpret_cast<const float2*>(row + D);
      const float2 v = *reinterpret_cast<const float2*>(row + 2 * D);
      const float2 kr = make_float2(k.x * c - k.y * s, k.x * s + k.y * c);
      *reinterpret_cast<float2*>(ks + 2 * tid) = kr;
      *reinterpret_cast<float2*>(vs + 2 * tid) = v;
      *reinterpret_cast<float2*>(kb + (int64_t)pos * D + 2 * tid) = kr;
      *reinterpret_cast<float2*>(vb + (int64_t)pos * D + 2 * tid) = v;
    }
  }
  __syncthreads();
  const float4* q4 = reinterpret_cast<const float4*>(qs);
  float m = -INFINITY;
  for (int t = t0 + tid; t < t1; t += 256) {
    const float4* k4 = t == pos ? reinterpret_cast<const float4*>(ks) : reinterpret_cast<const float4*>(kb + (int64_t)t * D);
    float s = 0.f;
    for (int c = 0; c < f4; ++c) {
      const float4 a = q4[c], k = k4[c];
      s += (a.x * k.x + a.y * k.y) + (a.z * k.z + a.w * k.w);
    }
    s *= inv_sqrt;
    sc[t - t0] = s;
    m = fmaxf(m, s);
  }
  m = block_max(m, red);
  float l = 0.f;
  for (int t = t0 + tid; t < t1; t += 256) {
    const float pr = expf(sc[t - t0] - m);
    sc[t - t0] = pr;
    l += pr;
  }
  l = block_sum(l, red);                 // (its barriers also publish the probabilities)
  const int groups = 256 / f4, c = tid % f4, tg = tid / f4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tg < groups) {
    for (int t = t0 + tg; t < t1; t += groups) {
      const float pr = sc[t - t0];
      const float4 v = t == pos ? reinterpret_cast<const float4*>(vs)[c]
                                : *reinterpret_cast<const float4*>(vb + (int64_t)t * D + 4 * c);
      acc.x += pr * v.x; acc.y += pr * v.y; acc.z += pr * v.z; acc.w += pr * v.w;
    }
  }
  __syncthreads();                       // scores are dead: reuse the buffer for the partial sums
  float4* part = reinterpret_cast<float4*>(sc);
  if (tg < groups) part[tg * f4 + c] = acc;
  __syncthreads();
  // combine in a fixed order: 8 threads per column quad add every 8th group, then one thread adds those 8
  if (tid < 8 * f4) {
    const int g0 = tid / f4;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int g = g0; g < groups; g += 8) { const float4 t = part[g * f4 + c]; r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w; }
    part[groups * f4 + tid] = r;
  }
  __syncthreads();
  if (tid < f4) {
    float4 r = part[groups * f4 + tid];
    for (int g = 1; g < 8; ++g) { const float4 t = part[groups * f4 + g * f4 + tid]; r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w; }
    reinterpret_cast<float4*>(out + 4)[tid] = r;
    if (tid == 0) { out[0] = m; out[1] = l; }
  }
}

// qkv rows `qkv_row_stride` floats apart; partials: (B, n_splits, H, 4 + head_dim) floats; caches hold `max_len` positions per sequence, `cache_batch_stride` floats between sequences; cos / sin
// tables (max_len, head_dim / 2).
extern "C" int pdn_decode_attention_f32(const float* qkv, int64_t qkv_row_stride, const float* cos_table,
                                        const float* sin_table, float* k_cache, float* v_cache, float* partials, int B,
                                        int H, int head_dim, int n_splits, int64_t cache_batch_stride, const int* pos,
                                        int max_len, void* stream) {
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
  hipLaunchKernelGGL(decode_attention_kernel, dim3(B * H * NS), dim3(256), shm, (hipStream_t)stream, qkv, qkv_row_stride,
                     cos_table, sin_table, k_cache, v_cache, partials, H, head_dim, NS, cache_batch_stride, pos,
                     1.f / sqrtf((float)head_dim));
  PDN_LAUNCH_CHECK();
  return PDN_OK;
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
  __syncthreads();
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
  const int p = pos ? *pos : 0;
  int64_t* hrow = hist ? *hist + (int64_t)p * B : nullptr;
  for (int b = 0; b < B; ++b) {
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int i = tid; i < n; i += 256) {
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
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < 4; ++w)
        if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
      const int64_t tok = idx == 0x7fffffff ? 0 : idx;
      ids[b] = tok;
      if (hrow) hrow[b] = tok;
      chosen = tok;
    }
    __syncthreads();
    if (emb) {
      const float* row = emb + chosen * emb_rs;
      for (int d = tid; d < D; d += 256) x_next[(int64_t)b * D + d] = row[d];
      __syncthreads();                       // `chosen` is rewritten for the next row
    }
  }
  if (tid == 0 && pos) *pos = p + 1;
}

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
