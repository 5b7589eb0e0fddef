// Sequences of 512 / 768 / 1024 positions on the PERSISTENT attention kernels (csrc/attention_p.hip: head dim 48, one
// 256-row image of K and of V per head in LDS, staged by DMA) -- llm/llama/model.py:112-121 at the lengths
// llm/llama/finetune.py:44 allows (max_seq_len 1024).
//
// The resident kernels of csrc/attention.hip take these lengths in 256-key chunks staged through registers between
// workgroup barriers: 4.3 ns per causal 32 x 32 tile pair forward against 2.4 on the persistent kernels, 11.0 against 8.9
// backward.  A longer sequence is therefore cut into 256-row blocks here and every (query block i, key block j <= i)
// pair runs as ONE launch of the persistent kernels over all heads -- diagonal pairs causal, the others full -- and the
// pairs of a query block are combined the way the key loop of a flash kernel combines its tiles:
//   forward   O_i = sum_j exp(lse_ij - lse_i) O_ij,  lse_i = log sum_j exp(lse_ij)        (att_merge_kernel)
//   backward  every pair recomputes P from the GLOBAL lse_i and delta_i = rowsum(dO_i o O_i) (both exact for a key subset),
//             dQ_i = sum_j dQ_ij, dK_j = sum_i dK_ij, dV_j = sum_i dV_ij      (added in the kernels' own stores: ACC)
// No tile pair is computed twice and none that the mask removes; what is added is one pass over O per extra key block
// forward, and backward a read of the gradient rows a pair adds to.
#include "common.h"
#include <stdint.h>
#include <stdlib.h>
#include <math.h>

int pdn_attention_p_supported(int L, int head_dim);
int pdn_attention_p_fwd(const float* q, const float* k, const float* v, float* o, float* lse, int B, int H, int L,
                        int head_dim, int64_t row_stride, int64_t batch_stride, int64_t o_row_stride,
                        int64_t o_batch_stride, int causal, void* stream);
int pdn_attention_p_bwd_tables(const float* q, const float* k, const float* v, const float* o, const float* d_o, const float* lse,
                               float* dq, float* dk, float* dv, int B, int H, int L, int head_dim, int64_t row_stride,
                               int64_t batch_stride, int64_t o_row_stride, int64_t o_batch_stride, int causal,
                               const float* rope_cos_q, const float* rope_sin_q, const float* rope_cos_k, const float* rope_sin_k,
                               float* delta, void* stream, int acc_q, int acc_kv);
extern "C" int pdn_malloc(void** ptr, int64_t bytes);
extern "C" int pdn_free(void* ptr);

constexpr int AB_ROWS = 256;

// O (rows of block i, strided) = w_a O + w_n O_new;  lse_out = log(exp(lse_a) + exp(lse_n));  w = exp(lse_x - lse_out).
// One thread per 16 bytes of a (batch, position, head) row; lse_* are compact (B * H, 256).
__global__ __launch_bounds__(256) void att_merge_kernel(float* __restrict__ O, int64_t o_rs, int64_t o_bs,
                                                        const float* __restrict__ On, const float* __restrict__ lse_a,
                                                        const float* __restrict__ lse_n, float* __restrict__ lse_out, int B,
                                                        int H, int hd4) {
  const int64_t total = (int64_t)B * AB_ROWS * H * hd4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % hd4);
    int64_t t = i / hd4;
    const int h = (int)(t % H); t /= H;
    const int p = (int)(t % AB_ROWS);
    const int b = (int)(t / AB_ROWS);
    const int64_t li = ((int64_t)b * H + h) * AB_ROWS + p;
    const float a = lse_a[li], n = lse_n[li];
    const float m = fmaxf(a, n);
    const float l = m + logf(expf(a - m) + expf(n - m));
    const float wa = expf(a - l), wn = expf(n - l);
    float4* op = reinterpret_cast<float4*>(O + (int64_t)b * o_bs + (int64_t)p * o_rs + (int64_t)h * hd4 * 4) + c4;
    const float4 x = *op, y = reinterpret_cast<const float4*>(On)[i];
    *op = make_float4(wa * x.x + wn * y.x, wa * x.y + wn * y.y, wa * x.z + wn * y.z, wa * x.w + wn * y.w);
    if (c4 == 0) lse_out[li] = l;
  }
}

// compact (B * H, 256) <-> rows [pos0, pos0 + 256) of the (B * H, L) array
__global__ __launch_bounds__(256) void att_lse_move_kernel(float* __restrict__ full, float* __restrict__ compact, int64_t BH, int L,
                                                           int pos0, int to_full) {
  const int64_t total = BH * AB_ROWS;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t bh = i / AB_ROWS;
    const int p = (int)(i - bh * AB_ROWS);
    if (to_full) full[bh * L + pos0 + p] = compact[i];
    else compact[i] = full[bh * L + pos0 + p];
  }
}

static int ab_grid(int64_t n) { return (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192); }

// 1 when a sequence of L positions is taken block-wise (rotation-free operands, no key bias: the caller checks those)
int pdn_attention_blocks_ok(int L, int head_dim) {
  static const int off = getenv("PDN_ATT_NO_BLOCKS") ? atoi(getenv("PDN_ATT_NO_BLOCKS")) : 0;
  return !off && L > AB_ROWS && L % AB_ROWS == 0 && L <= 1024 && pdn_attention_p_supported(AB_ROWS, head_dim);
}

// Scratch from the library's allocator.  The pool orders reuse on the COMPUTE stream only (csrc/runtime.hip): a block
// freed here is safe for the next compute-stream user, whose kernels queue behind ours.  With the opt-in second stream
// (PDN_TWO_STREAM=1, core/fused/_common.py) a side-stream allocation could take the block while our kernels still read
// it, so in that mode the free waits for the stream first.
struct AbTemp {
  void* p = nullptr;
  hipStream_t st = nullptr;
  int get(int64_t bytes, void* stream) { st = (hipStream_t)stream; return pdn_malloc(&p, bytes); }
  ~AbTemp() {
    static const bool two_stream = getenv("PDN_TWO_STREAM") && atoi(getenv("PDN_TWO_STREAM")) != 0;
    if (p && two_stream) (void)hipStreamSynchronize(st);
    if (p) pdn_free(p);
  }
};

int pdn_attention_blocks_fwd(const float* q, const float* k, const float* v, float* o, float* lse, int B, int H, int L,
                             int hd, int64_t rs, int64_t bs, int64_t o_rs, int64_t o_bs, int causal, void* stream) {
  const int nb = L / AB_ROWS;
  const int64_t BH = (int64_t)B * H, lse_n = BH * AB_ROWS, on = (int64_t)B * AB_ROWS * H * hd;
  AbTemp t_o, t_l;
  int rc = t_o.get(on * 4, stream);
  if (rc) return rc;
  rc = t_l.get(3 * lse_n * 4, stream);
  if (rc) return rc;
  float* On = (float*)t_o.p;
  float* la = (float*)t_l.p;
  float* lb = la + lse_n;
  float* ln = lb + lse_n;
  hipStream_t st = (hipStream_t)stream;
  for (int i = 0; i < nb; ++i) {
    const float* qi = q + (int64_t)i * AB_ROWS * rs;
    float* oi = o + (int64_t)i * AB_ROWS * o_rs;
    const int jn = causal ? i + 1 : nb;
    for (int j = 0; j < jn; ++j) {
      const float* kj = k + (int64_t)j * AB_ROWS * rs;
      const float* vj = v + (int64_t)j * AB_ROWS * rs;
      const int diag = causal && j == i;
      if (j == 0) {
        rc = pdn_attention_p_fwd(qi, kj, vj, oi, la, B, H, AB_ROWS, hd, rs, bs, o_rs, o_bs, diag, stream);
        if (rc) return rc;
      } else {
        rc = pdn_attention_p_fwd(qi, kj, vj, On, ln, B, H, AB_ROWS, hd, rs, bs, (int64_t)H * hd, (int64_t)AB_ROWS * H * hd, diag, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(att_merge_kernel, dim3(ab_grid(on / 4)), dim3(256), 0, st, oi, o_rs, o_bs, On, la, ln, lb, B, H, hd / 4);
        PDN_LAUNCH_CHECK();
        float* sw = la; la = lb; lb = sw;
      }
    }
    hipLaunchKernelGGL(att_lse_move_kernel, dim3(ab_grid(lse_n)), dim3(256), 0, st, lse, la, BH, L, i * AB_ROWS, 1);
    PDN_LAUNCH_CHECK();
  }
  return PDN_OK;
}

// workspace: 2 * B * H * 256 floats (delta of a query block, its lse in the compact layout)
int pdn_attention_blocks_bwd(const float* q, const float* k, const float* v, const float* o, const float* d_o, const float* lse,
                             float* dq, float* dk, float* dv, int B, int H, int L, int hd, int64_t rs, int64_t bs, int64_t o_rs,
                             int64_t o_bs, int causal, const float* rope_cos, const float* rope_sin, float* workspace,
                             void* stream) {
  const int nb = L / AB_ROWS;
  const int64_t BH = (int64_t)B * H, lse_n = BH * AB_ROWS;
  float* delta = workspace;
  float* lc = workspace + lse_n;
  // The first pair that contributes to a gradient block writes it, every further one ADDS in its store (the ACC
  // instantiations of the backward kernels): no scratch, no extra pass.  (First form: scratch arrays + one add pass per
  // further contribution -- 3 passes at L = 512, 18 at 1024.)
  hipStream_t st = (hipStream_t)stream;
  const int64_t tab = (int64_t)AB_ROWS * (hd / 2);             // floats of the (L, hd / 2) cos / sin tables per block
  bool k_seen[8] = {false, false, false, false, false, false, false, false};
  for (int i = 0; i < nb; ++i) {
    hipLaunchKernelGGL(att_lse_move_kernel, dim3(ab_grid(lse_n)), dim3(256), 0, st, const_cast<float*>(lse), lc, BH, L, i * AB_ROWS, 0);
    PDN_LAUNCH_CHECK();
    const int64_t qo = (int64_t)i * AB_ROWS * rs, oo = (int64_t)i * AB_ROWS * o_rs;
    const int jn = causal ? i + 1 : nb;
    for (int j = 0; j < jn; ++j) {
      const int64_t ko = (int64_t)j * AB_ROWS * rs;
      const int rc = pdn_attention_p_bwd_tables(q + qo, k + ko, v + ko, o + oo, d_o + oo, lc, dq + qo, dk + ko, dv + ko, B, H,
                                                AB_ROWS, hd, rs, bs, o_rs, o_bs, causal && j == i,
                                                rope_cos ? rope_cos + i * tab : nullptr, rope_sin ? rope_sin + i * tab : nullptr,
                                                rope_cos ? rope_cos + j * tab : nullptr, rope_sin ? rope_sin + j * tab : nullptr,
                                                delta, stream, j > 0, k_seen[j]);
      if (rc) return rc;
      k_seen[j] = true;
    }
  }
  return PDN_OK;
}
