// float64 matmul for gfx950 (v_mfma_f64_16x16x4_f64): the reference's default dtype is float64
// (`Tensor(data, dtype=None)` keeps NumPy's, pydynet/core/tensor.py:65-94; nn layers built without
// `dtype=` are float64 too), and its own tests draw float64 operands for `@`
// (tests/test_tensor_basic.py:16,107-117).  A script that never says float32 must still run on the HIP device
// with the right numbers: this is the correctness path for it -- strided / broadcast-batched operands like
// pdn_gemm_f32, operands fed to the MFMA straight from global memory through the caches (no LDS staging,
// no split-K).  The benchmarked hot path is float32 and does not come through here.
#include "common.h"

typedef double f64x4 __attribute__((ext_vector_type(4)));

struct Gemm64Params {
  int M, N, K, nb2;
  double alpha, beta;
  const double* A; const double* B; double* C;
  int64_t a_rs, a_cs, b_rs, b_cs, ldc;
  int64_t a_b1, a_b2, b_b1, b_b2, c_b1, c_b2;
};

// One wave64 per 32 x 32 tile of C (2 x 2 MFMA tiles of 16 x 16), four waves per workgroup.
// MFMA operand layout: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15];
// C/D: column = lane & 15, row = (lane >> 4) + 4 * reg.
__global__ __launch_bounds__(256) void gemm_f64_mfma_kernel(Gemm64Params p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, lk = lane >> 4;
  const int tiles_n = (p.N + 31) / 32;
  const int tile = blockIdx.x * 4 + wave;
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  if (tm * 32 >= p.M) return;
  const int batch = blockIdx.y, b1 = batch / p.nb2, b2 = batch - b1 * p.nb2;
  const double* A = p.A + b1 * p.a_b1 + b2 * p.a_b2;
  const double* B = p.B + b1 * p.b_b1 + b2 * p.b_b2;
  double* C = p.C + b1 * p.c_b1 + b2 * p.c_b2;
  f64x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;
  int rows[2], cols[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    rows[i] = tm * 32 + i * 16 + l15;
    cols[i] = tn * 32 + i * 16 + l15;
  }
  for (int k0 = 0; k0 < p.K; k0 += 4) {
    const int k = k0 + lk;
    double a[2], b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      a[i] = (rows[i] < p.M && k < p.K) ? A[(int64_t)rows[i] * p.a_rs + (int64_t)k * p.a_cs] : 0.0;
      b[i] = (cols[i] < p.N && k < p.K) ? B[(int64_t)k * p.b_rs + (int64_t)cols[i] * p.b_cs] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = tm * 32 + i * 16 + lk + 4 * r, n = tn * 32 + j * 16 + l15;
        if (m < p.M && n < p.N) {
          double* dst = C + (int64_t)m * p.ldc + n;
          double v = p.alpha * acc[i][j][r];
          if (p.beta != 0.0) v += p.beta * *dst;
          *dst = v;
        }
      }
}

/* C[b1,b2] = alpha * A[b1,b2] (M x K) * B[b1,b2] (K x N) + beta * C[b1,b2], float64; element strides and the two
 * broadcastable batch dims as in pdn_gemm_f32 (`x.data @ y.data` and both products of matmul.grad_fn,
 * core/tensor.py:657-676, for float64 tensors). */
extern "C" int pdn_gemm_f64(int M, int N, int K, double alpha, const double* A, int64_t a_rs, int64_t a_cs,
                            const double* B, int64_t b_rs, int64_t b_cs, double beta, double* C, int64_t ldc,
                            int nb1, int nb2, int64_t a_bs1, int64_t a_bs2, int64_t b_bs1, int64_t b_bs2,
                            int64_t c_bs1, int64_t c_bs2, void* stream) {
  PDN_CHECK_ARG(M >= 0 && N >= 0 && K >= 0 && nb1 >= 0 && nb2 >= 0, "pdn_gemm_f64: negative extent");
  if (M == 0 || N == 0 || nb1 == 0 || nb2 == 0) return PDN_OK;
  PDN_CHECK_ARG(A && B && C, "pdn_gemm_f64: null operand");
  PDN_CHECK_ARG((int64_t)nb1 * nb2 <= 65535, "pdn_gemm_f64: too many batches");
  Gemm64Params p{M, N, K, nb2, alpha, beta, A, B, C, a_rs, a_cs, b_rs, b_cs, ldc,
                 a_bs1, a_bs2, b_bs1, b_bs2, c_bs1, c_bs2};
  const int64_t tiles = (int64_t)((M + 31) / 32) * ((N + 31) / 32);
  hipLaunchKernelGGL(gemm_f64_mfma_kernel, dim3((unsigned)((tiles + 3) / 4), nb1 * nb2), dim3(256), 0,
                     (hipStream_t)stream, p);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}
