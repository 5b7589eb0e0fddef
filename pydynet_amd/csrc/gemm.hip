// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32), gfx950 only.
//
// Replaces the reference's `x.data @ y.data` and the two products in matmul.grad_fn
// (pydynet/core/tensor.py:657-676) which dispatch to OpenBLAS / cuBLAS through `xp`.
//
//   C[b1,b2] = alpha * A[b1,b2] (MxK) * B[b1,b2] (KxN)  (+ bias[N])  (+ beta * C[b1,b2])
//
// Operands are addressed through element strides so that transposed / head-split views
// (q.transpose(0,2,1,3), W.T, x.T ...) are consumed in place, never copied:
//   A(m,k) = A[m*a_rs + k*a_cs]      B(k,n) = B[k*b_rs + n*b_cs]      C(m,n) = C[m*ldc + n]
//
// Kernel structure (one workgroup = 2 or 4 wave64, one wave per SIMD, 2+ workgroups per CU):
//   * block tile BM x BN x BK, wave tile (WM*32) x (WN*32) of 32x32 MFMA accumulators; BK is
//     32 for tiles up to 128x128 and 16 for the 256-wide ones so that every shape keeps
//     <= 74 KB of LDS (two workgroups per CU) and one barrier per >= 2048 MFMA cycles;
//   * global -> registers -> LDS staging, double-buffered in LDS: the loads for tile t+1 are
//     issued before the MFMAs of tile t, and their LDS writes are slotted into the MIDDLE of
//     that MFMA stream (the matrix pipe keeps executing queued MFMAs while the wave issues
//     ds_write), so staging costs no matrix-pipe time; one barrier per k-tile;
//   * the contraction index inside an MFMA is permuted (half-wave h, step j -> k = 8t+4h+j)
//     so a K-contiguous operand is fetched from LDS with ONE ds_read_b128 per four MFMAs
//     and an M/N-contiguous operand with conflict-free ds_read_b32 -- no transposes anywhere;
//   * interior tiles run a bounds-test-free copy of the main loop;
//   * epilogue: accumulators are parked in LDS one 32-row band at a time and written as
//     16-byte pieces (16 lanes cover 256 B of a row) with alpha / bias / beta*C applied;
//   * XCD-aware tile order: each XCD walks a contiguous run of 8x8 super-tiles so the A and
//     B panels it touches stay in its private 4 MiB L2;
//   * split-K (workspace + deterministic reduce) when the output has too few tiles to
//     fill 256 CUs.
// Besides the tiled kernel this file holds the wave-streaming kernels for weight gradients
// (small output, K = tokens: gemm_tn_stream_*), the skinny kernel for one to four output rows
// (decode: gemm_skinny_kernel) and the host-side selection between them (pdn_gemm_f32).
// f32 MFMA is an exact k-ordered fmaf chain, so results match a scalar fp32 dot product
// in a different summation order only (tolerance documented in tests/).
#include "common.h"
#include <vector>
#include <mutex>
#include <stdlib.h>
#include <stdio.h>
#include <type_traits>

#include "gemm_tiled.h"    // GemmParams, tile loaders, gemm_f32_mfma_kernel, split-K reduce
#include "gemm_stream.h"   // gemm_tn_stream_{,lds_,dma_}kernel, gemm_skinny_kernel

// ---- host side ----------------------------------------------------------------------
struct TileCfg {
  int waves_m, waves_n, wm, wn, bk;
  float eff;  // relative efficiency of the tile shape (operand bytes per MFMA, barrier rate)
};
static const TileCfg kCfgs[] = {
    // eff values are fitted to the round-1 tile sweep on MI355X (tools/gemm_bench.py, profiles/)
    {2, 2, 2, 2, 32, 1.00f},  // 0: 128 x 128
    {4, 1, 1, 3, 32, 0.97f},  // 1: 128 x  96   (N = 288 = 3*96)
    {1, 4, 3, 1, 32, 0.95f},  // 2:  96 x 128   (M = 288, weight gradients)
    {4, 1, 1, 2, 32, 0.85f},  // 3: 128 x  64
    {1, 4, 2, 1, 32, 0.90f},  // 4:  64 x 128
    {2, 2, 1, 1, 32, 0.60f},  // 5:  64 x  64
    {2, 2, 4, 2, 16, 0.85f},  // 6: 256 x 128
    {2, 2, 2, 4, 16, 1.10f},  // 7: 128 x 256   (only offered to very wide outputs, see below)
    {4, 1, 2, 3, 16, 0.99f},  // 8: 256 x  96
    {1, 4, 3, 2, 16, 0.70f},  // 9:  96 x 256
    {2, 1, 1, 3, 32, 0.78f},  // 10: 64 x  96   (2 waves: finer quantisation for N = 288)
    {1, 2, 3, 1, 32, 0.70f},  // 11: 96 x  64
    {4, 1, 1, 3, 16, 1.15f},  // 12: 128 x 96, BK 16: 51 KB of LDS -> three workgroups per CU (offered selectively)
    // (tried in round 2: {4, 1, 1, 9, 16} = 128 x 288, the whole 288-wide row panel in one workgroup so that the
    //  65536 x 32000 dlogits operand of the lm_head input gradient is read from HBM once instead of 1.76 times
    //  (PMC: FETCH 14.8 GB vs 8.4 GB algorithmic) -- 10.45 ms against 9.79 ms for the 96 x 128 tile: the product is
    //  matrix-pipe bound, the re-reads are L2 / MALL hits that cost nothing; not kept)
};
static const int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);
static const int kScalarCfg = 5;

template <int WMV, int WNV, int WM, int WN, int BK, bool VEC, bool MASKS = false>
static void launch_layout(const GemmParams& p, bool a_kin, bool b_kin, dim3 grid, hipStream_t st) {
  constexpr int NT = WMV * WNV * 64;
  if (a_kin && b_kin)
    hipLaunchKernelGGL((gemm_f32_mfma_kernel<WMV, WNV, WM, WN, BK, true, true, VEC, false, MASKS>), grid, dim3(NT), 0, st, p);
  else if (a_kin && !b_kin)
    hipLaunchKernelGGL((gemm_f32_mfma_kernel<WMV, WNV, WM, WN, BK, true, false, VEC, false, MASKS>), grid, dim3(NT), 0, st, p);
  else if (!a_kin && b_kin)
    hipLaunchKernelGGL((gemm_f32_mfma_kernel<WMV, WNV, WM, WN, BK, false, true, VEC, false, MASKS>), grid, dim3(NT), 0, st, p);
  else if (!MASKS && p.colsum && VEC)   // x^T @ g with the bias gradient (column sums of g) fused in
    hipLaunchKernelGGL((gemm_f32_mfma_kernel<WMV, WNV, WM, WN, BK, false, false, VEC, VEC && !MASKS, false>), grid, dim3(NT), 0, st, p);
  else
    hipLaunchKernelGGL((gemm_f32_mfma_kernel<WMV, WNV, WM, WN, BK, false, false, VEC, false, MASKS>), grid, dim3(NT), 0, st, p);
}

// ---- optional per-launch timing of the dominant kernel (bench.py roofline) ------------
static std::mutex g_prof_mu;
static bool g_prof_on = false;
struct ProfRec { hipEvent_t e0, e1; double flops; int family; double bytes = 0.0; };   // family 0: tiled MFMA kernel, 1: wave-streaming TN kernel, 2: row-resident kernel, 3: output-resident kernel, 4: its weight-gradient (TN) form
static std::vector<ProfRec> g_prof;

// for the fused-epilogue entry points of the other GEMM files: time a launch under a family when profiling is on
// (token = index + 1 of the open record, 0 when profiling is off)
int pdn_gemm_prof_begin(int family, double flops, double bytes, void* stream) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (!g_prof_on) return 0;
  ProfRec rec;
  if (hipEventCreate(&rec.e0) != hipSuccess || hipEventCreate(&rec.e1) != hipSuccess) return 0;
  rec.flops = flops; rec.family = family; rec.bytes = bytes;
  (void)hipEventRecord(rec.e0, (hipStream_t)stream);
  g_prof.push_back(rec);
  return (int)g_prof.size();
}
void pdn_gemm_prof_end(int token, void* stream) {
  if (!token) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (token <= (int)g_prof.size()) (void)hipEventRecord(g_prof[token - 1].e1, (hipStream_t)stream);
}

extern "C" int pdn_gemm_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = on != 0;
  return PDN_OK;
}

// The row-resident launches with a fused epilogue (family 5: SwiGLU forward / backward, RoPE -- csrc/gemm_rowres.hip) do
// a bandwidth pass inside a GEMM: they are reported apart, with their algorithmic HBM bytes beside their FLOPs.  Collects
// AND removes their records; records left in place are counted with the row-resident family by the call below.
extern "C" int pdn_gemm_prof_collect_fused(double* ms, double* flops, double* bytes, int64_t* launches) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  double tm = 0, tf = 0, tb = 0;
  int64_t n = 0;
  std::vector<ProfRec> keep;
  for (auto& r : g_prof) {
    if (r.family != 5) { keep.push_back(r); continue; }
    PDN_HIP(hipEventSynchronize(r.e1));
    float t = 0.f;
    PDN_HIP(hipEventElapsedTime(&t, r.e0, r.e1));
    tm += t; tf += r.flops; tb += r.bytes; n++;
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  g_prof.swap(keep);
  if (ms) *ms = tm;
  if (flops) *flops = tf;
  if (bytes) *bytes = tb;
  if (launches) *launches = n;
  return PDN_OK;
}

// per kernel family: [0] gemm_f32_mfma_kernel (+ its split-K reduce), [1] gemm_tn_stream_*_kernel,
// [2] gemm_rowres_kernel (csrc/gemm_rowres.hip), [3] gemm_outres_kernel, [4] gemm_outres_tn_kernel (csrc/gemm_outres.hip);
// the three output arrays have FIVE entries each
#define PDN_GEMM_FAMILIES 5
extern "C" int pdn_gemm_prof_collect_families(double* ms5, double* flops5, int64_t* launches5) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  double ms[PDN_GEMM_FAMILIES] = {0}, fl[PDN_GEMM_FAMILIES] = {0};
  int64_t n[PDN_GEMM_FAMILIES] = {0};
  for (auto& r : g_prof) {
    PDN_HIP(hipEventSynchronize(r.e1));
    float t = 0.f;
    PDN_HIP(hipEventElapsedTime(&t, r.e0, r.e1));
    const int f = r.family == 5 ? 2 : (r.family < 0 || r.family >= PDN_GEMM_FAMILIES ? 0 : r.family);
    ms[f] += t; fl[f] += r.flops; n[f]++;
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  for (int f = 0; f < PDN_GEMM_FAMILIES; ++f) {
    if (ms5) ms5[f] = ms[f];
    if (flops5) flops5[f] = fl[f];
    if (launches5) launches5[f] = n[f];
  }
  g_prof.clear();
  return PDN_OK;
}

extern "C" int pdn_gemm_prof_collect(double* total_ms, double* total_flops, int64_t* launches) {
  double ms[PDN_GEMM_FAMILIES], fl[PDN_GEMM_FAMILIES];
  int64_t n[PDN_GEMM_FAMILIES];
  int rc = pdn_gemm_prof_collect_families(ms, fl, n);
  if (rc) return rc;
  double tm = 0, tf = 0;
  int64_t tn = 0;
  for (int f = 0; f < PDN_GEMM_FAMILIES; ++f) { tm += ms[f]; tf += fl[f]; tn += n[f]; }
  if (total_ms) *total_ms = tm;
  if (total_flops) *total_flops = tf;
  if (launches) *launches = tn;
  return PDN_OK;
}

int pdn_gemm_outres_tn_plan(int N, int K, int* nw_out, int* k_per_split_out);
int pdn_gemm_outres_tn_launch(const float* X, const float* G, float* C, int N, int K, int64_t ldx, int64_t ldg,
                              int64_t ldc, int64_t slab, int nw, int k_per_split, void* stream);
int pdn_gemm_outres_tn_blocks_launch(const float* X, const float* G, float* C, int N, int K, int64_t ldx, int64_t ldg,
                                     int nb_cols, int nw, int k_per_split, void* stream);
extern "C" int64_t pdn_gemm_outres_workspace_bytes(int M, int K);
extern "C" int pdn_gemm_outres_plan(int M, int K, int* nw, int* kps);
extern "C" int pdn_gemm_outres_ws_f32(const float* A, const float* B, float* C, const float* bias, const float* residual,
                                      int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int b_trans,
                                      void* workspace, int64_t workspace_bytes, void* stream);
extern "C" int pdn_gemm_outres_supported(int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int b_trans);
extern "C" int pdn_gemm_outres_f32(const float* A, const float* B, float* C, const float* bias,
                                   const float* residual, int M, int N, int K, int64_t lda, int64_t ldb,
                                   int64_t ldc, int b_trans, void* stream);
extern "C" int pdn_gemm_rowres_supported(int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int b_trans);
int pdn_rowtile_plain_ok(int M, int N, int b_trans);
extern "C" int pdn_gemm_rowres_f32(const float* A, const float* B, float* C, const float* bias,
                                   const float* residual, int M, int N, int K, int64_t lda, int64_t ldb,
                                   int64_t ldc, int b_trans, void* stream);
int pdn_gemm_rowres_blocks(const float* A, const float* B, float* C, int M, int N, int K, int64_t lda, int64_t ldb,
                           int64_t ldc, int b_trans, int nblocks, int64_t b_block_stride, void* stream);

extern "C" int64_t pdn_gemm_f32_workspace_bytes(int M, int N, int K, int nbatch) {
  // Enough for up to 64 splits of one output; pdn_gemm_f32 never uses more than it is given.
  (void)K;
  return (int64_t)64 * M * N * (int64_t)nbatch * 4;
}

// csrc/gemm_narrow.hip: products with at most 16 output columns (bandwidth kernels)
bool pdn_gemm_narrow_nn_ok(int M, int N, int K, int64_t a_rs, int64_t a_cs, const void* A);
int pdn_gemm_narrow_nn_launch(const float* A, int64_t lda, const float* B, int64_t b_rs, int64_t b_cs, const float* bias,
                              float* C, int64_t ldc, int M, int N, int K, void* stream);
int pdn_gemm_narrow_tn_plan(int Mc, int N, int K, int64_t a_rs, int64_t a_cs, int64_t b_cs, const void* A, int64_t ws_cap_floats,
                            int* kps);
int pdn_gemm_narrow_tn_launch(const float* A, int64_t a_cs, const float* B, int64_t b_rs, float* slabs, int Mc, int N, int K,
                              int kps, int splits, void* stream);

// pdn_gemm_f32 proper.  `relu_mask` / `grad_mask` (GemmParams) are the one-bit-per-element epilogues of
// pdn_linear_relu_fwd_f32 / pdn_linear_dx_masked_f32: with either, the product goes to the tiled kernel unsplit.
static int gemm_f32_impl(int M, int N, int K, float alpha, const float* A, int64_t a_rs,
                         int64_t a_cs, const float* B, int64_t b_rs, int64_t b_cs, float beta,
                         float* C, int64_t ldc, const float* bias, int nb1, int nb2,
                         int64_t a_bs1, int64_t a_bs2, int64_t b_bs1, int64_t b_bs2,
                         int64_t c_bs1, int64_t c_bs2, const float* residual,
                         float* b_colsum, int colsum_accumulate, void* workspace,
                         int64_t workspace_bytes, void* stream, uint32_t* relu_mask, const uint32_t* grad_mask,
                         float* mask_colsum = nullptr) {
  const bool ext_on = relu_mask || grad_mask;
  PDN_CHECK_ARG(!ext_on || (nb1 == 1 && nb2 == 1 && N % 32 == 0 && !b_colsum),
                "pdn_gemm_f32: the mask epilogues need one batch and N a multiple of 32 (N = %d)", N);
  PDN_CHECK_ARG(M >= 0 && N >= 0 && K >= 0 && nb1 >= 0 && nb2 >= 0, "pdn_gemm_f32: negative extent");
  if (M == 0 || N == 0 || nb1 == 0 || nb2 == 0) return PDN_OK;
  PDN_CHECK_ARG(A && B && C, "pdn_gemm_f32: null operand");
  PDN_CHECK_ARG(ldc >= N, "pdn_gemm_f32: ldc (%lld) < N (%d)", (long long)ldc, N);
  hipStream_t st = (hipStream_t)stream;
  const int nbatch = nb1 * nb2;

  GemmParams p;
  p.A = A; p.B = B; p.C = C; p.bias = bias; p.ws = (float*)workspace;
  p.residual = residual; p.colsum = b_colsum; p.colsum_acc = colsum_accumulate;
  p.relu_mask = relu_mask; p.grad_mask = grad_mask; p.mask_colsum = grad_mask ? mask_colsum : nullptr;
  p.M = M; p.N = N; p.K = K;
  p.a_rs = a_rs; p.a_cs = a_cs; p.b_rs = b_rs; p.b_cs = b_cs; p.ldc = ldc;
  p.nb2 = nb2;
  p.a_bs1 = a_bs1; p.a_bs2 = a_bs2; p.b_bs1 = b_bs1; p.b_bs2 = b_bs2; p.c_bs1 = c_bs1; p.c_bs2 = c_bs2;
  p.alpha = alpha; p.beta = beta;

  // LDS layout per operand: contraction-contiguous (KIN) unless the m/n index is the unit stride.
  const bool a_kin = !(a_rs == 1 && a_cs != 1) || M == 1;
  const bool b_kin = (b_rs == 1 && b_cs != 1) || (b_rs == 1 && N == 1);
  auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
  auto m4 = [](int64_t v) { return (v & 3) == 0; };
  // float4 path: unit stride on the staged-contiguous index, every row start 16B aligned,
  // extents along the contiguous index multiples of 4.
  bool vec = al16(A) && al16(B) && m4(a_bs1) && m4(a_bs2) && m4(b_bs1) && m4(b_bs2);
  if (a_kin) vec = vec && a_cs == 1 && m4(a_rs) && m4(K);
  else vec = vec && a_rs == 1 && m4(a_cs) && m4(M);
  if (b_kin) vec = vec && b_rs == 1 && m4(b_cs) && m4(K);
  else vec = vec && b_cs == 1 && m4(b_rs) && m4(N);

  if (b_colsum) {
    // fused column sums need B staged n-contiguous, A m-contiguous (the dW = x^T @ g form), the
    // float4 path, a single batch and no k-split (each column is summed by exactly one block)
    if (a_kin || b_kin || !vec || nbatch != 1) {
      pdn_set_error("pdn_gemm_f32: b_colsum needs the x^T @ g layout (A m-contiguous, B n-contiguous, aligned, unbatched)");
      return PDN_EUNSUPPORTED;
    }
  }
  // ---- one to four output rows (decode): stream the weight matrix once ----------------------
  // (a workgroup walks ALL of K for its 128 columns: right when the weight matrix is wide, wrong for a
  //  long-K / narrow-N product such as the input-weight gradient of an RNN with one input feature)
  if (M <= 4 && !ext_on && nbatch == 1 && !b_colsum && b_cs == 1 && a_cs == 1 && m4(N) && m4(b_rs) && m4(ldc) &&
      al16(B) && al16(C) && (!bias || al16(bias)) && (!residual || al16(residual)) && K > 0 &&
      (K <= 4096 || (int64_t)K <= 4ll * N)) {
    const dim3 g((N + 127) / 128);
    switch (M) {
      case 1: hipLaunchKernelGGL((gemm_skinny_kernel<1>), g, dim3(256), 0, st, p); break;
      case 2: hipLaunchKernelGGL((gemm_skinny_kernel<2>), g, dim3(256), 0, st, p); break;
      case 3: hipLaunchKernelGGL((gemm_skinny_kernel<3>), g, dim3(256), 0, st, p); break;
      default: hipLaunchKernelGGL((gemm_skinny_kernel<4>), g, dim3(256), 0, st, p); break;
    }
    PDN_LAUNCH_CHECK();
    return PDN_OK;
  }
  // ---- a handful of output columns (a classifier head): bandwidth kernels (gemm_narrow.hip) ----
  if (nbatch == 1 && !ext_on && N <= 16 && alpha == 1.f && !residual && !b_colsum && !getenv("PDN_GEMM_NO_NARROW")) {
    if (beta == 0.f && pdn_gemm_narrow_nn_ok(M, N, K, a_rs, a_cs, A)) {
      if (getenv("PDN_GEMM_DEBUG")) fprintf(stderr, "pdn_gemm_f32 M=%d N=%d K=%d -> narrow NN\n", M, N, K);
      return pdn_gemm_narrow_nn_launch(A, a_rs, B, b_rs, b_cs, bias, C, ldc, M, N, K, stream);
    }
    int kps = 0;
    const int64_t cap = (workspace && workspace_bytes > 0 && al16(workspace)) ? workspace_bytes / 4 : 0;
    const int sp = !bias ? pdn_gemm_narrow_tn_plan(M, N, K, a_rs, a_cs, b_cs, A, cap, &kps) : 0;
    if (sp > 0) {
      if (getenv("PDN_GEMM_DEBUG")) fprintf(stderr, "pdn_gemm_f32 M=%d N=%d K=%d -> narrow TN (%d slabs)\n", M, N, K, sp);
      const int rc = pdn_gemm_narrow_tn_launch(A, a_cs, B, b_rs, (float*)workspace, M, N, K, kps, sp, stream);
      if (rc) return rc;
      p.splits = sp;
      const int64_t total = (int64_t)M * N;
      const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
      const int rvec = (N % 4 == 0) && (ldc % 4 == 0) && al16(C);
      hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, p, 1, rvec);
      PDN_LAUNCH_CHECK();
      return PDN_OK;
    }
  }
  // ---- tall A, contraction 288, wide-enough output: rows of A resident in registers (gemm_rowres.hip) ----
  // measured against the tiled kernel at 65536 rows: N 864 +19 %, 1536 +14 %, 768 (either B orientation)
  // +11..14 %, 32000 +4 %; N 288 equal (left to the tiled kernel); with a residual the tiled epilogue wins
  // ---- tall A, output exactly the model width (288), longer contraction: outputs resident in accumulators
  // (gemm_outres.hip).  Measured against the tiled kernel at 65536 rows: K 768 +6 %, 864 +17 %, 1536 +20 %,
  // 32000 +15 % (91.7 % of the matrix peak); at 32768 rows (4-wave workgroups, one wave per SIMD) K >= 1536 only.
  if (nbatch == 1 && !ext_on && N == 288 && a_cs == 1 && alpha == 1.f && beta == 0.f && !b_colsum && K >= 768 &&
      al16(A) && al16(B) && !getenv("PDN_GEMM_NO_OUTRES")) {
    const int bt = (b_rs == 1 && b_cs != 1) ? 1 : 0;
    const int64_t ldb = bt ? b_cs : b_rs;
    const bool big = (M + 255) / 256 >= 224, mid = (M + 127) / 128 >= 224 && K >= 1536;
    // fewer rows than that: K cut into ranges over grid.y (slabs in the workspace), when K is long enough
    int pl_nw, pl_kps;
    const int64_t or_ws = (workspace && workspace_bytes > 0) ? workspace_bytes : 0;
    // (the plan may also cut K where unsplit 4-wave workgroups would fill the chip: 8-wave ones over two ranges)
    const bool cut = !big && M >= 8192 && K >= 1536 && pdn_gemm_outres_plan(M, K, &pl_nw, &pl_kps) > 1 &&
                     or_ws >= pdn_gemm_outres_workspace_bytes(M, K);
    if ((big || mid || cut) && (bt || b_cs == 1) && pdn_gemm_outres_supported(M, N, K, a_rs, ldb, ldc, bt)) {
      bool prof;
      ProfRec rec;
      {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        prof = g_prof_on;
      }
      if (prof) {
        PDN_HIP(hipEventCreate(&rec.e0));
        PDN_HIP(hipEventCreate(&rec.e1));
        rec.flops = 2.0 * M * (double)N * (double)K;
        rec.family = 3;
        PDN_HIP(hipEventRecord(rec.e0, st));
      }
      if (getenv("PDN_GEMM_DEBUG"))
        fprintf(stderr, "pdn_gemm_f32 M=%d N=%d K=%d -> output-resident (%s)\n", M, N, K, bt ? "NT" : "NN");
      const int rc = cut ? pdn_gemm_outres_ws_f32(A, B, C, bias, residual, M, N, K, a_rs, ldb, ldc, bt, workspace, or_ws, stream)
                         : pdn_gemm_outres_f32(A, B, C, bias, residual, M, N, K, a_rs, ldb, ldc, bt, stream);
      if (rc) return rc;
      if (prof) {
        PDN_HIP(hipEventRecord(rec.e1, st));
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof.push_back(rec);
      }
      return PDN_OK;
    }
  }
  // A batch whose members share A and write side by side into one packed buffer (x against Wq | Wk | Wv,
  // x against Wg | Wu: core/fused/) is ONE such product with B in equally spaced column blocks.
  {
    const bool one_dim = nb1 == 1 || nb2 == 1;
    const int64_t a_bs = nb1 == 1 ? a_bs2 : a_bs1, b_bs = nb1 == 1 ? b_bs2 : b_bs1, c_bs = nb1 == 1 ? c_bs2 : c_bs1;
    const bool blocks = nbatch > 1 && one_dim && a_bs == 0 && c_bs == N && ldc >= (int64_t)nbatch * N && !bias &&
                        N % 96 == 0 && (b_bs & 3) == 0;
    const int64_t n_all = (int64_t)N * (blocks ? nbatch : 1);
    // (round 5: the tile-piece kernel also takes what the chunk kernel lost to the tiled one -- outputs narrower than 768
    //  columns, i.e. the 288 x 288 products of the attention output projection, and a residual added in the store)
    const int bt0 = (b_rs == 1 && b_cs != 1) ? 1 : 0;
    const bool tilepiece = nbatch == 1 && K == 288 && n_all >= 96 && n_all % 32 == 0 && M >= 8192 &&
                           pdn_rowtile_plain_ok(M, (int)n_all, bt0);
    if ((nbatch == 1 || blocks) && !ext_on && K == 288 && a_cs == 1 && alpha == 1.f && beta == 0.f && !b_colsum &&
        (tilepiece || (!residual && n_all >= 768)) && n_all < (1 << 30) && M >= 8192 && al16(A) && al16(B) &&
        (!residual || al16(residual)) && !getenv("PDN_GEMM_NO_ROWRES")) {
      const int bt = (b_rs == 1 && b_cs != 1) ? 1 : 0;
      const int64_t ldb = bt ? b_cs : b_rs;
      if ((bt || b_cs == 1) && pdn_gemm_rowres_supported(M, N, K, a_rs, ldb, ldc, bt)) {
        bool prof;
        ProfRec rec;
        {
          std::lock_guard<std::mutex> lk(g_prof_mu);
          prof = g_prof_on;
        }
        if (prof) {
          PDN_HIP(hipEventCreate(&rec.e0));
          PDN_HIP(hipEventCreate(&rec.e1));
          rec.flops = 2.0 * M * (double)n_all * (double)K;
          rec.family = 2;
          PDN_HIP(hipEventRecord(rec.e0, st));
        }
        if (getenv("PDN_GEMM_DEBUG"))
          fprintf(stderr, "pdn_gemm_f32 M=%d N=%lld K=%d -> row-resident (%s, %d block%s)\n", M, (long long)n_all, K,
                  bt ? "NT" : "NN", blocks ? nbatch : 1, blocks ? "s" : "");
        const int rc = blocks ? pdn_gemm_rowres_blocks(A, B, C, M, (int)n_all, K, a_rs, ldb, ldc, bt, nbatch, b_bs, stream)
                              : pdn_gemm_rowres_f32(A, B, C, bias, residual, M, N, K, a_rs, ldb, ldc, bt, stream);
        if (rc) return rc;
        if (prof) {
          PDN_HIP(hipEventRecord(rec.e1, st));
          std::lock_guard<std::mutex> lk(g_prof_mu);
          g_prof.push_back(rec);
        }
        return PDN_OK;
      }
    }
  }
  const int64_t ws_cap = (workspace && workspace_bytes > 0) ? workspace_bytes / 4 : 0;
  // ---- weight gradient of a wide layer out of the model width: C (288 x N) = x^T (288 x K) g (K x N), K = tokens --
  // (the lm_head: N = 32000).  Output-resident over the 288 rows, g read once straight into MFMA operands
  // (gemm_outres_tn_kernel); K split so that the grid fills the chip once, slabs combined (with beta) below.
  if (nbatch == 1 && !ext_on && M == 288 && a_rs == 1 && a_cs >= 288 && b_cs == 1 && alpha == 1.f && !b_colsum && !bias && !residual &&
      N >= 8192 && N % 32 == 0 && K % 32 == 0 && K >= 4096 && m4(a_cs) && m4(b_rs) && al16(A) && al16(B) &&
      (int64_t)288 * ldc < (1ll << 30) && (int64_t)32 * b_rs < (1ll << 30) && !getenv("PDN_GEMM_NO_OUTRES")) {
    int nw = 8, kps = K;
    const int splits = pdn_gemm_outres_tn_plan(N, K, &nw, &kps);
    const bool direct = splits == 1 && beta == 0.f;
    if (direct || (int64_t)splits * M * N <= ws_cap) {
      bool prof;
      ProfRec rec;
      {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        prof = g_prof_on;
      }
      if (prof) {
        PDN_HIP(hipEventCreate(&rec.e0));
        PDN_HIP(hipEventCreate(&rec.e1));
        rec.flops = 2.0 * M * (double)N * (double)K;
        rec.family = 4;
        PDN_HIP(hipEventRecord(rec.e0, st));
      }
      if (getenv("PDN_GEMM_DEBUG"))
        fprintf(stderr, "pdn_gemm_f32 M=%d N=%d K=%d -> output-resident TN (splits %d)\n", M, N, K, splits);
      int rc = direct ? pdn_gemm_outres_tn_launch(A, B, C, N, K, a_cs, b_rs, ldc, 0, nw, kps, stream)
                      : pdn_gemm_outres_tn_launch(A, B, (float*)workspace, N, K, a_cs, b_rs, N, (int64_t)M * N, nw, kps, stream);
      if (rc) return rc;
      if (!direct) {
        p.splits = splits;
        const int64_t total = (int64_t)M * N;
        const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        const int rvec = (ldc % 4 == 0) && al16(C) && al16(workspace);
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, p, 1, rvec);
        PDN_LAUNCH_CHECK();
      }
      if (prof) {
        PDN_HIP(hipEventRecord(rec.e1, st));
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof.push_back(rec);
      }
      return PDN_OK;
    }
  }
  // ---- the packed weight gradients of a transformer block out of the model width: x^T (288 x K) against dq | dk | dv
  // (three 288-column blocks of one (K x 864) matrix) or dgate | dup (two 768-column blocks), one output per block.
  // The same output-resident TN kernel, K split 40-64 ways so that the grid fills the chip, slabs in the batched
  // layout of the split-K reduce (which adds beta * C).  Against the wave-streaming kernel: 864 columns -16 %,
  // 1536 columns -11 % (60 instead of 170 operand bytes per MFMA; the slab pass costs 30-40 us).
  {
    const bool one_dim = nb1 == 1 || nb2 == 1;
    const int64_t a_bs = nb1 == 1 ? a_bs2 : a_bs1, b_bs = nb1 == 1 ? b_bs2 : b_bs1;
    const int64_t n_all = (int64_t)N * nbatch;
    if (nbatch > 1 && one_dim && M == 288 && a_bs == 0 && b_bs == N && a_rs == 1 && a_cs >= 288 && b_cs == 1 &&
        b_rs >= n_all && alpha == 1.f && !b_colsum && !bias && !residual && N % 32 == 0 && n_all >= 768 && n_all <= 8192 &&
        K % 32 == 0 && K >= 16384 && m4(a_cs) && m4(b_rs) && al16(A) && al16(B) && !getenv("PDN_GEMM_NO_OUTRES")) {
      int nw = 8, kps = K;
      const int splits = pdn_gemm_outres_tn_plan((int)n_all, K, &nw, &kps);
      if (splits > 1 && (int64_t)splits * M * n_all <= ws_cap && nbatch * splits <= 65535) {
        bool prof;
        ProfRec rec;
        {
          std::lock_guard<std::mutex> lk(g_prof_mu);
          prof = g_prof_on;
        }
        if (prof) {
          PDN_HIP(hipEventCreate(&rec.e0));
          PDN_HIP(hipEventCreate(&rec.e1));
          rec.flops = 2.0 * M * (double)n_all * (double)K;
          rec.family = 4;
          PDN_HIP(hipEventRecord(rec.e0, st));
        }
        if (getenv("PDN_GEMM_DEBUG"))
          fprintf(stderr, "pdn_gemm_f32 M=%d N=%lld K=%d -> output-resident TN (%d blocks, splits %d)\n", M,
                  (long long)n_all, K, nbatch, splits);
        int rc = pdn_gemm_outres_tn_blocks_launch(A, B, (float*)workspace, (int)n_all, K, a_cs, b_rs, N, nw, kps, stream);
        if (rc) return rc;
        p.splits = splits;
        const int64_t total = (int64_t)M * n_all;
        const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        const int rvec = (ldc % 4 == 0) && al16(C) && m4(c_bs1) && m4(c_bs2) && al16(workspace);
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, p, nbatch, rvec);
        PDN_LAUNCH_CHECK();
        if (prof) {
          PDN_HIP(hipEventRecord(rec.e1, st));
          std::lock_guard<std::mutex> lk(g_prof_mu);
          g_prof.push_back(rec);
        }
        return PDN_OK;
      }
    }
  }
  // ---- weight-gradient form (small output, long K): wave-streaming kernel --------------------
  constexpr int SNW = 8;                                  // waves per streaming workgroup
  bool use_stream = false;
  int sshape = 0;
  // Which weight gradients it takes (tools/gemm_fc_sweep.py, same process, warm clocks, K = 4096 ... 65536):
  //   * outputs BELOW 2^20 elements: at 1024 x 1024 -- config 2's second layer -- the tiled kernel with its k-split measured
  //     1065 us against 1144 us, and the MLP step 5.44 -> 5.32 ms (PDN_GEMM_STREAM_MAX overrides the bound);
  //   * from 2^18 elements on only where the tiled kernel's 128 / 96-wide tiles would PAD the output by more than 4 %:
  //     784 x 1024 (config 2's first layer: 9 x 8 tiles of 96 x 128 = +10 %) 954 us here against 1030-1080 us tiled, but
  //     512 x 512 and 512 x 1536 (the Transformer example, the dim-512 Llama) 38 / 85 us tiled against 45 / 109 us here at
  //     K = 5632 and 280 / 804 against 330 / 890 us at K = 65536; below 2^18 (288 x 288, 288 x 768) this kernel wins at every K.
  bool stream_fits = (int64_t)M * N * nbatch <= (getenv("PDN_GEMM_STREAM_MAX") ? atol(getenv("PDN_GEMM_STREAM_MAX")) : (1 << 20) - 1);
  if (stream_fits && (int64_t)M * N * nbatch >= (1 << 18) && !getenv("PDN_GEMM_STREAM_MAX")) {
    static const int kPadTiles[3][2] = {{128, 128}, {128, 96}, {96, 128}};
    double pad = 1e30;
    for (const auto& t : kPadTiles)
      pad = std::min(pad, (double)(cdiv64(M, t[0]) * t[0]) * (double)(cdiv64(N, t[1]) * t[1]) / ((double)M * N));
    stream_fits = pad > 1.04;
  }
  if (!ext_on && !a_kin && !b_kin && a_rs == 1 && b_cs == 1 && !b_colsum && K >= 2048 && stream_fits && nbatch <= 64 &&
      !getenv("PDN_GEMM_NO_STREAM")) {
    // tile shape of the LDS-staged kernel (in 32-row / 32-column MFMA tiles): 3 x 3 unless another one pads the output
    // less (784 x 1024: 9 x 11 tiles of 96 x 96 = 912 K accumulators, 5 x 16 of 160 x 64 = 819 K)
    static const int kShapes[5][2] = {{3, 3}, {5, 2}, {2, 5}, {4, 2}, {2, 4}};
    // time per accumulator of the padded output relative to 3 x 3 (tools/mlp_dw_probe.py, 65536 tokens: 784 x 1024,
    // 1024 x 1024, 768 x 768, 512 x 2048); an exact 3 x 3 tiling takes the DMA-staged kernel
    static const double kShapeCost[5] = {1.0, 1.05, 1.08, 1.07, 1.02};
    sshape = 0;
    if (vec) {
      double best_area = (double)cdiv64(M, 96) * cdiv64(N, 96) * 96 * 96 * ((M % 96 == 0 && N % 96 == 0) ? 0.975 : 1.0);
      for (int c = 1; c < 5; ++c) {
        const int th = kShapes[c][0] * 32, tw = kShapes[c][1] * 32;
        const double area = (double)cdiv64(M, th) * cdiv64(N, tw) * th * tw * kShapeCost[c];
        if (area < best_area) { best_area = area; sshape = c; }
      }
      if (const char* e = getenv("PDN_GEMM_STREAM_SHAPE")) { const int c = atoi(e); if (c >= 0 && c < 5) sshape = c; }
    }
    const int sth = kShapes[sshape][0] * 32, stw = kShapes[sshape][1] * 32;
    const int tm = (int)cdiv64(M, sth), tn = (int)cdiv64(N, stw), tiles = tm * tn * nbatch;
    if (tiles <= 256) {
      // one 8-wave workgroup per CU (2 waves / SIMD): rounds of 256 blocks; each extra slab costs a
      // write + read of the output in the reduce pass (~1.7 KB/clk), plus its launch
      double best_cost = 1e300;
      int best_kps = 0, best_sp = 1;
      const int smax = (int)std::min<int64_t>(64, K / (SNW * 16));
      for (int s = 1; s <= smax; ++s) {
        if (s > 1 && (int64_t)s * M * N * nbatch > ws_cap) break;
        const int kw = (int)cdiv64(cdiv64(K, (int64_t)s * SNW), 8) * 8;
        const int kps = kw * SNW, sp = (int)cdiv64(K, kps);
        const double rounds = (double)cdiv64((int64_t)tiles * sp, 256);
        double cost = rounds * (kw * 0.5 * (kShapes[sshape][0] * kShapes[sshape][1]) * 64 * (SNW / 4.0) + 6000.0);
        if (sp > 1) cost += (double)sp * M * N * nbatch * 8.0 / 1700.0 + 5000.0;
        if (cost < best_cost) { best_cost = cost; best_kps = kps; best_sp = sp; }
      }
      if (const char* e = getenv("PDN_GEMM_STREAM_SPLITS")) {     // tuning override
        const int s = atoi(e);
        if (s >= 1 && (s == 1 || (int64_t)s * M * N * nbatch <= ws_cap)) {
          const int kw = (int)cdiv64(cdiv64(K, (int64_t)s * SNW), 8) * 8;
          best_kps = kw * SNW; best_sp = (int)cdiv64(K, best_kps);
        }
      }
      use_stream = true;
      p.tiles_m = tm; p.tiles_n = tn; p.k_per_split = best_kps; p.splits = best_sp;
    }
  }

  // ---- pick tile shape and k-split with a small cost model -----------------------------
  // The matrix pipes of a CU are shared by its resident workgroups, so MFMA throughput is
  // counted per CU (256), not per residency slot: n blocks take ceil(n/256) block-times until
  // the grid is large enough (>= 4 rounds) for the dispatcher to even the load out.
  int best = -1, best_splits = 1;
  double best_cost = 1e300;
  static const bool no_fixed = getenv("PDN_GEMM_NO_FIXED") != nullptr;      // A/B switch for the residency-round term
  for (int c = 0; c < kNumCfgs && !use_stream; ++c) {
    if (!vec && c != kScalarCfg) continue;  // scalar staging: 64x64 only
    if (ext_on && !(c == 0 || c == 3 || c == 6 || c == 7 || c == 8 || c == kScalarCfg)) continue;   // the tile shapes built with MASKS
    if (mask_colsum && c == 8) continue;    // (a lane keeps its columns over a band only when 64 % (8 WN) == 0)
    // 128 x 256: very wide outputs -- and (round 6) the 1024-wide Linear + ReLU layers of config 2 at chip-filling row
    // counts, where it measures 0.86 of the fp32-MFMA peak against 0.71-0.80 for 128 x 128 (tools/gemm_fc_sweep.py)
    if (c == 7 && N < 2048 && !(ext_on && N >= 1024 && N % 256 == 0 && M >= 16384)) continue;
    if (c == 6 && ext_on && !(N >= 1024 && M >= 16384)) continue;
    if (c == 8 && N < 768) continue;        // 256-row tiles lose on narrow outputs (one block per CU)
    const int BM = kCfgs[c].waves_m * kCfgs[c].wm * 32, BN = kCfgs[c].waves_n * kCfgs[c].wn * 32;
    const int bk = kCfgs[c].bk;
    const int64_t tiles = cdiv64(M, BM) * cdiv64(N, BN) * nbatch;
    // 128x96 with BK 16 keeps THREE workgroups per CU: worth it exactly when two-per-CU would
    // leave a half-empty last round of short blocks (measured: 73 -> 58 us on 32768x288x288)
    if (c == 12 && !(tiles > 512 && tiles <= 768 && K <= 1024 && a_kin && !b_kin)) continue;
    const int ktiles = (int)cdiv64(K > 0 ? K : 1, bk);
    for (int s = 1; s <= 64; s *= 2) {
      if (s > 1 && (b_colsum || ext_on)) break;
      if (s > 1) {
        if (ktiles * bk / s < 128) break;
        if ((int64_t)s * M * N * nbatch > ws_cap) break;
      }
      const int kps = (int)cdiv64(ktiles, s);
      // Model (fitted to tools/gemm_sweep_dw.py and the tile sweep): a block keeps `waves_pb`
      // SIMDs busy for BM*BN*K/waves_pb MFMA-cycles; blocks are dealt round-robin over the 256
      // CUs, so a grid needs ceil(blocks/slots) rounds until it is large enough to even out; a CU
      // with at least two resident blocks hides their barriers/prologues behind each other
      // (~0.7 -> 1.0 matrix-pipe duty), which is why ~480 blocks beat 240 longer ones.
      const double waves_pb = kCfgs[c].waves_m * kCfgs[c].waves_n;
      const double blocks = (double)tiles * s;
      const double slots = 256.0 * (4.0 / waves_pb);
      const double rounds = blocks >= 4 * slots ? blocks / slots : (double)cdiv64((int64_t)blocks, (int64_t)slots);
      const double wps = blocks * waves_pb / 1024.0;   // resident waves per SIMD in a round
      const double util = wps >= 1.875 ? 1.0 : (wps > 1.0 ? 0.7 + 0.3 * (wps - 1.0) / 0.875 : 0.7);
      const double eff = kCfgs[c].eff;
      double cost = rounds * ((double)BM * BN * (kps * bk + 64) / waves_pb * 4.0 / (eff * util));
      // ... plus what a block costs besides its MFMAs (first tiles in, accumulators out: ~25k cycles = 10 us, fitted at
      // 16384 x 288 x {288, 768, 864}: 38 us for 384 blocks of 128 x 96 against 46 us for 768 of 64 x 128), paid once
      // per RESIDENCY round -- co-resident blocks overlap theirs -- which is what makes few fat blocks win on short
      // products even when they leave CUs half loaded (round 3; the unit is 1/156 matrix-pipe cycle)
      if (!no_fixed) {
        const double per_cu = c == 12 ? 3.0 : 2.0;
        cost += (double)cdiv64((int64_t)blocks, (int64_t)(256.0 * per_cu)) * 3.9e6;
      }
      if (s > 1) cost += (double)M * N * nbatch * (s + 1) * 0.5 + 1.0e6;   // slab pass + extra launch
      if (cost < best_cost) { best_cost = cost; best = c; best_splits = s; }
    }
  }
  PDN_CHECK_ARG(best >= 0 || use_stream, "pdn_gemm_f32: no tile configuration");
  if (!use_stream && best_splits == 1 && !b_colsum && !ext_on && K >= 8192) {
    // 2-4 very long blocks per CU: the last one of each CU runs without a co-resident partner to
    // hide its barriers behind; halving the blocks evens that out (measured -3..4 % on the
    // 32768 x 288 x 32000 and 288 x 32000 x 32768 products, slab pass included)
    const int BMb = kCfgs[best].waves_m * kCfgs[best].wm * 32, BNb = kCfgs[best].waves_n * kCfgs[best].wn * 32;
    const int64_t blocks = cdiv64(M, BMb) * cdiv64(N, BNb) * nbatch;
    if (blocks > 512 && blocks < 1024 && (int64_t)2 * M * N * nbatch <= ws_cap) best_splits = 2;
  }
  // Mid-size products that 128 x 128 tiles cannot spread over the chip (fewer than 256 of them) or whose contraction is
  // short (K <= 512): 64 x 64 tiles with a k-split.  Config 3's fully connected layers (4096 x 3200 x 500 in its three
  // forms) measured inside the step: 1.390 -> 1.335 ms with `5,8` for all of them, every other choice in between
  // (round 6; the cost model above prices a block's fixed cost too high for these).
  // The k-split pays in the weight-gradient form only (long contraction over the rows): with the rows on the output side
  // 5632 x 512 over K = 1536 measured 85 us unsplit against 94 / 96 / 103 us split 2 / 4 / 8 ways, 8192 x 1024 over 1024
  // 143 against 161 / 177 / 233 us (tools/gemm_fc_sweep.py) -- and the Linear + ReLU products (mask epilogues: no split) take
  // the same tiles: 5632 x 512 -> 1536 with its ReLU 104 -> 8x us.
  if (!use_stream && !b_colsum && vec && nbatch == 1 && (int64_t)M * N >= (1 << 20) && (int64_t)M * N <= (1 << 24) &&
      K <= 4096 && K >= 128 && (cdiv64(M, 128) * cdiv64(N, 128) < 256 || K <= 512)) {
    int sp = (a_kin || ext_on) ? 1 : 8;
    while (sp > 1 && (K / sp < 64 || (int64_t)sp * M * N > ws_cap)) sp >>= 1;
    best = kScalarCfg;
    best_splits = sp;
  }
  if (use_stream) best = 0;
  else if (const char* e = getenv("PDN_GEMM_CFG")) {            // tuning override: "<cfg>[,<splits>]"
    int c = -1, sp = 0;
    if (sscanf(e, "%d,%d", &c, &sp) >= 1 && c >= 0 && c < kNumCfgs && vec &&
        (!ext_on || c == 0 || c == 3 || c == 6 || c == 7 || (c == 8 && !mask_colsum) || c == kScalarCfg)) {
      best = c;
      if (sp >= 1 && (sp == 1 || ((int64_t)sp * M * N * nbatch <= ws_cap && !b_colsum && !ext_on))) best_splits = sp;
    }
  }
  if (getenv("PDN_GEMM_DEBUG"))
    fprintf(stderr, "pdn_gemm_f32 M=%d N=%d K=%d nb=%d akin=%d bkin=%d vec=%d -> %s cfg=%d splits=%d\n", M, N, K, nbatch,
            (int)a_kin, (int)b_kin, (int)vec, use_stream ? "stream" : "tiled", best, best_splits);
  const TileCfg& cfg = kCfgs[best];
  const int BM = cfg.waves_m * cfg.wm * 32, BN = cfg.waves_n * cfg.wn * 32;
  if (!use_stream) {
    p.tiles_m = (int)cdiv64(M, BM);
    p.tiles_n = (int)cdiv64(N, BN);
    const int ktiles = (int)cdiv64(K > 0 ? K : 1, cfg.bk);
    p.k_per_split = (int)cdiv64(ktiles, best_splits) * cfg.bk;
    p.splits = (int)cdiv64(K > 0 ? K : 1, p.k_per_split);
    if (p.splits < 1) p.splits = 1;
  }
  PDN_CHECK_ARG((int64_t)nbatch * p.splits <= 65535, "pdn_gemm_f32: batch*splits too large (%d)",
                nbatch * p.splits);
  dim3 grid(p.tiles_m * p.tiles_n, nbatch * p.splits);

  bool prof;
  ProfRec rec;
  {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    prof = g_prof_on;
  }
  if (prof) {
    PDN_HIP(hipEventCreate(&rec.e0));
    PDN_HIP(hipEventCreate(&rec.e1));
    rec.flops = 2.0 * M * N * (double)K * nbatch;
    rec.family = use_stream ? 1 : 0;
    PDN_HIP(hipEventRecord(rec.e0, st));
  }

  if (use_stream) {
    const dim3 sgrid(p.tiles_m * p.tiles_n * p.splits, nbatch);
    const bool exact = M % 96 == 0 && N % 96 == 0;
    if (vec && sshape != 0) {                            // (set for aligned operands only)
      switch (sshape) {
        case 1: hipLaunchKernelGGL((gemm_tn_stream_lds_kernel<5, 2, SNW, true>), sgrid, dim3(SNW * 64), 0, st, p); break;
        case 2: hipLaunchKernelGGL((gemm_tn_stream_lds_kernel<2, 5, SNW, true>), sgrid, dim3(SNW * 64), 0, st, p); break;
        case 3: hipLaunchKernelGGL((gemm_tn_stream_lds_kernel<4, 2, SNW, true>), sgrid, dim3(SNW * 64), 0, st, p); break;
        default: hipLaunchKernelGGL((gemm_tn_stream_lds_kernel<2, 4, SNW, true>), sgrid, dim3(SNW * 64), 0, st, p); break;
      }
    } else if (vec && !getenv("PDN_GEMM_STREAM_DIRECT")) {      // 16-byte loads staged through LDS
      if (exact && !getenv("PDN_GEMM_STREAM_NODMA"))
        hipLaunchKernelGGL((gemm_tn_stream_dma_kernel<3, 3, SNW>), sgrid, dim3(SNW * 64), 0, st, p);
      else if (exact) hipLaunchKernelGGL((gemm_tn_stream_lds_kernel<3, 3, SNW, false>), sgrid, dim3(SNW * 64), 0, st, p);
      else hipLaunchKernelGGL((gemm_tn_stream_lds_kernel<3, 3, SNW, true>), sgrid, dim3(SNW * 64), 0, st, p);
    } else {                                             // unaligned operands: dword loads, no LDS
      if (exact) hipLaunchKernelGGL((gemm_tn_stream_kernel<3, 3, SNW, false>), sgrid, dim3(SNW * 64), 0, st, p);
      else hipLaunchKernelGGL((gemm_tn_stream_kernel<3, 3, SNW, true>), sgrid, dim3(SNW * 64), 0, st, p);
    }
  } else if (ext_on) {
    if (!vec) launch_layout<2, 2, 1, 1, 32, false, true>(p, a_kin, b_kin, grid, st);
    else switch (best) {
      case 0: launch_layout<2, 2, 2, 2, 32, true, true>(p, a_kin, b_kin, grid, st); break;
      case 3: launch_layout<4, 1, 1, 2, 32, true, true>(p, a_kin, b_kin, grid, st); break;
      case 6: launch_layout<2, 2, 4, 2, 16, true, true>(p, a_kin, b_kin, grid, st); break;
      case 7: launch_layout<2, 2, 2, 4, 16, true, true>(p, a_kin, b_kin, grid, st); break;
      case 8: launch_layout<4, 1, 2, 3, 16, true, true>(p, a_kin, b_kin, grid, st); break;
      default: launch_layout<2, 2, 1, 1, 32, true, true>(p, a_kin, b_kin, grid, st); break;
    }
  } else if (vec) {
    switch (best) {
      case 0: launch_layout<2, 2, 2, 2, 32, true>(p, a_kin, b_kin, grid, st); break;
      case 1: launch_layout<4, 1, 1, 3, 32, true>(p, a_kin, b_kin, grid, st); break;
      case 2: launch_layout<1, 4, 3, 1, 32, true>(p, a_kin, b_kin, grid, st); break;
      case 3: launch_layout<4, 1, 1, 2, 32, true>(p, a_kin, b_kin, grid, st); break;
      case 4: launch_layout<1, 4, 2, 1, 32, true>(p, a_kin, b_kin, grid, st); break;
      case 6: launch_layout<2, 2, 4, 2, 16, true>(p, a_kin, b_kin, grid, st); break;
      case 7: launch_layout<2, 2, 2, 4, 16, true>(p, a_kin, b_kin, grid, st); break;
      case 8: launch_layout<4, 1, 2, 3, 16, true>(p, a_kin, b_kin, grid, st); break;
      case 9: launch_layout<1, 4, 3, 2, 16, true>(p, a_kin, b_kin, grid, st); break;
      case 10: launch_layout<2, 1, 1, 3, 32, true>(p, a_kin, b_kin, grid, st); break;
      case 11: launch_layout<1, 2, 3, 1, 32, true>(p, a_kin, b_kin, grid, st); break;
      case 12: launch_layout<4, 1, 1, 3, 16, true>(p, a_kin, b_kin, grid, st); break;
      default: launch_layout<2, 2, 1, 1, 32, true>(p, a_kin, b_kin, grid, st); break;
    }
  } else {
    launch_layout<2, 2, 1, 1, 32, false>(p, a_kin, b_kin, grid, st);
  }
  PDN_LAUNCH_CHECK();
  if (p.splits > 1) {
    const int64_t total = (int64_t)M * N * nbatch;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    const int rvec = (N % 4 == 0) && (ldc % 4 == 0) && al16(C) && m4(c_bs1) && m4(c_bs2) && (!bias || al16(bias)) &&
                     (!residual || al16(residual)) && al16(workspace);
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, p, nbatch, rvec);
    PDN_LAUNCH_CHECK();
  }
  if (prof) {
    PDN_HIP(hipEventRecord(rec.e1, st));
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back(rec);
  }
  return PDN_OK;
}

extern "C" int pdn_gemm_f32(int M, int N, int K, float alpha, const float* A, int64_t a_rs,
                            int64_t a_cs, const float* B, int64_t b_rs, int64_t b_cs, float beta,
                            float* C, int64_t ldc, const float* bias, int nb1, int nb2,
                            int64_t a_bs1, int64_t a_bs2, int64_t b_bs1, int64_t b_bs2,
                            int64_t c_bs1, int64_t c_bs2, const float* residual,
                            float* b_colsum, int colsum_accumulate, void* workspace,
                            int64_t workspace_bytes, void* stream) {
  return gemm_f32_impl(M, N, K, alpha, A, a_rs, a_cs, B, b_rs, b_cs, beta, C, ldc, bias, nb1, nb2, a_bs1, a_bs2, b_bs1, b_bs2,
                       c_bs1, c_bs2, residual, b_colsum, colsum_accumulate, workspace, workspace_bytes, stream, nullptr, nullptr);
}

// ======================================================================================
// `relu(linear(x))` as one product and its backward without an elementwise pass (examples/pydynet/mnist.py:70-78:
// Linear -> ReLU -> Linear -> ReLU -> Linear; nn/functional.py:31-32 relu = maximum(0., x), tensor.py:808-814 its gradient
// passes where out == x, i.e. x >= 0).  The forward stores h = max(0, x W + b) and ONE BIT per element (x W + b >= 0);
// the consumer's input-gradient product applies those bits in its store (pdn_linear_dx_masked_f32), so the gradient
// with respect to the pre-activation is what reaches this layer; anything else uses pdn_relu_mask_bwd_f32.
extern "C" int pdn_relu_mask_supported(int64_t rows, int cols) { return rows > 0 && cols > 0 && cols % 32 == 0; }

extern "C" int pdn_linear_relu_fwd_f32(const float* x, int64_t x_rs, const float* W, int64_t w_rs, int64_t w_cs,
                                       const float* bias, float* h, int64_t ldh, uint32_t* mask, int M, int N, int K,
                                       void* stream) {
  PDN_CHECK_ARG(x && W && h && mask, "pdn_linear_relu_fwd_f32: null operand");
  PDN_CHECK_ARG(pdn_relu_mask_supported(M, N), "pdn_linear_relu_fwd_f32: out features must be a multiple of 32 (N = %d)", N);
  pdn_count(PDN_CNT_LINEAR_RELU_FWD);
  return gemm_f32_impl(M, N, K, 1.f, x, x_rs, 1, W, w_rs, w_cs, 0.f, h, ldh, bias, 1, 1, 0, 0, 0, 0, 0, 0, nullptr, nullptr, 0,
                       nullptr, 0, stream, mask, nullptr);
}

// dx (M x fin) = mask o (g (M x fout) W^T + existing);  W is (fin x fout) with strides (w_rs, w_cs).
// colsum_partials (may be null): (ceil(M / 32) x fin) floats, row b = the column sums of dx over rows 32 b .. 32 b + 31 --
// the bias gradient of the layer below is their sum over b (no pass over dx for it).
extern "C" int pdn_linear_dx_masked_f32(const float* g, int64_t g_rs, const float* W, int64_t w_rs, int64_t w_cs,
                                        float* dx, int64_t ld, const float* existing, const uint32_t* mask,
                                        float* colsum_partials, int M, int fin, int fout, void* stream) {
  PDN_CHECK_ARG(g && W && dx && mask, "pdn_linear_dx_masked_f32: null operand");
  PDN_CHECK_ARG(pdn_relu_mask_supported(M, fin), "pdn_linear_dx_masked_f32: in features must be a multiple of 32 (%d)", fin);
  PDN_CHECK_ARG(!colsum_partials || ((uintptr_t)colsum_partials & 15) == 0, "pdn_linear_dx_masked_f32: unaligned partials");
  pdn_count(PDN_CNT_LINEAR_DX_MASKED);
  return gemm_f32_impl(M, fin, fout, 1.f, g, g_rs, 1, W, w_cs, w_rs, 0.f, dx, ld, nullptr, 1, 1, 0, 0, 0, 0, 0, 0, existing,
                       nullptr, 0, nullptr, 0, stream, nullptr, mask, colsum_partials);
}

__global__ __launch_bounds__(256) void relu_mask_bwd_kernel(const float4* __restrict__ g, const uint32_t* __restrict__ mask,
                                                            float4* __restrict__ dz, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const uint32_t w = mask[i >> 3] >> (4 * (i & 7));
    float4 v = g[i];
    v.x = (w & 1u) ? v.x : 0.f; v.y = (w & 2u) ? v.y : 0.f; v.z = (w & 4u) ? v.z : 0.f; v.w = (w & 8u) ? v.w : 0.f;
    dz[i] = v;
  }
}

// dz = mask o g over a contiguous (rows x cols) array, cols a multiple of 32 (g == dz allowed)
extern "C" int pdn_relu_mask_bwd_f32(const float* g, const uint32_t* mask, float* dz, int64_t rows, int cols, void* stream) {
  PDN_CHECK_ARG(g && mask && dz, "pdn_relu_mask_bwd_f32: null operand");
  PDN_CHECK_ARG(pdn_relu_mask_supported(rows, cols) && (((uintptr_t)g | (uintptr_t)dz) & 15) == 0,
                "pdn_relu_mask_bwd_f32: cols must be a multiple of 32 and the arrays 16-byte aligned");
  const int64_t n4 = rows * cols / 4;
  const int blocks = (int)std::min<int64_t>((n4 + 255) / 256, 8192);
  hipLaunchKernelGGL(relu_mask_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)g, mask, (float4*)dz, n4);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

// ======================================================================================
// SwiGLU in the stores of the tiled kernel, for the model widths the row-resident kernels (csrc/gemm_rowres.hip,
// contraction 288) do not take -- llm/llama/model.py:56-58 forward, its backward through `dh = dy W_down^T`
// (tensor.py:670).  Forward: [Wg | Wu] is packed once per call with its columns in alternating groups of 32 gate / 32 up
// columns (zero-padded to whole 256-column tiles), so a wave's accumulator block holds a gate group and ITS up group side
// by side; the store writes gate, up (the saved [gate | up] layout) and h = silu(gate) * up.  Backward: the store of
// dy W_down^T reads the saved gate / up and writes d[gate | up]; dh never exists.
__global__ __launch_bounds__(256) void swiglu_pack_weights_kernel(const float* __restrict__ wg, int64_t w_stride, float* __restrict__ out,
                                                                  int K, int F, int Np) {
  const int64_t total = (int64_t)K * (Np / 4);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int k = (int)(i / (Np / 4)), c = (int)(i - (int64_t)k * (Np / 4)) * 4;       // packed column c .. c + 3
    const int grp = c >> 6, up = (c >> 5) & 1, col = 32 * grp + (c & 31);             // source column in Wg / Wu
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < F) v = *reinterpret_cast<const float4*>(wg + (up ? w_stride : 0) + (int64_t)k * F + col);
    *reinterpret_cast<float4*>(out + (int64_t)k * Np + c) = v;
  }
}

static int swiglu_packed_cols(int F) { return (2 * F + 255) / 256 * 256; }

extern "C" int pdn_gateup_swiglu_tiled_supported(int M, int F, int K) {
  return M >= 4096 && M % 128 == 0 && F >= 256 && F % 32 == 0 && K >= 64 && K % 4 == 0 && !getenv("PDN_NO_TILED_SWIGLU");
}
extern "C" int64_t pdn_gateup_swiglu_tiled_workspace_bytes(int F, int K) { return (int64_t)K * swiglu_packed_cols(F) * 4; }

static void swiglu_params(GemmParams& p, int M, int N, int K, int BM, int BN, int BK) {
  p.bias = nullptr; p.residual = nullptr; p.colsum = nullptr; p.colsum_acc = 0;
  p.relu_mask = nullptr; p.grad_mask = nullptr; p.mask_colsum = nullptr; p.ws = nullptr;
  p.M = M; p.N = N; p.K = K; p.nb2 = 1;
  p.a_bs1 = p.a_bs2 = p.b_bs1 = p.b_bs2 = p.c_bs1 = p.c_bs2 = 0;
  p.alpha = 1.f; p.beta = 0.f; p.splits = 1;
  p.k_per_split = (K + BK - 1) / BK * BK;
  p.tiles_m = (M + BM - 1) / BM; p.tiles_n = (N + BN - 1) / BN;
}

// gu (M x 2 F) = x [Wg | Wu], h (M x F) = silu(gate) * up;  Wg, Wu (K x F) row-major, w_stride floats apart
extern "C" int pdn_gateup_swiglu_tiled_fwd_f32(const float* x, int64_t ldx, const float* w_gate, int64_t w_stride, float* gu,
                                               float* h, int M, int F, int K, void* workspace, int64_t workspace_bytes,
                                               void* stream) {
  PDN_CHECK_ARG(x && w_gate && gu && h && workspace, "pdn_gateup_swiglu_tiled_fwd_f32: null operand");
  PDN_CHECK_ARG(pdn_gateup_swiglu_tiled_supported(M, F, K) && ldx % 4 == 0 && w_stride % 4 == 0 &&
                    ((((uintptr_t)x | (uintptr_t)w_gate | (uintptr_t)gu | (uintptr_t)h | (uintptr_t)workspace) & 15) == 0),
                "pdn_gateup_swiglu_tiled_fwd_f32: unsupported shape or alignment (M %d, F %d, K %d)", M, F, K);
  const int Np = swiglu_packed_cols(F);
  if (workspace_bytes < (int64_t)K * Np * 4) { pdn_set_error("pdn_gateup_swiglu_tiled_fwd_f32: workspace too small"); return PDN_EWORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  float* packed = (float*)workspace;
  hipLaunchKernelGGL(swiglu_pack_weights_kernel, dim3((unsigned)std::min<int64_t>(((int64_t)K * (Np / 4) + 255) / 256, 2048)), dim3(256), 0, st,
                     w_gate, w_stride, packed, K, F, Np);
  PDN_LAUNCH_CHECK();
  GemmParams p;
  p.A = x; p.B = packed; p.C = gu; p.a_rs = ldx; p.a_cs = 1; p.b_rs = Np; p.b_cs = 1; p.ldc = 2 * (int64_t)F;
  p.swi_h = h; p.swi_gu = nullptr; p.swi_ldh = F; p.swi_F = F;
  int tok = pdn_gemm_prof_begin(0, 2.0 * M * 2.0 * F * K, 0.0, stream);
  if (Np >= 2048) {                                            // 128 x 256 tiles
    swiglu_params(p, M, Np, K, 128, 256, 16);
    hipLaunchKernelGGL((gemm_f32_mfma_kernel<2, 2, 2, 4, 16, true, false, true, false, false, 1>), dim3(p.tiles_m * p.tiles_n), dim3(256), 0, st, p);
  } else {                                                     // 128 x 128
    swiglu_params(p, M, Np, K, 128, 128, 32);
    hipLaunchKernelGGL((gemm_f32_mfma_kernel<2, 2, 2, 2, 32, true, false, true, false, false, 1>), dim3(p.tiles_m * p.tiles_n), dim3(256), 0, st, p);
  }
  PDN_LAUNCH_CHECK();
  pdn_gemm_prof_end(tok, stream);
  pdn_count(PDN_CNT_TILED_SWIGLU_FWD);
  return PDN_OK;
}

extern "C" int pdn_swiglu_bwd_tiled_supported(int M, int F, int K) {
  return M >= 4096 && M % 128 == 0 && F >= 256 && F % 4 == 0 && K >= 64 && K % 4 == 0 && !getenv("PDN_NO_TILED_SWIGLU");
}

// dgu (M x 2 F) = d[gate | up] from dh = dy (M x K) W_down^T, W_down (F x K) row-major, and the saved gu (M x 2 F)
extern "C" int pdn_swiglu_bwd_tiled_f32(const float* dy, int64_t ldy, const float* w_down, const float* gu, float* dgu, int M,
                                        int F, int K, void* stream) {
  PDN_CHECK_ARG(dy && w_down && gu && dgu, "pdn_swiglu_bwd_tiled_f32: null operand");
  PDN_CHECK_ARG(pdn_swiglu_bwd_tiled_supported(M, F, K) && ldy % 4 == 0 &&
                    ((((uintptr_t)dy | (uintptr_t)w_down | (uintptr_t)gu | (uintptr_t)dgu) & 15) == 0),
                "pdn_swiglu_bwd_tiled_f32: unsupported shape or alignment (M %d, F %d, K %d)", M, F, K);
  hipStream_t st = (hipStream_t)stream;
  GemmParams p;
  p.A = dy; p.B = w_down; p.C = dgu; p.a_rs = ldy; p.a_cs = 1; p.b_rs = 1; p.b_cs = K; p.ldc = 2 * (int64_t)F;
  p.swi_h = nullptr; p.swi_gu = gu; p.swi_ldh = 0; p.swi_F = F;
  int tok = pdn_gemm_prof_begin(0, 2.0 * M * (double)F * K, 0.0, stream);
  swiglu_params(p, M, F, K, 128, 128, 32);
  hipLaunchKernelGGL((gemm_f32_mfma_kernel<2, 2, 2, 2, 32, true, true, true, false, false, 2>), dim3(p.tiles_m * p.tiles_n), dim3(256), 0, st, p);
  PDN_LAUNCH_CHECK();
  pdn_gemm_prof_end(tok, stream);
  pdn_count(PDN_CNT_TILED_SWIGLU_BWD);
  return PDN_OK;
}

// ======================================================================================
// Backward of `linear -> cross entropy` (llm/llama/model.py:179 + nn/functional.py:364-381) without the
// (rows x V) gradient of the logits in memory: both products form it from the saved logits as they consume it,
//   dlogits[t][v] = (exp(logits[t][v] - lse[t]) - [v == targets[t]]) * gscale * (upstream ? upstream[0] : 1)
//   dx (rows x in)  = dlogits W^T (+ dx_residual)          W (in x V) row-major
//   dW (in x V)     = dw_beta * dW + x^T dlogits           x (rows x in), row stride ldx
//   dbias (V)       = db_beta * dbias + column sums of dlogits
// in = 288 (output-resident kernels, csrc/gemm_outres.hip).  The cross-entropy pass then only reads the logits
// (row statistics) instead of reading them and writing a gradient of the same size.
// ======================================================================================
int pdn_outres_ce_dx_launch(const float* logits, int64_t ldl, const float* lse, const int64_t* targets, float gscale,
                            const float* gdev, const float* W, int64_t ldw, float* dx, int64_t ldc,
                            const float* residual, int M, int V, void* workspace, int64_t workspace_bytes, void* stream);
int pdn_outres_ce_dw_launch(const float* X, const float* logits, float* C, int N, int K, int64_t ldx, int64_t ldg,
                            int64_t ldc, int64_t slab, int nw, int k_per_split, const float* lse,
                            const int64_t* targets, float gscale, const float* gdev, float* colsum, void* stream);

int pdn_outres_ce_dx_deferred_launch(const float* logits, int64_t ldl, const float* rowmax, int max_parts,
                                     const int64_t* targets, float gscale, const float* W, int64_t ldw, float* dx,
                                     int64_t ldc, float* lse_out, int M, int V, void* workspace, int64_t workspace_bytes,
                                     void* stream);
extern "C" int pdn_linear_ce_dx_deferred_supported(int64_t M, int V, int K);

// Input gradient of `linear -> cross entropy` from the logits and their ROW MAXIMA (pdn_linear_rowmax_fwd_f32: `max_parts`
// vectors of `rows` maxima), leaving the rows' log-sum-exp as a by-product:
//   dx[t] = gscale * (sum_v exp(l[t][v] - max[t]) W[:, v] / Z[t] - W[:, target[t]]),  Z[t] = sum_v exp(l[t][v] - max[t]),
//   lse[t] = max[t] + log Z[t].
// Independent of the upstream gradient (a scalar factor the caller applies), so a training step can run it in the FORWARD
// pass: the loss is then sum(lse - l[target]) and no pass over the logits for the statistics exists.  W (in x V)
// row-major, in = 288.  workspace: pdn_linear_ce_dx_deferred_workspace_bytes (few rows: the vocabulary is cut into
// ranges over the grid, whose unnormalised rows and row sums a second kernel adds up).
extern "C" int pdn_linear_ce_dx_deferred_f32(const float* logits, const float* rowmax, int max_parts, const int64_t* targets,
                                             float gscale, const float* W, float* dx, float* lse, int64_t rows, int V,
                                             int in_features, void* workspace, int64_t workspace_bytes, void* stream) {
  if (rows == 0 || V == 0) return PDN_OK;
  PDN_CHECK_ARG(logits && rowmax && targets && W && dx && lse && max_parts >= 1, "pdn_linear_ce_dx_deferred_f32: null operand");
  if (!pdn_linear_ce_dx_deferred_supported(rows, V, in_features)) {
    pdn_set_error("pdn_linear_ce_dx_deferred_f32: unsupported shape rows=%lld V=%d in=%d", (long long)rows, V, in_features);
    return PDN_EUNSUPPORTED;
  }
  PDN_CHECK_ARG(((((uintptr_t)logits | (uintptr_t)W) & 15) == 0), "pdn_linear_ce_dx_deferred_f32: 16-byte alignment required");
  const int tk = pdn_gemm_prof_begin(3, 2.0 * (double)rows * (double)V * (double)in_features, 0.0, stream);
  const int rc = pdn_outres_ce_dx_deferred_launch(logits, V, rowmax, max_parts, targets, gscale, W, V, dx, in_features, lse,
                                                  (int)rows, V, workspace, workspace_bytes, stream);
  pdn_gemm_prof_end(tk, stream);
  return rc;
}

extern "C" int pdn_linear_ce_supported(int64_t rows, int V, int in_features) {
  return in_features == 288 && V % 32 == 0 && V >= 32 && rows % 32 == 0 && rows >= 32 && rows < (1ll << 31) &&
         (int64_t)288 * V < (1ll << 30);
}

extern "C" int64_t pdn_linear_ce_workspace_bytes(int64_t rows, int V, int in_features) {
  if (!pdn_linear_ce_supported(rows, V, in_features)) return 0;
  int nw, kps;
  const int splits = pdn_gemm_outres_tn_plan(V, (int)rows, &nw, &kps);
  // [weight-gradient slabs | column-sum slabs | input-gradient slabs (K split over the grid when rows are few)]
  return (int64_t)splits * (in_features + 1) * V * 4 + pdn_gemm_outres_workspace_bytes((int)rows, V);
}

extern "C" int pdn_linear_ce_backward_f32(const float* x, int64_t ldx, const float* logits, const float* lse,
                                          const int64_t* targets, float gscale, const float* upstream,
                                          const float* W, float* dx, const float* dx_residual, float* dW,
                                          float dw_beta, float* dbias, float db_beta, int64_t rows, int V,
                                          int in_features, void* workspace, int64_t workspace_bytes, void* stream) {
  if (rows == 0 || V == 0) return PDN_OK;
  PDN_CHECK_ARG(x && logits && lse && targets && W, "pdn_linear_ce_backward_f32: null operand");
  if (!pdn_linear_ce_supported(rows, V, in_features)) {
    pdn_set_error("pdn_linear_ce_backward_f32: unsupported shape rows=%lld V=%d in=%d (in 288, V and rows multiples of 32)",
                  (long long)rows, V, in_features);
    return PDN_EUNSUPPORTED;
  }
  PDN_CHECK_ARG((ldx & 3) == 0 && ldx >= in_features && ((((uintptr_t)x | (uintptr_t)logits | (uintptr_t)W) & 15) == 0),
                "pdn_linear_ce_backward_f32: 16-byte alignment required");
  hipStream_t st = (hipStream_t)stream;
  bool prof;
  {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    prof = g_prof_on;
  }
  // the two products count as launches of the output-resident family in the GEMM profile (bench.py roofline)
  auto prof_begin = [&](ProfRec& rec) {
    if (!prof) return;
    (void)hipEventCreate(&rec.e0);
    (void)hipEventCreate(&rec.e1);
    rec.flops = 2.0 * (double)rows * (double)V * (double)in_features;
    rec.family = 3;
    (void)hipEventRecord(rec.e0, st);
  };
  auto prof_end = [&](ProfRec& rec) {
    if (!prof) return;
    (void)hipEventRecord(rec.e1, st);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back(rec);
  };
  if (dx) {
    ProfRec rec;
    prof_begin(rec);
    int nw_, kps_;
    const int64_t dw_bytes = (int64_t)pdn_gemm_outres_tn_plan(V, (int)rows, &nw_, &kps_) * (in_features + 1) * V * 4;
    const int64_t dx_bytes = pdn_gemm_outres_workspace_bytes((int)rows, V);
    void* dx_ws = (workspace && workspace_bytes >= dw_bytes + dx_bytes && dx_bytes > 0) ? (char*)workspace + dw_bytes : nullptr;
    int rc = pdn_outres_ce_dx_launch(logits, V, lse, targets, gscale, upstream, W, V, dx, in_features, dx_residual,
                                     (int)rows, V, dx_ws, dx_ws ? dx_bytes : 0, stream);
    if (rc) return rc;
    prof_end(rec);
  }
  ProfRec rec_w;
  if (dW || dbias) { prof_begin(rec_w); rec_w.family = 4; }
  if (dW || dbias) {
    int nw, kps;
    const int splits = pdn_gemm_outres_tn_plan(V, (int)rows, &nw, &kps);
    const int64_t need = (int64_t)splits * (in_features + 1) * V * 4;
    PDN_CHECK_ARG(workspace && workspace_bytes >= need, "pdn_linear_ce_backward_f32: workspace too small (%lld < %lld)",
                  (long long)workspace_bytes, (long long)need);
    float* slabs = (float*)workspace;
    float* cs = slabs + (int64_t)splits * in_features * V;
    int rc = pdn_outres_ce_dw_launch(x, logits, slabs, V, (int)rows, ldx, V, V, (int64_t)in_features * V, nw, kps, lse,
                                     targets, gscale, upstream, dbias ? cs : nullptr, stream);
    if (rc) return rc;
    GemmParams p{};
    p.N = V; p.nb2 = 1; p.splits = splits;
    if (dW) {
      p.M = in_features; p.C = dW; p.ldc = V; p.ws = slabs; p.beta = dw_beta;
      const int64_t total = (int64_t)p.M * V;
      const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
      const int rvec = ((uintptr_t)dW & 15) == 0 && ((uintptr_t)slabs & 15) == 0;
      hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, p, 1, rvec);
      PDN_LAUNCH_CHECK();
    }
    if (dbias) {
      p.M = 1; p.C = dbias; p.ldc = V; p.ws = cs; p.beta = db_beta;
      const int blocks = (V + 255) / 256 < 2048 ? (V + 255) / 256 : 2048;
      const int rvec = ((uintptr_t)dbias & 15) == 0 && ((uintptr_t)cs & 15) == 0;
      hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, p, 1, rvec);
      PDN_LAUNCH_CHECK();
    }
    prof_end(rec_w);
  }
  return PDN_OK;
}
