// Ablation timing of gemm_tn_stream_lds_kernel (which part of the loop costs MFMA time?).
// Build: hipcc -O3 --offload-arch=gfx950 stream_ablate.hip -o stream_ablate.bin
#include "../../pydynet_amd/csrc/gemm.hip"
#include <stdio.h>

template <int AB>
static void run_dma(const char* name, GemmParams p, int nblk) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) {
    hipEventRecord(e0);
    for (int it = 0; it < 10; ++it)
      hipLaunchKernelGGL((gemm_tn_stream_dma_kernel<3, 3, 8, AB>), dim3(nblk), dim3(512), 0, 0, p);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fl = 2.0 * p.M * p.N * p.K;
  printf("%-46s %7.1f us  %6.1f TFLOP/s\n", name, ms * 100, fl / (ms / 10) / 1e9);
}

template <int AB>
static void run(const char* name, GemmParams p, int nblk) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) {
    hipEventRecord(e0);
    for (int it = 0; it < 10; ++it)
      hipLaunchKernelGGL((gemm_tn_stream_lds_kernel<3, 3, 8, false, AB>), dim3(nblk), dim3(512), 0, 0, p);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fl = 2.0 * p.M * p.N * p.K;
  printf("%-46s %7.1f us  %6.1f TFLOP/s\n", name, ms * 100, fl / (ms / 10) / 1e9);
}

int main() {
  const int T = 32768;
  for (int N : {288, 768}) {
    GemmParams p = {};
    const int M = 288;
    float *x, *g, *ws;
    hipMalloc(&x, (size_t)T * M * 4); hipMalloc(&g, (size_t)T * N * 4); hipMalloc(&ws, (size_t)64 * M * N * 4);
    hipMemset(x, 0, (size_t)T * M * 4); hipMemset(g, 0, (size_t)T * N * 4);
    p.A = x; p.B = g; p.C = ws; p.ws = ws; p.M = M; p.N = N; p.K = T;
    p.a_rs = 1; p.a_cs = M; p.b_rs = N; p.b_cs = 1; p.ldc = N; p.nb2 = 1; p.alpha = 1.f; p.beta = 0.f;
    p.tiles_m = 3; p.tiles_n = N / 96;
    const int tiles = p.tiles_m * p.tiles_n, s = 252 / tiles;
    const int kw = ((T + s * 8 - 1) / (s * 8) + 7) / 8 * 8;
    p.k_per_split = kw * 8; p.splits = (T + p.k_per_split - 1) / p.k_per_split;
    const int nblk = tiles * p.splits;
    printf("-- 288 x %d, K=%d: %d blocks, %d k per wave (%d groups)\n", N, T, nblk, kw, kw / 8);
    run<0>("full kernel", p, nblk);
    run<1>("no global loads in loop", p, nblk);
    run<2>("no LDS parking (ds_write)", p, nblk);
    run<3>("no global loads, no parking", p, nblk);
    run<4>("operands read once (no ds_read in loop)", p, nblk);
    run<7>("MFMA only", p, nblk);
    run_dma<0>("DMA kernel", p, nblk);
    run_dma<1>("DMA kernel, no DMA/waits in loop", p, nblk);
    hipFree(x); hipFree(g); hipFree(ws);
  }
  return 0;
}
