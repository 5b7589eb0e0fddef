// Microbenchmark: sustained v_mfma_f32_32x32x2_f32 rate on gfx950 with 1..2 waves per SIMD, with and
// without interleaved VALU / LDS traffic.  Build: hipcc -O3 --offload-arch=gfx950 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int VALU, int LDSR>
__global__ __launch_bounds__(512, 1) void k(float* out, int iters, float a, float b) {
  __shared__ float lds[4096];
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float x = a + threadIdx.x, y = b;
  lds[threadIdx.x] = x;
  __syncthreads();
  const float* lp = lds + (threadIdx.x & 63);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
      float f[NACC];
#pragma unroll
      for (int i = 0; i < NACC; ++i) f[i] = LDSR ? lp[64 * ((i + rep) & 15)] : x;
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[i], y, acc[i], 0, 0, 0);
        if (VALU) { x = x * 1.0001f + 0.5f; }
        if (VALU > 1) { y = y * 0.9999f + 0.25f; }
      }
    }
  }
  float s = x + y;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int VALU, int LDSR>
void run(const char* name, int threads, int blocks) {
  float* out; hipMalloc(&out, 4 * 512 * 4096);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, VALU, LDSR>), dim3(blocks), dim3(threads), 0, 0, out, 10, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, VALU, LDSR>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * (threads / 64) * iters * 4.0 * NACC * 4096.0;
  printf("%-44s blocks=%d threads=%d  %.3f ms  %.1f TFLOP/s\n", name, blocks, threads, ms, flops / ms / 1e9);
  hipFree(out);
}

int main() {
  run<9, 0, 0>("9 acc, 1 wave/SIMD", 256, 256);
  run<9, 0, 0>("9 acc, 2 waves/SIMD", 512, 256);
  run<4, 0, 0>("4 acc, 2 waves/SIMD", 512, 256);
  run<2, 0, 0>("2 acc, 2 waves/SIMD", 512, 256);
  run<1, 0, 0>("1 acc (dependent chain), 2 waves/SIMD", 512, 256);
  run<9, 1, 0>("9 acc + 1 VALU fma per MFMA, 2 waves/SIMD", 512, 256);
  run<9, 2, 0>("9 acc + 2 VALU fma per MFMA, 2 waves/SIMD", 512, 256);
  run<9, 0, 1>("9 acc + 1 ds_read_b32 per MFMA, 2 waves/SIMD", 512, 256);
  run<9, 0, 1>("9 acc + ds_read, 1 wave/SIMD", 256, 256);
  run<9, 0, 0>("9 acc, 2 waves/SIMD, 2 rounds", 512, 512);
  return 0;
}
