// Sustained v_mfma_f32_32x32x2_f32 rate on gfx950 as a function of run length and operand data:
// constant operands vs N(0,1) operands read from LDS (the matrix pipe's power, hence the clock the
// chip sustains, depends on how many bits toggle).  Build: hipcc -O3 --offload-arch=gfx950 mfma_sustained.hip -o mfma_sustained.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int RANDOM>
__global__ __launch_bounds__(512, 1) void k(const float* __restrict__ src, float* out, int iters) {
  __shared__ float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = RANDOM ? src[i] : 1.0f;
  __syncthreads();
  f32x16 acc[9];
  for (int i = 0; i < 9; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const float* lp = lds + (threadIdx.x & 63);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 8; ++rep) {
      float a[3], b[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) { a[i] = lp[64 * (3 * rep + i)]; b[i] = lp[4096 + 64 * (3 * rep + i)]; }
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[3 * i + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[3 * i + j], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 9; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int RANDOM>
void run(const char* name, const float* src, float* out, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<RANDOM>), dim3(256), dim3(512), 0, 0, src, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = 256.0 * 8 * iters * 72.0 * 4096.0;
  printf("%-28s iters=%7d  %9.3f ms  %6.1f TFLOP/s  (%.1f %% of 157.3)\n", name, iters, ms, flops / ms / 1e9, flops / ms / 1e9 / 1.573);
}

int main() {
  float* h = (float*)malloc(8192 * 4);
  srand(1);
  for (int i = 0; i < 8192; ++i) {
    const float u1 = (rand() + 1.f) / (RAND_MAX + 2.f), u2 = (rand() + 1.f) / (RAND_MAX + 2.f);
    h[i] = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
  }
  float *src, *out; hipMalloc(&src, 8192 * 4); hipMalloc(&out, 256 * 512 * 4);
  hipMemcpy(src, h, 8192 * 4, hipMemcpyHostToDevice);
  run<0>("warm-up", src, out, 1000);
  for (int iters : {1000, 10000, 100000}) {
    run<0>("constant operands", src, out, iters);
    run<1>("N(0,1) operands", src, out, iters);
  }
  run<1>("N(0,1) operands", src, out, 400000);
  run<0>("constant operands", src, out, 400000);
  return 0;
}
