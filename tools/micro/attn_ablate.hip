// Ablation timing of attention_fwd_kernel: which phase costs the time?
// Build: hipcc -O3 --offload-arch=gfx950 -I../../pydynet_amd/csrc attn_ablate.hip ../../pydynet_amd/csrc/abi.hip -o attn_ablate.bin
#include "../../pydynet_amd/csrc/attention.hip"
#include <stdio.h>

template <int AB>
static void run(const char* name, const float* q, const float* k, const float* v, float* o, float* lse, int B, int H, int L) {
  const size_t shm = (size_t)att_fwd_lds_bytes(L, 48);
  hipFuncSetAttribute((const void*)attention_fwd_kernel<48, AB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) {
    hipEventRecord(e0);
    for (int it = 0; it < 10; ++it)
      hipLaunchKernelGGL((attention_fwd_kernel<48, AB>), dim3(B * H), dim3(512), shm, 0, q, k, v, o, lse, H, L,
                         (int64_t)H * 48, (int64_t)L * H * 48, (int64_t)H * 48, (int64_t)L * H * 48, sqrtf(48.f), 1, (const float*)nullptr, (const float*)nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-52s %7.1f us\n", name, ms * 100);
}

int main() {
  const int B = 256, H = 6, L = 256;
  const size_t n = (size_t)B * L * H * 48;
  float *q, *k, *v, *o, *lse;
  hipMalloc(&q, n * 4); hipMalloc(&k, n * 4); hipMalloc(&v, n * 4); hipMalloc(&o, n * 4); hipMalloc(&lse, (size_t)B * H * L * 4);
  hipMemset(q, 0, n * 4); hipMemset(k, 0, n * 4); hipMemset(v, 0, n * 4);
  run<0>("full kernel", q, k, v, o, lse, B, H, L);
  run<1>("no K/V staging", q, k, v, o, lse, B, H, L);
  run<2>("no S^T MFMAs", q, k, v, o, lse, B, H, L);
  run<4>("no softmax arithmetic", q, k, v, o, lse, B, H, L);
  run<8>("no PV MFMAs", q, k, v, o, lse, B, H, L);
  run<16>("no output store", q, k, v, o, lse, B, H, L);
  run<1 | 16>("no staging, no store", q, k, v, o, lse, B, H, L);
  run<2 | 4 | 8>("staging + store only", q, k, v, o, lse, B, H, L);
  run<1 | 4 | 16>("MFMAs only (S + PV)", q, k, v, o, lse, B, H, L);
  run<1 | 2 | 8 | 16>("softmax only", q, k, v, o, lse, B, H, L);
  return 0;
}
