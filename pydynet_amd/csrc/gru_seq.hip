// Persistent GRU sequence kernels for gfx950 (hidden size 32): the whole time loop of
// pydynet/nn/modules/rnn.py:640-708 (GRU.forward -> cell_forward, a Python loop of GRUCell.forward,
// :537-544) for one layer / one direction runs inside ONE launch, forward and backward.
//
//   [z, r] = sigmoid(x Wx1 + h Wh1 + b1);  n = tanh(x Wx2 + (r*h) Wh2 + b2);  h' = (1-z) h + z n
//
// The recurrence couples only the 32 hidden units of ONE sequence, never two sequences, so the batch is
// cut into tiles of 32 sequences and one wave64 owns a tile for all T steps -- no inter-workgroup
// synchronisation, no launch per step (the per-step path costs 4 launches forward, 5 backward: 360 for
// T = 40, the whole step was launch-bound at ~4 ms).  The input projections x Wx (+ b) of all steps are
// hoisted into two GEMMs by the caller and arrive as g1x (T, B, 64), g2x (T, B, 32).
//
// MFMA formulation (v_mfma_f32_32x32x2_f32, exact fp32): everything is kept TRANSPOSED,
//   G^T[out][b] += Wh^T[out][k] * h^T[k][b]
// so that an accumulator holds the sequence index in its LANE and 16 of the 32 hidden units in its
// REGISTERS.  The gate algebra is then elementwise on registers, and the new h^T accumulator is fed to the
// next step's MFMA as the B operand AS IT IS: inside one MFMA the two half-waves may contract over any two
// k as long as A and B agree, and register r of an accumulator holds units row(r, half) -- so MFMA r
// contracts over k = row(r, 0) / row(r, 1) and its A operand is the matching pair of weight rows, which
// are loop-invariant and live in 48 registers.  h never touches LDS or HBM inside the loop; per step the
// kernel reads the hoisted projections and writes z, r, r*h, n, h (saved for backward) as float4 per lane.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GH 32

__device__ __forceinline__ int g_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

__device__ __forceinline__ float g_sigmoid(float x) {           // tensor.py:999-1003 (overflow-safe piecewise)
  return x > 0.f ? 1.f / (1.f + expf(-x)) : 1.f - 1.f / (1.f + expf(x));
}
__device__ __forceinline__ float g_tanh(float x) {              // tensor.py:1012-1016
  return x > 0.f ? 2.f / (1.f + expf(-2.f * x)) - 1.f : 1.f - 2.f / (1.f + expf(2.f * x));
}

// accumulator-layout load / store of a (B, C) row block: lane = sequence b, register r = unit col0 + row(r, half)
__device__ __forceinline__ void g_load(f32x16& a, const float* __restrict__ base, int64_t row_stride, int b, int col0,
                                       int half, bool ok) {
  // (unconditional: `b` is clamped by the caller for lanes past the batch, whose values are never stored -- a load
  //  under `if (ok)` is followed by the copy that merges it with the zero of the untaken path, and that copy WAITS for
  //  the load: the one-step-ahead prefetch of the recurrence would be no prefetch)
  (void)ok;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 v = *reinterpret_cast<const float4*>(base + (int64_t)b * row_stride + col0 + 8 * q + 4 * half);
    a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
  }
}
__device__ __forceinline__ void g_store(const f32x16& a, float* __restrict__ base, int64_t row_stride, int b, int col0,
                                        int half, bool ok) {
  if (!ok) return;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    *reinterpret_cast<float4*>(base + (int64_t)b * row_stride + col0 + 8 * q + 4 * half) =
        make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
}

// out^T += W-operand x src^T, the B operand being the accumulator `src` itself (16 MFMAs)
__device__ __forceinline__ void g_mma(f32x16& out, const float (&w)[16], const f32x16& src) {
#pragma unroll
  for (int r = 0; r < 16; ++r) out = __builtin_amdgcn_mfma_f32_32x32x2f32(w[r], src[r], out, 0, 0, 0);
}

// forward: h0 (B, 32); wh1 (32, 64), wh2 (32, 32) row-major (in, out); outputs (T, B, 32) each
__global__ __launch_bounds__(256, 1) void gru_seq_fwd_kernel(const float* __restrict__ g1x, const float* __restrict__ g2x,
                                                          const float* __restrict__ h0, const float* __restrict__ wh1,
                                                          const float* __restrict__ wh2, float* __restrict__ Z,
                                                          float* __restrict__ R, float* __restrict__ RH,
                                                          float* __restrict__ N, float* __restrict__ OUT, int T,
                                                          int B) {
  const int lane = threadIdx.x & 63, li = lane & 31, half = lane >> 5;
  const int tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int b = tile * 32 + li;
  if (tile * 32 >= B) return;                                   // wave-uniform
  const bool ok = b < B;
  const int bl = ok ? b : B - 1;                                // row the loads of a lane past the batch read
  // A operands: lane (i = li, half) of MFMA r supplies W^T[out = i][k = row(r, half)] = W[k][out]
  float wz[16], wr[16], wn[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int k = g_row(r, half);
    wz[r] = wh1[k * 64 + li];
    wr[r] = wh1[k * 64 + 32 + li];
    wn[r] = wh2[k * 32 + li];
  }
  f32x16 h;
  g_load(h, h0, GH, bl, 0, half, ok);
  // the hoisted projections of step t+1 do not depend on step t: their loads are issued one step ahead, so
  // the only latency left on the recurrence's critical path is MFMA -> gate math -> MFMA
  f32x16 nz, nr, nn;
  g_load(nz, g1x, 64, bl, 0, half, ok);
  g_load(nr, g1x, 64, bl, 32, half, ok);
  g_load(nn, g2x, 32, bl, 0, half, ok);
  for (int t = 0; t < T; ++t) {
    const int64_t off = (int64_t)t * B;
    f32x16 az = nz, ar = nr, an = nn;
    {
      const int64_t on = (int64_t)min(t + 1, T - 1) * B;        // (the last step re-reads itself: unconditional loads)
      g_load(nz, g1x + on * 64, 64, bl, 0, half, ok);
      g_load(nr, g1x + on * 64, 64, bl, 32, half, ok);
      g_load(nn, g2x + on * 32, 32, bl, 0, half, ok);
    }
    g_mma(az, wz, h);
    g_mma(ar, wr, h);
    f32x16 rh;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      az[r] = g_sigmoid(az[r]);
      ar[r] = g_sigmoid(ar[r]);
      rh[r] = ar[r] * h[r];
    }
    g_mma(an, wn, rh);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      an[r] = g_tanh(an[r]);
      h[r] = (1.f - az[r]) * h[r] + az[r] * an[r];
    }
    g_store(az, Z + off * 32, 32, b, 0, half, ok);
    g_store(ar, R + off * 32, 32, b, 0, half, ok);
    g_store(rh, RH + off * 32, 32, b, 0, half, ok);
    g_store(an, N + off * 32, 32, b, 0, half, ok);
    g_store(h, OUT + off * 32, 32, b, 0, half, ok);
  }
}

// backward: walks t = T-1 .. 0 with dh in registers.
//   dh += g[t];  dn = dh z;  dG2 = dn (1 - n^2);  dz = dh (n - hprev);  dG1[:, :32] = dz z (1 - z)
//   drh = dG2 Wh2^T;  dr = drh hprev;  dG1[:, 32:] = dr r (1 - r)
//   dh_prev = dh (1 - z) + drh r + dG1 Wh1^T
// A operands: X^T[k][b] += W[k][out] dG^T[out][b]: lane (i = k, half) of MFMA r supplies W[k = i][out = row(r, half)]
__global__ __launch_bounds__(256, 1) void gru_seq_bwd_kernel(const float* __restrict__ G, const float* __restrict__ Z,
                                                          const float* __restrict__ R, const float* __restrict__ N,
                                                          const float* __restrict__ OUT, const float* __restrict__ h0,
                                                          const float* __restrict__ wh1, const float* __restrict__ wh2,
                                                          float* __restrict__ dG1, float* __restrict__ dG2,
                                                          float* __restrict__ dh0, int T, int B) {
  const int lane = threadIdx.x & 63, li = lane & 31, half = lane >> 5;
  const int tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int b = tile * 32 + li;
  if (tile * 32 >= B) return;
  const bool ok = b < B;
  float wz[16], wr[16], wn[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int o = g_row(r, half);
    wz[r] = wh1[li * 64 + o];
    wr[r] = wh1[li * 64 + 32 + o];
    wn[r] = wh2[li * 32 + o];
  }
  f32x16 dh;
#pragma unroll
  for (int r = 0; r < 16; ++r) dh[r] = 0.f;
  f32x16 pg, pz, pr, pn, ph;                                    // operands of the step about to run
  const int bl = ok ? b : B - 1;                                // row the loads of a lane past the batch read
  auto fetch = [&](int t) {
    const int64_t o2 = (int64_t)t * B;
    g_load(pg, G + o2 * 32, 32, bl, 0, half, ok);
    g_load(pz, Z + o2 * 32, 32, bl, 0, half, ok);
    g_load(pr, R + o2 * 32, 32, bl, 0, half, ok);
    g_load(pn, N + o2 * 32, 32, bl, 0, half, ok);
    // h of the step before: one address select instead of two loads under a branch (GH = row stride of h0)
    const float* hsrc = t > 0 ? OUT + (o2 - B) * 32 : h0;
    g_load(ph, hsrc, t > 0 ? 32 : GH, bl, 0, half, ok);
  };
  fetch(T - 1);
  for (int t = T - 1; t >= 0; --t) {
    const int64_t off = (int64_t)t * B;
    f32x16 g = pg, z = pz, rr = pr, n = pn, hp = ph;
    fetch(t > 0 ? t - 1 : 0);                                   // saved tensors of the next step: no dependence (step 0
                                                                // re-reads itself: unconditional loads)
    f32x16 dg2, dgz, dgr, drh, dhp;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float d = dh[r] + g[r];
      dg2[r] = (1.f - n[r] * n[r]) * (d * z[r]);
      dgz[r] = z[r] * (1.f - z[r]) * (d * (n[r] - hp[r]));
      dhp[r] = d * (1.f - z[r]);
      drh[r] = 0.f;
    }
    g_mma(drh, wn, dg2);                                        // drh^T = Wh2 dG2^T
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      dgr[r] = rr[r] * (1.f - rr[r]) * (drh[r] * hp[r]);
      dhp[r] += drh[r] * rr[r];
    }
    g_mma(dhp, wz, dgz);                                        // + Wh1[:, :32] dGz^T
    g_mma(dhp, wr, dgr);                                        // + Wh1[:, 32:] dGr^T
    g_store(dgz, dG1 + off * 64, 64, b, 0, half, ok);
    g_store(dgr, dG1 + off * 64, 64, b, 32, half, ok);
    g_store(dg2, dG2 + off * 32, 32, b, 0, half, ok);
    dh = dhp;
  }
  if (dh0) g_store(dh, dh0, GH, b, 0, half, ok);
}

extern "C" {

int pdn_gru_seq_supported(int hidden) { return hidden == GH ? 1 : 0; }

/* g1x (T, B, 2H) = x Wx1 (+ b1), g2x (T, B, H) = x Wx2 (+ b2) for all steps; h0 (B, H); wh1 (H, 2H), wh2 (H, H)
 * contiguous.  Writes z, r, r*h, n and the hidden states out (T, B, H) each.  H must be 32. */
int pdn_gru_seq_fwd_f32(const float* g1x, const float* g2x, const float* h0, const float* wh1, const float* wh2,
                        float* z, float* r, float* rh, float* n, float* out, int T, int B, int H, void* stream) {
  if (T == 0 || B == 0) return PDN_OK;
  PDN_CHECK_ARG(g1x && g2x && h0 && wh1 && wh2 && z && r && rh && n && out, "pdn_gru_seq_fwd_f32: null operand");
  if (H != GH) {
    pdn_set_error("pdn_gru_seq_fwd_f32: hidden size %d (the persistent kernel serves 32)", H);
    return PDN_EUNSUPPORTED;
  }
  PDN_CHECK_ARG(((((uintptr_t)g1x | (uintptr_t)g2x | (uintptr_t)h0 | (uintptr_t)z | (uintptr_t)r | (uintptr_t)rh |
                   (uintptr_t)n | (uintptr_t)out) & 15) == 0), "pdn_gru_seq_fwd_f32: 16-byte alignment required");
  // one wave per workgroup while that still leaves CUs idle (a wave is latency-bound on its recurrence:
  // it wants a CU's caches and issue slots for itself), four per workgroup for large batches
  const int tiles = (B + 31) / 32, wpb = tiles <= 512 ? 1 : 4;
  hipLaunchKernelGGL(gru_seq_fwd_kernel, dim3((tiles + wpb - 1) / wpb), dim3(64 * wpb), 0, (hipStream_t)stream, g1x, g2x,
                     h0, wh1, wh2, z, r, rh, n, out, T, B);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

/* g (T, B, H) = gradient of the hidden states; writes dG1 (T, B, 2H), dG2 (T, B, H) (the gradients of the
 * pre-activations: the caller forms every weight / input gradient from them with long-K GEMMs) and dh0 (B, H). */
int pdn_gru_seq_bwd_f32(const float* g, const float* z, const float* r, const float* n, const float* out,
                        const float* h0, const float* wh1, const float* wh2, float* dg1, float* dg2, float* dh0, int T,
                        int B, int H, void* stream) {
  if (T == 0 || B == 0) return PDN_OK;
  PDN_CHECK_ARG(g && z && r && n && out && h0 && wh1 && wh2 && dg1 && dg2, "pdn_gru_seq_bwd_f32: null operand");
  if (H != GH) {
    pdn_set_error("pdn_gru_seq_bwd_f32: hidden size %d (the persistent kernel serves 32)", H);
    return PDN_EUNSUPPORTED;
  }
  const int tiles = (B + 31) / 32, wpb = tiles <= 512 ? 1 : 4;
  hipLaunchKernelGGL(gru_seq_bwd_kernel, dim3((tiles + wpb - 1) / wpb), dim3(64 * wpb), 0, (hipStream_t)stream, g, z, r,
                     n, out, h0, wh1, wh2, dg1, dg2, dh0, T, B);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

}  // extern "C"
