// Last-axis LayerNorm and the sigmoid-gated GELU of the CLIP blocks (gfx950, fp32, HBM-bound).
//
//   llm/clip/model.py:66-80   CLIPLayerNorm.forward: mean / var over the LAST axis, then
//                             (x - mean) / sqrt(var + eps) * scale + shift      (9 generic nodes)
//   llm/clip/model.py:92-95   MLP: x * sigmoid(1.702 * x)                        (3 generic nodes)
// (The reference's own nn.LayerNorm normalises over the LEADING axes: that one is `colnorm` in fused.hip.)
//
// One wave64 per row, the row lives in registers (cols <= 2048, 16 B per lane loads), wave-shuffle
// reductions, no LDS in forward; backward leaves per-workgroup partial sums of dscale / dshift that a
// fixed-order column reduction combines (deterministic).  Algorithmic bytes: fwd 8 B, bwd 16 B per
// element (+ 4 B when the gradient the input already holds is folded in).
#include "common.h"

static inline int rn_wave_grid(int64_t rows) {
  int64_t g = (rows + 3) / 4;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

template <int VPL>
__global__ void rowln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                 const float* __restrict__ b, float* __restrict__ y, float* __restrict__ mean,
                                 float* __restrict__ rstd, int64_t rows, int cols, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int n4 = cols >> 2;
  float4 wv[VPL], bv[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int idx = lane + 64 * i;
    wv[i] = idx < n4 ? reinterpret_cast<const float4*>(w)[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    bv[i] = idx < n4 ? reinterpret_cast<const float4*>(b)[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int64_t row = wave; row < rows; row += nwaves) {
    const float4* xr = reinterpret_cast<const float4*>(x + row * cols);
    float4 v[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int idx = lane + 64 * i;
      v[i] = idx < n4 ? xr[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mu = wave_sum(s) / (float)cols;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int idx = lane + 64 * i;
      if (idx < n4) {
        v[i].x -= mu; v[i].y -= mu; v[i].z -= mu; v[i].w -= mu;
        ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
      }
    }
    const float sd = sqrtf(wave_sum(ss) / (float)cols + eps);      // divide by sqrt, as the reference does
    if (lane == 0) { mean[row] = mu; rstd[row] = 1.f / sd; }
    float4* yr = reinterpret_cast<float4*>(y + row * cols);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int idx = lane + 64 * i;
      if (idx < n4) {
        float4 t;
        t.x = v[i].x / sd * wv[i].x + bv[i].x; t.y = v[i].y / sd * wv[i].y + bv[i].y;
        t.z = v[i].z / sd * wv[i].z + bv[i].z; t.w = v[i].w / sd * wv[i].w + bv[i].w;
        yr[idx] = t;
      }
    }
  }
}

// dz = dy * w;  dx = rstd * (dz - mean(dz) - xhat * mean(dz * xhat));  dw += dy * xhat;  db += dy
template <int VPL>
__global__ void rowln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                 const float* __restrict__ mean, const float* __restrict__ rstd,
                                 const float* __restrict__ dy, const float* __restrict__ res,
                                 float* __restrict__ dx, float* __restrict__ part, int64_t rows, int cols) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [4 waves][2][cols]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t wave = blockIdx.x * 4ll + wid, nwaves = (int64_t)gridDim.x * 4;
  const int n4 = cols >> 2;
  float4 wv[VPL], aw[VPL], ab[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int idx = lane + 64 * i;
    wv[i] = idx < n4 ? reinterpret_cast<const float4*>(w)[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    aw[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    ab[i] = aw[i];
  }
  for (int64_t row = wave; row < rows; row += nwaves) {
    const float4* xr = reinterpret_cast<const float4*>(x + row * cols);
    const float4* gr = reinterpret_cast<const float4*>(dy + row * cols);
    const float mu = mean[row], rs = rstd[row];
    float4 xh[VPL], dz[VPL];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int idx = lane + 64 * i;
      float4 xv = make_float4(mu, mu, mu, mu), gv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < n4) { xv = xr[idx]; gv = gr[idx]; }
      xh[i].x = (xv.x - mu) * rs; xh[i].y = (xv.y - mu) * rs; xh[i].z = (xv.z - mu) * rs; xh[i].w = (xv.w - mu) * rs;
      dz[i].x = gv.x * wv[i].x; dz[i].y = gv.y * wv[i].y; dz[i].z = gv.z * wv[i].z; dz[i].w = gv.w * wv[i].w;
      aw[i].x += gv.x * xh[i].x; aw[i].y += gv.y * xh[i].y; aw[i].z += gv.z * xh[i].z; aw[i].w += gv.w * xh[i].w;
      ab[i].x += gv.x; ab[i].y += gv.y; ab[i].z += gv.z; ab[i].w += gv.w;
      s1 += (dz[i].x + dz[i].y) + (dz[i].z + dz[i].w);
      s2 += (dz[i].x * xh[i].x + dz[i].y * xh[i].y) + (dz[i].z * xh[i].z + dz[i].w * xh[i].w);
    }
    s1 = wave_sum(s1) / (float)cols;
    s2 = wave_sum(s2) / (float)cols;
    float4* dr = reinterpret_cast<float4*>(dx + row * cols);
    const float4* rr = reinterpret_cast<const float4*>(res ? res + row * cols : nullptr);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int idx = lane + 64 * i;
      if (idx < n4) {
        float4 t;
        t.x = (dz[i].x - s1 - xh[i].x * s2) * rs; t.y = (dz[i].y - s1 - xh[i].y * s2) * rs;
        t.z = (dz[i].z - s1 - xh[i].z * s2) * rs; t.w = (dz[i].w - s1 - xh[i].w * s2) * rs;
        if (res) { const float4 e = rr[idx]; t.x += e.x; t.y += e.y; t.z += e.z; t.w += e.w; }
        dr[idx] = t;
      }
    }
  }
  if (part) {
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int idx = lane + 64 * i;
      if (idx < n4) {
        reinterpret_cast<float4*>(lds + (wid * 2) * cols)[idx] = aw[i];
        reinterpret_cast<float4*>(lds + (wid * 2 + 1) * cols)[idx] = ab[i];
      }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * cols; c += blockDim.x) {
      const int which = c / cols, cc = c - which * cols;
      const float s = (lds[(0 + which) * cols + cc] + lds[(2 + which) * cols + cc]) +
                      (lds[(4 + which) * cols + cc] + lds[(6 + which) * cols + cc]);
      part[((int64_t)blockIdx.x * 2 + which) * cols + cc] = s;
    }
  }
}

// out_w[c] (+)= sum_b part[b][0][c], out_b[c] (+)= sum_b part[b][1][c]  -- fixed order
__global__ void rowln_reduce_kernel(const float* __restrict__ part, int nb, int cols, float* __restrict__ dw,
                                    float* __restrict__ db, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= 2 * cols) return;
  const int which = c / cols, cc = c - which * cols;
  float s = 0.f;
  for (int b = 0; b < nb; ++b) s += part[((int64_t)b * 2 + which) * cols + cc];
  float* out = which ? db : dw;
  if (out) out[cc] = accumulate ? out[cc] + s : s;
}

// y = x * sigmoid(a * x):  dy/dx = s * (1 + a * x * (1 - s)),  s = sigmoid(a * x)
__global__ void gated_sigmoid_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float a, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, n4 = n >> 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    float4 r;
    r.x = v.x / (1.f + expf(-a * v.x)); r.y = v.y / (1.f + expf(-a * v.y));
    r.z = v.z / (1.f + expf(-a * v.z)); r.w = v.w / (1.f + expf(-a * v.w));
    reinterpret_cast<float4*>(y)[i] = r;
  }
  for (int64_t i = n4 * 4 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride)
    y[i] = x[i] / (1.f + expf(-a * x[i]));
}

__device__ __forceinline__ float gs_grad(float x, float a) {
  const float s = 1.f / (1.f + expf(-a * x));
  return s * (1.f + a * x * (1.f - s));
}

__global__ void gated_sigmoid_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                         float* __restrict__ dx, float a, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, n4 = n >> 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const float4 g = reinterpret_cast<const float4*>(dy)[i];
    float4 r;
    r.x = g.x * gs_grad(v.x, a); r.y = g.y * gs_grad(v.y, a); r.z = g.z * gs_grad(v.z, a); r.w = g.w * gs_grad(v.w, a);
    reinterpret_cast<float4*>(dx)[i] = r;
  }
  for (int64_t i = n4 * 4 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride)
    dx[i] = dy[i] * gs_grad(x[i], a);
}

extern "C" {

int64_t pdn_layernorm_bwd_workspace_bytes(int64_t rows, int cols) {
  int64_t nb = (rows + 15) / 16; if (nb > 1024) nb = 1024; if (nb < 1) nb = 1;
  return nb * 2 * (int64_t)cols * 4;
}

/* y = (x - mean) / sqrt(var + eps) * w + b over the LAST axis; mean, rstd (rows,) are saved for backward */
int pdn_layernorm_fwd_f32(const float* x, const float* w, const float* b, float* y, float* mean, float* rstd,
                          int64_t rows, int cols, float eps, void* stream) {
  if (rows == 0) return PDN_OK;
  PDN_CHECK_ARG(x && w && b && y && mean && rstd, "pdn_layernorm_fwd_f32: null operand");
  PDN_CHECK_ARG(cols > 0 && cols % 4 == 0 && cols <= 2048, "pdn_layernorm_fwd_f32: cols=%d must be a multiple of 4 and <= 2048", cols);
  PDN_CHECK_ARG((((uintptr_t)x | (uintptr_t)w | (uintptr_t)b | (uintptr_t)y) & 15) == 0, "pdn_layernorm_fwd_f32: 16B alignment");
  hipStream_t st = (hipStream_t)stream;
  const int g = rn_wave_grid(rows), vpl = (cols / 4 + 63) / 64;
#define LF(V) hipLaunchKernelGGL((rowln_fwd_kernel<V>), dim3(g), dim3(256), 0, st, x, w, b, y, mean, rstd, rows, cols, eps)
  switch (vpl) { case 1: LF(1); break; case 2: LF(2); break; case 3: LF(3); break; case 4: LF(4); break;
                 case 5: LF(5); break; case 6: LF(6); break; case 7: LF(7); break; default: LF(8); }
#undef LF
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

/* dx (+ dx_residual), dw (+)= sum dy * xhat, db (+)= sum dy (accumulate selects += vs =; dw / db nullable) */
int pdn_layernorm_bwd_f32(const float* x, const float* w, const float* mean, const float* rstd, const float* dy,
                          const float* dx_residual, float* dx, float* dw, float* db, int accumulate, int64_t rows,
                          int cols, void* workspace, int64_t workspace_bytes, void* stream) {
  if (rows == 0) return PDN_OK;
  PDN_CHECK_ARG(x && w && mean && rstd && dy && dx, "pdn_layernorm_bwd_f32: null operand");
  PDN_CHECK_ARG(cols > 0 && cols % 4 == 0 && cols <= 2048, "pdn_layernorm_bwd_f32: cols=%d must be a multiple of 4 and <= 2048", cols);
  hipStream_t st = (hipStream_t)stream;
  int64_t nb = (rows + 15) / 16; if (nb > 1024) nb = 1024; if (nb < 1) nb = 1;
  float* part = nullptr;
  if (dw || db) {
    if (!workspace || workspace_bytes < nb * 2 * (int64_t)cols * 4) {
      pdn_set_error("pdn_layernorm_bwd_f32: workspace too small");
      return PDN_EWORKSPACE;
    }
    part = (float*)workspace;
  }
  const int vpl = (cols / 4 + 63) / 64;
  const size_t shm = (size_t)8 * cols * sizeof(float);
#define LB(V) hipLaunchKernelGGL((rowln_bwd_kernel<V>), dim3((unsigned)nb), dim3(256), shm, st, x, w, mean, rstd, dy, dx_residual, dx, part, rows, cols)
  switch (vpl) { case 1: LB(1); break; case 2: LB(2); break; case 3: LB(3); break; case 4: LB(4); break;
                 case 5: LB(5); break; case 6: LB(6); break; case 7: LB(7); break; default: LB(8); }
#undef LB
  PDN_LAUNCH_CHECK();
  if (part) {
    hipLaunchKernelGGL(rowln_reduce_kernel, dim3((2 * cols + 255) / 256), dim3(256), 0, st, part, (int)nb, cols, dw, db,
                       accumulate);
    PDN_LAUNCH_CHECK();
  }
  return PDN_OK;
}

/* y = x * sigmoid(alpha * x) and its gradient (alpha = 1.702: CLIP's quick-GELU; alpha = 1: SiLU) */
int pdn_gated_sigmoid_fwd_f32(const float* x, float* y, float alpha, int64_t n, void* stream) {
  if (n == 0) return PDN_OK;
  PDN_CHECK_ARG(x && y && ((((uintptr_t)x | (uintptr_t)y) & 15) == 0), "pdn_gated_sigmoid_fwd_f32: bad operand");
  int64_t g = (n / 4 + 255) / 256; if (g > 8192) g = 8192; if (g < 1) g = 1;
  hipLaunchKernelGGL(gated_sigmoid_fwd_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, y, alpha, n);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

int pdn_gated_sigmoid_bwd_f32(const float* x, const float* dy, float* dx, float alpha, int64_t n, void* stream) {
  if (n == 0) return PDN_OK;
  PDN_CHECK_ARG(x && dy && dx && ((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0),
                "pdn_gated_sigmoid_bwd_f32: bad operand");
  int64_t g = (n / 4 + 255) / 256; if (g > 8192) g = 8192; if (g < 1) g = 1;
  hipLaunchKernelGGL(gated_sigmoid_bwd_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, dy, dx, alpha, n);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

}  // extern "C"
