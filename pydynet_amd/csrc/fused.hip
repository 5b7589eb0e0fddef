// Fused bandwidth kernels for the training hot path (gfx950).  Each replaces a chain of
// generic tape nodes of the reference with ONE forward and ONE backward pass over HBM:
//
//   softmax            pydynet/nn/functional.py:43-49   (4 nodes; + `/sqrt(hd)` and `+mask`
//                      of llm/llama/model.py:113-117 folded in for attention scores)
//   RMSNorm            pydynet/nn/modules/norm.py:245-248 (6 nodes)
//   SiLU / SwiGLU      functional.py:39-40, llm/llama/model.py:56-58 (4+1 nodes)
//   RoPE               llm/llama/model.py:23-44 (26 nodes)
//   embedding          functional.py:14-20 gather; tensor.py:937-940 scatter-ASSIGN backward
//   cross entropy      functional.py:364-381 (7 nodes, five (N,V) temporaries)
//   relu backward      tensor.py:814-815 with functional.py:31-32
//
// Rows are handled one wave64 per row when they fit in registers (<= 1024 floats: 288 for
// RMSNorm, 256 for attention scores), loads are 16 B per lane, row reductions are wave
// shuffles -- no LDS.  Everything is fp32; reductions are in a fixed order (deterministic).
#include "common.h"

// ======================================================================================
// softmax over the last (contiguous) dim, optional attention prologue:
//   y = softmax(x / divisor + causal_mask)     masked positions contribute exactly 0
// causal: rows are grouped in blocks of `L` query positions; position (r % L) may attend to
// key columns c <= (r % L) + start_pos (mask of llm/llama/model.py:199-203).
// ======================================================================================
template <int VPL>
__global__ void softmax_fwd_wave_kernel(const float* __restrict__ x, float* __restrict__ y,
                                        int64_t rows, int cols, float divisor, int causal_L,
                                        int start_pos) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int n4 = cols >> 2;
  for (int64_t row = wave; row < rows; row += nwaves) {
    const float4* xr = reinterpret_cast<const float4*>(x + row * cols);
    const int limit = causal_L > 0 ? (int)(row % causal_L) + start_pos : cols;  // last valid col
    float4 v[VPL];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int idx = lane + 64 * i;
      v[i] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      if (idx < n4 && 4 * idx <= limit) {
        float4 t = xr[idx];
        t.x = t.x / divisor; t.y = t.y / divisor; t.z = t.z / divisor; t.w = t.w / divisor;
        const int c = 4 * idx;
        if (c + 1 > limit) t.y = -INFINITY;
        if (c + 2 > limit) t.z = -INFINITY;
        if (c + 3 > limit) t.w = -INFINITY;
        v[i] = t;
      }
      m = fmaxf(m, fmaxf(fmaxf(v[i].x, v[i].y), fmaxf(v[i].z, v[i].w)));
    }
    m = wave_max(m);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      v[i].x = expf(v[i].x - m); v[i].y = expf(v[i].y - m);
      v[i].z = expf(v[i].z - m); v[i].w = expf(v[i].w - m);
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    s = wave_sum(s);
    float4* yr = reinterpret_cast<float4*>(y + row * cols);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int idx = lane + 64 * i;
      if (idx < n4) {
        float4 t = v[i];
        t.x /= s; t.y /= s; t.z /= s; t.w /= s;
        yr[idx] = t;
      }
    }
  }
}

// dx = (dy - sum(dy*y)) * y / divisor
template <int VPL>
__global__ void softmax_bwd_wave_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                        float* __restrict__ dx, int64_t rows, int cols,
                                        float divisor) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int n4 = cols >> 2;
  for (int64_t row = wave; row < rows; row += nwaves) {
    const float4* yr = reinterpret_cast<const float4*>(y + row * cols);
    const float4* gr = reinterpret_cast<const float4*>(dy + row * cols);
    float4 p[VPL], g[VPL];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int idx = lane + 64 * i;
      p[i] = make_float4(0.f, 0.f, 0.f, 0.f); g[i] = p[i];
      if (idx < n4) { p[i] = yr[idx]; g[i] = gr[idx]; }
      dot += (p[i].x * g[i].x + p[i].y * g[i].y) + (p[i].z * g[i].z + p[i].w * g[i].w);
    }
    dot = wave_sum(dot);
    float4* dr = reinterpret_cast<float4*>(dx + row * cols);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int idx = lane + 64 * i;
      if (idx < n4) {
        float4 t;
        t.x = (g[i].x - dot) * p[i].x / divisor; t.y = (g[i].y - dot) * p[i].y / divisor;
        t.z = (g[i].z - dot) * p[i].z / divisor; t.w = (g[i].w - dot) * p[i].w / divisor;
        dr[idx] = t;
      }
    }
  }
}

// Any row length: one workgroup per row, three passes (second and third hit L2).
__global__ void softmax_fwd_block_kernel(const float* __restrict__ x, float* __restrict__ y,
                                         int64_t rows, int cols, float divisor, int causal_L,
                                         int start_pos) {
  __shared__ float red[16];
  for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
    const float* xr = x + row * cols;
    float* yr = y + row * cols;
    const int limit = causal_L > 0 ? (int)(row % causal_L) + start_pos : cols;
    float m = -INFINITY;
    for (int c = threadIdx.x; c < cols; c += blockDim.x)
      if (c <= limit) m = fmaxf(m, xr[c] / divisor);
    m = block_max(m, red);
    float s = 0.f;
    for (int c = threadIdx.x; c < cols; c += blockDim.x)
      if (c <= limit) s += expf(xr[c] / divisor - m);
    s = block_sum(s, red);
    for (int c = threadIdx.x; c < cols; c += blockDim.x)
      yr[c] = (c <= limit) ? expf(xr[c] / divisor - m) / s : 0.f;
    __syncthreads();
  }
}
__global__ void softmax_bwd_block_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                         float* __restrict__ dx, int64_t rows, int cols,
                                         float divisor) {
  __shared__ float red[16];
  for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
    const float* yr = y + row * cols; const float* gr = dy + row * cols;
    float dot = 0.f;
    for (int c = threadIdx.x; c < cols; c += blockDim.x) dot += yr[c] * gr[c];
    dot = block_sum(dot, red);
    for (int c = threadIdx.x; c < cols; c += blockDim.x)
      dx[row * cols + c] = (gr[c] - dot) * yr[c] / divisor;
    __syncthreads();
  }
}

static inline int wave_grid(int64_t rows) {
  int64_t b = (rows + 3) / 4;
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (int)b;
}

extern "C" int pdn_softmax_fwd_f32(const float* x, float* y, int64_t rows, int cols,
                                   float divisor, int causal_L, int start_pos, void* stream) {
  if (rows == 0 || cols == 0) return PDN_OK;
  PDN_CHECK_ARG(x && y && rows > 0 && cols > 0, "pdn_softmax_fwd_f32: bad arguments");
  PDN_CHECK_ARG(divisor != 0.f, "pdn_softmax_fwd_f32: divisor 0");
  hipStream_t st = (hipStream_t)stream;
  const bool vec = (cols % 4 == 0) && cols <= 1024 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0;
  if (vec) {
    const int g = wave_grid(rows);
    const int vpl = (cols / 4 + 63) / 64;
#define SFW(V) hipLaunchKernelGGL((softmax_fwd_wave_kernel<V>), dim3(g), dim3(256), 0, st, x, y, rows, cols, divisor, causal_L, start_pos)
    if (vpl == 1) SFW(1); else if (vpl == 2) SFW(2); else if (vpl == 3) SFW(3); else SFW(4);
  } else {
    const int g = (int)(rows < 4096 ? rows : 4096);
    hipLaunchKernelGGL(softmax_fwd_block_kernel, dim3(g), dim3(256), 0, st, x, y, rows, cols,
                       divisor, causal_L, start_pos);
  }
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

extern "C" int pdn_softmax_bwd_f32(const float* y, const float* dy, float* dx, int64_t rows,
                                   int cols, float divisor, void* stream) {
  if (rows == 0 || cols == 0) return PDN_OK;
  PDN_CHECK_ARG(y && dy && dx && rows > 0 && cols > 0, "pdn_softmax_bwd_f32: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const bool vec = (cols % 4 == 0) && cols <= 1024 &&
                   (((uintptr_t)y | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0;
  if (vec) {
    const int g = wave_grid(rows);
    const int vpl = (cols / 4 + 63) / 64;
#define SBW(V) hipLaunchKernelGGL((softmax_bwd_wave_kernel<V>), dim3(g), dim3(256), 0, st, y, dy, dx, rows, cols, divisor)
    if (vpl == 1) SBW(1); else if (vpl == 2) SBW(2); else if (vpl == 3) SBW(3); else SBW(4);
  } else {
    const int g = (int)(rows < 4096 ? rows : 4096);
    hipLaunchKernelGGL(softmax_bwd_block_kernel, dim3(g), dim3(256), 0, st, y, dy, dx, rows, cols, divisor);
  }
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

// ======================================================================================
// RMSNorm:  rms = sqrt(mean(x*x) + eps);  y = x / rms * w        (norm.py:245-248)
// backward: dz = dy*w;  dx = (dz - z*mean(z*dz)) / rms;  dw = sum_rows dy*z,  z = x/rms
// ======================================================================================
template <int VPL>
__global__ void rmsnorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                   float* __restrict__ y, float* __restrict__ rms, int64_t rows,
                                   int cols, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int n4 = cols >> 2;
  float4 wv[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int idx = lane + 64 * i;
    wv[i] = idx < n4 ? reinterpret_cast<const float4*>(w)[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int64_t row = wave; row < rows; row += nwaves) {
    const float4* xr = reinterpret_cast<const float4*>(x + row * cols);
    float4 v[VPL];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int idx = lane + 64 * i;
      v[i] = idx < n4 ? xr[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
      ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    ss = wave_sum(ss);
    const float r = sqrtf(ss / (float)cols + eps);
    if (lane == 0 && rms) rms[row] = r;
    float4* yr = reinterpret_cast<float4*>(y + row * cols);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int idx = lane + 64 * i;
      if (idx < n4) {
        float4 t;
        t.x = v[i].x / r * wv[i].x; t.y = v[i].y / r * wv[i].y;
        t.z = v[i].z / r * wv[i].z; t.w = v[i].w / r * wv[i].w;
        yr[idx] = t;
      }
    }
  }
}

// Each workgroup (4 waves) walks its rows and leaves one dw partial row in `dw_part`.
template <int VPL>
__global__ void rmsnorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                   const float* __restrict__ rms, const float* __restrict__ dy,
                                   const float* __restrict__ res, float* __restrict__ dx,
                                   float* __restrict__ dw_part, int64_t rows, int cols) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [4][cols]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t wave = blockIdx.x * 4ll + wid;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const int n4 = cols >> 2;
  float4 wv[VPL], acc[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int idx = lane + 64 * i;
    wv[i] = idx < n4 ? reinterpret_cast<const float4*>(w)[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int64_t row = wave; row < rows; row += nwaves) {
    const float4* xr = reinterpret_cast<const float4*>(x + row * cols);
    const float4* gr = reinterpret_cast<const float4*>(dy + row * cols);
    const float rinv = 1.0f / rms[row];
    float4 z[VPL], dz[VPL];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int idx = lane + 64 * i;
      float4 xv = make_float4(0.f, 0.f, 0.f, 0.f), gv = xv;
      if (idx < n4) { xv = xr[idx]; gv = gr[idx]; }
      z[i].x = xv.x * rinv; z[i].y = xv.y * rinv; z[i].z = xv.z * rinv; z[i].w = xv.w * rinv;
      dz[i].x = gv.x * wv[i].x; dz[i].y = gv.y * wv[i].y; dz[i].z = gv.z * wv[i].z; dz[i].w = gv.w * wv[i].w;
      acc[i].x += gv.x * z[i].x; acc[i].y += gv.y * z[i].y; acc[i].z += gv.z * z[i].z; acc[i].w += gv.w * z[i].w;
      dot += (z[i].x * dz[i].x + z[i].y * dz[i].y) + (z[i].z * dz[i].z + z[i].w * dz[i].w);
    }
    dot = wave_sum(dot) / (float)cols;
    float4* dr = reinterpret_cast<float4*>(dx + row * cols);
    const float4* rr = reinterpret_cast<const float4*>(res ? res + row * cols : nullptr);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int idx = lane + 64 * i;
      if (idx < n4) {
        float4 t;
        t.x = (dz[i].x - z[i].x * dot) * rinv; t.y = (dz[i].y - z[i].y * dot) * rinv;
        t.z = (dz[i].z - z[i].z * dot) * rinv; t.w = (dz[i].w - z[i].w * dot) * rinv;
        if (res) { const float4 e = rr[idx]; t.x += e.x; t.y += e.y; t.z += e.z; t.w += e.w; }
        dr[idx] = t;
      }
    }
  }
  if (dw_part) {
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int idx = lane + 64 * i;
      if (idx < n4) reinterpret_cast<float4*>(lds + wid * cols)[idx] = acc[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < cols; c += blockDim.x)
      dw_part[(int64_t)blockIdx.x * cols + c] =
          (lds[c] + lds[cols + c]) + (lds[2 * cols + c] + lds[3 * cols + c]);
  }
}

// out[c] (+)= sum_b part[b][c]   -- fixed order
// block = 16 columns x 64 row lanes (1024 threads): each lane sums every 64th partial row (four
// independent loads in flight), then the 64 lane sums are combined through LDS in a fixed order.
#define COLSUM_COLS 16
__global__ __launch_bounds__(1024) void colsum_partials_kernel(const float* __restrict__ part, int nb, int cols,
                                                               float* __restrict__ out, int accumulate) {
  __shared__ float red[64][COLSUM_COLS];
  const int tx = threadIdx.x & (COLSUM_COLS - 1), ty = threadIdx.x / COLSUM_COLS;
  const int c = blockIdx.x * COLSUM_COLS + tx;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < cols) {
    int b = ty;
    for (; b + 192 < nb; b += 256) {
      s0 += part[(int64_t)b * cols + c];
      s1 += part[(int64_t)(b + 64) * cols + c];
      s2 += part[(int64_t)(b + 128) * cols + c];
      s3 += part[(int64_t)(b + 192) * cols + c];
    }
    for (; b < nb; b += 64) s0 += part[(int64_t)b * cols + c];
  }
  red[ty][tx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ty == 0 && c < cols) {
    float t = red[0][tx];
#pragma unroll
    for (int k = 1; k < 64; ++k) t += red[k][tx];
    out[c] = accumulate ? out[c] + t : t;
  }
}

extern "C" int64_t pdn_rmsnorm_bwd_workspace_bytes(int64_t rows, int cols) {
  int64_t nb = (rows + 15) / 16; if (nb > 1024) nb = 1024; if (nb < 1) nb = 1;
  return nb * (int64_t)cols * 4;
}

extern "C" int pdn_rmsnorm_fwd_f32(const float* x, const float* w, float* y, float* rms,
                                   int64_t rows, int cols, float eps, void* stream) {
  if (rows == 0) return PDN_OK;
  PDN_CHECK_ARG(x && w && y, "pdn_rmsnorm_fwd_f32: null operand");
  PDN_CHECK_ARG(cols > 0 && cols % 4 == 0 && cols <= 2048,
                "pdn_rmsnorm_fwd_f32: cols=%d must be a multiple of 4 and <= 2048", cols);
  PDN_CHECK_ARG((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) == 0, "pdn_rmsnorm_fwd_f32: 16B alignment");
  hipStream_t st = (hipStream_t)stream;
  const int g = wave_grid(rows), vpl = (cols / 4 + 63) / 64;
#define RF(V) hipLaunchKernelGGL((rmsnorm_fwd_kernel<V>), dim3(g), dim3(256), 0, st, x, w, y, rms, rows, cols, eps)
  switch (vpl) { case 1: RF(1); break; case 2: RF(2); break; case 3: RF(3); break; case 4: RF(4); break;
                 case 5: RF(5); break; case 6: RF(6); break; case 7: RF(7); break; default: RF(8); }
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

// dw: if dw != NULL, dw (+)= column sums (accumulate_dw selects += vs =). workspace from
// pdn_rmsnorm_bwd_workspace_bytes.  dx_residual (nullable, shape of dx) is added to dx: the
// gradient x already received from another consumer, folded in instead of a separate add pass.
extern "C" int pdn_rmsnorm_bwd_f32(const float* x, const float* w, const float* rms,
                                   const float* dy, const float* dx_residual, float* dx,
                                   float* dw, int accumulate_dw,
                                   int64_t rows, int cols, void* workspace,
                                   int64_t workspace_bytes, void* stream) {
  if (rows == 0) return PDN_OK;
  PDN_CHECK_ARG(x && w && rms && dy && dx, "pdn_rmsnorm_bwd_f32: null operand");
  PDN_CHECK_ARG(cols > 0 && cols % 4 == 0 && cols <= 2048,
                "pdn_rmsnorm_bwd_f32: cols=%d must be a multiple of 4 and <= 2048", cols);
  hipStream_t st = (hipStream_t)stream;
  int64_t nb = (rows + 15) / 16; if (nb > 1024) nb = 1024; if (nb < 1) nb = 1;
  float* part = nullptr;
  if (dw) {
    if (workspace_bytes < nb * (int64_t)cols * 4 || !workspace) {
      pdn_set_error("pdn_rmsnorm_bwd_f32: workspace too small");
      return PDN_EWORKSPACE;
    }
    part = (float*)workspace;
  }
  const int vpl = (cols / 4 + 63) / 64;
  const size_t shm = (size_t)4 * cols * sizeof(float);
#define RB(V) hipLaunchKernelGGL((rmsnorm_bwd_kernel<V>), dim3((unsigned)nb), dim3(256), shm, st, x, w, rms, dy, dx_residual, dx, part, rows, cols)
  switch (vpl) { case 1: RB(1); break; case 2: RB(2); break; case 3: RB(3); break; case 4: RB(4); break;
                 case 5: RB(5); break; case 6: RB(6); break; case 7: RB(7); break; default: RB(8); }
  PDN_LAUNCH_CHECK();
  if (dw) {
    hipLaunchKernelGGL(colsum_partials_kernel, dim3((cols + COLSUM_COLS - 1) / COLSUM_COLS), dim3(1024), 0, st, part,
                       (int)nb, cols, dw, accumulate_dw);
    PDN_LAUNCH_CHECK();
  }
  return PDN_OK;
}

// ======================================================================================
// SiLU / SwiGLU  (functional.py:39-40:  silu(x) = x / (1 + exp(-x)))
//   mode 0: y = silu(g)                 dg = dy * s*(1 + g*(1-s)),  s = 1/(1+exp(-g))
//   mode 1: y = silu(g) * u             dg = dy*u*s*(1+g*(1-s)),  du = dy*silu(g)
// ======================================================================================
__device__ __forceinline__ float silu_f(float g) { return g / (1.f + expf(-g)); }
__device__ __forceinline__ float dsilu_f(float g) {
  const float s = 1.f / (1.f + expf(-g));
  return s * (1.f + g * (1.f - s));
}

__global__ void swiglu_fwd_kernel(const float* __restrict__ g, const float* __restrict__ u,
                                  float* __restrict__ y, int64_t n4, int64_t total) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 a = reinterpret_cast<const float4*>(g)[i];
    float4 r;
    r.x = silu_f(a.x); r.y = silu_f(a.y); r.z = silu_f(a.z); r.w = silu_f(a.w);
    if (u) {
      const float4 b = reinterpret_cast<const float4*>(u)[i];
      r.x *= b.x; r.y *= b.y; r.z *= b.z; r.w *= b.w;
    }
    reinterpret_cast<float4*>(y)[i] = r;
  }
  for (int64_t i = n4 * 4 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += stride)
    y[i] = u ? silu_f(g[i]) * u[i] : silu_f(g[i]);
}

__global__ void swiglu_bwd_kernel(const float* __restrict__ g, const float* __restrict__ u,
                                  const float* __restrict__ dy, float* __restrict__ dg,
                                  float* __restrict__ du, int64_t n4, int64_t total) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 a = reinterpret_cast<const float4*>(g)[i];
    const float4 d = reinterpret_cast<const float4*>(dy)[i];
    float4 rg;
    rg.x = d.x * dsilu_f(a.x); rg.y = d.y * dsilu_f(a.y);
    rg.z = d.z * dsilu_f(a.z); rg.w = d.w * dsilu_f(a.w);
    if (u) {
      const float4 b = reinterpret_cast<const float4*>(u)[i];
      float4 ru;
      ru.x = d.x * silu_f(a.x); ru.y = d.y * silu_f(a.y); ru.z = d.z * silu_f(a.z); ru.w = d.w * silu_f(a.w);
      rg.x *= b.x; rg.y *= b.y; rg.z *= b.z; rg.w *= b.w;
      reinterpret_cast<float4*>(du)[i] = ru;
    }
    reinterpret_cast<float4*>(dg)[i] = rg;
  }
  for (int64_t i = n4 * 4 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += stride) {
    float r = dy[i] * dsilu_f(g[i]);
    if (u) { du[i] = dy[i] * silu_f(g[i]); r *= u[i]; }
    dg[i] = r;
  }
}

static inline int stream_grid(int64_t n4) {
  int64_t b = (n4 + 255) / 256;
  if (b > 2048) b = 2048; if (b < 1) b = 1;
  return (int)b;
}
static inline bool aligned16(const void* a, const void* b = nullptr, const void* c = nullptr,
                             const void* d = nullptr, const void* e = nullptr) {
  return (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d | (uintptr_t)e) & 15) == 0;
}

// u == NULL selects plain SiLU.
extern "C" int pdn_swiglu_fwd_f32(const float* g, const float* u, float* y, int64_t n, void* stream) {
  if (n == 0) return PDN_OK;
  PDN_CHECK_ARG(g && y && n > 0, "pdn_swiglu_fwd_f32: bad arguments");
  const int64_t n4 = aligned16(g, u, y) ? n / 4 : 0;
  hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(stream_grid(n4 ? n4 : n)), dim3(256), 0,
                     (hipStream_t)stream, g, u, y, n4, n);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}
extern "C" int pdn_swiglu_bwd_f32(const float* g, const float* u, const float* dy, float* dg,
                                  float* du, int64_t n, void* stream) {
  if (n == 0) return PDN_OK;
  PDN_CHECK_ARG(g && dy && dg && n > 0 && (!u || du), "pdn_swiglu_bwd_f32: bad arguments");
  const int64_t n4 = aligned16(g, u, dy, dg, du) ? n / 4 : 0;
  hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(stream_grid(n4 ? n4 : n)), dim3(256), 0,
                     (hipStream_t)stream, g, u, dy, dg, du, n4, n);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

// SwiGLU on a PACKED projection: gu is (rows, 2F) with gate in columns [0, F) and up in [F, 2F) (the two
// FFN projections written by one batched GEMM); y is (rows, F), dgu is (rows, 2F).  F % 4 == 0.
__global__ void swiglu_rows_fwd_kernel(const float* __restrict__ gu, float* __restrict__ y, int64_t rows, int F4) {
  const int64_t total = rows * F4, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / F4;
    const int c = (int)(i - r * F4);
    const float4* row = reinterpret_cast<const float4*>(gu) + r * 2 * F4;
    const float4 a = row[c], b = row[F4 + c];
    float4 o;
    o.x = silu_f(a.x) * b.x; o.y = silu_f(a.y) * b.y; o.z = silu_f(a.z) * b.z; o.w = silu_f(a.w) * b.w;
    reinterpret_cast<float4*>(y)[i] = o;
  }
}
__global__ void swiglu_rows_bwd_kernel(const float* __restrict__ gu, const float* __restrict__ dy,
                                       float* __restrict__ dgu, int64_t rows, int F4) {
  const int64_t total = rows * F4, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / F4;
    const int c = (int)(i - r * F4);
    const float4* row = reinterpret_cast<const float4*>(gu) + r * 2 * F4;
    float4* drow = reinterpret_cast<float4*>(dgu) + r * 2 * F4;
    const float4 a = row[c], b = row[F4 + c], g = reinterpret_cast<const float4*>(dy)[i];
    float4 dg, du;
    dg.x = g.x * b.x * dsilu_f(a.x); dg.y = g.y * b.y * dsilu_f(a.y);
    dg.z = g.z * b.z * dsilu_f(a.z); dg.w = g.w * b.w * dsilu_f(a.w);
    du.x = g.x * silu_f(a.x); du.y = g.y * silu_f(a.y); du.z = g.z * silu_f(a.z); du.w = g.w * silu_f(a.w);
    drow[c] = dg;
    drow[F4 + c] = du;
  }
}
extern "C" int pdn_swiglu_rows_fwd_f32(const float* gu, float* y, int64_t rows, int F, void* stream) {
  if (rows == 0 || F == 0) return PDN_OK;
  PDN_CHECK_ARG(gu && y && F % 4 == 0 && ((((uintptr_t)gu | (uintptr_t)y) & 15) == 0), "pdn_swiglu_rows_fwd_f32: bad arguments");
  hipLaunchKernelGGL(swiglu_rows_fwd_kernel, dim3(stream_grid(rows * (F / 4))), dim3(256), 0, (hipStream_t)stream, gu, y,
                     rows, F / 4);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}
extern "C" int pdn_swiglu_rows_bwd_f32(const float* gu, const float* dy, float* dgu, int64_t rows, int F, void* stream) {
  if (rows == 0 || F == 0) return PDN_OK;
  PDN_CHECK_ARG(gu && dy && dgu && F % 4 == 0 && ((((uintptr_t)gu | (uintptr_t)dy | (uintptr_t)dgu) & 15) == 0),
                "pdn_swiglu_rows_bwd_f32: bad arguments");
  hipLaunchKernelGGL(swiglu_rows_bwd_kernel, dim3(stream_grid(rows * (F / 4))), dim3(256), 0, (hipStream_t)stream, gu, dy,
                     dgu, rows, F / 4);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

// relu backward: dx = (maximum(0,x) == x) ? dy : 0   -> x >= 0 passes (grad at 0 is 1)
__global__ void relu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                float* __restrict__ dx, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride)
    dx[i] = (fmaxf(0.f, x[i]) == x[i]) ? dy[i] : 0.f;
}
extern "C" int pdn_relu_bwd_f32(const float* x, const float* dy, float* dx, int64_t n, void* stream) {
  if (n == 0) return PDN_OK;
  PDN_CHECK_ARG(x && dy && dx && n > 0, "pdn_relu_bwd_f32: bad arguments");
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(stream_grid(n / 2 + 1)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, n);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

// ======================================================================================
// RoPE on interleaved pairs (llm/llama/model.py:23-44).  x: (rows = B*L, heads, hd),
// tables cos/sin: (L, hd/2) already offset by start_pos.  sign=+1 forward, -1 backward.
//   y[2i]   = x[2i]*cos - sign*x[2i+1]*sin      y[2i+1] = sign*x[2i]*sin + x[2i+1]*cos
// ======================================================================================
__global__ void rope_kernel(const float* __restrict__ x, const float* __restrict__ cosT,
                            const float* __restrict__ sinT, float* __restrict__ y, int64_t rows,
                            int L, int heads, int half, float sign) {
  const int64_t pairs_per_row = (int64_t)heads * half;
  const int64_t total = rows * pairs_per_row;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t row = i / pairs_per_row;
    const int j = (int)(i - row * pairs_per_row) % half;
    const int pos = (int)(row % L);
    const float c = cosT[(int64_t)pos * half + j], s = sign * sinT[(int64_t)pos * half + j];
    const float2 v = reinterpret_cast<const float2*>(x)[i];
    float2 o;
    o.x = v.x * c - v.y * s;
    o.y = v.x * s + v.y * c;
    reinterpret_cast<float2*>(y)[i] = o;
  }
}
// the same on rows that are `x_rs` / `y_rs` floats apart (the q | k column blocks of a packed q | k | v projection,
// rotated in place before the persistent attention kernels read them: core/fused/attn.py)
// (x and y may be the SAME buffer -- core/fused/attn.py rotates q | k in place -- so neither is __restrict__: every
// thread reads its own pair before it writes it)
__global__ void rope_rows_kernel(const float* x, const float* __restrict__ cosT, const float* __restrict__ sinT,
                                 float* y, int64_t rows, int L, int heads, int half, int64_t x_rs, int64_t y_rs,
                                 float sign) {
  const int64_t pairs_per_row = (int64_t)heads * half;
  const int64_t total = rows * pairs_per_row;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t row = i / pairs_per_row;
    const int pr = (int)(i - row * pairs_per_row), j = pr % half;
    const int pos = (int)(row % L);
    const float c = cosT[(int64_t)pos * half + j], s = sign * sinT[(int64_t)pos * half + j];
    const float2 v = *reinterpret_cast<const float2*>(x + row * x_rs + 2 * pr);
    *reinterpret_cast<float2*>(y + row * y_rs + 2 * pr) = make_float2(v.x * c - v.y * s, v.x * s + v.y * c);
  }
}
extern "C" int pdn_rope_rows_f32(const float* x, const float* cos_t, const float* sin_t, float* y, int64_t rows, int L,
                                 int heads, int head_dim, int64_t x_row_stride, int64_t y_row_stride, int backward,
                                 void* stream) {
  if (rows == 0) return PDN_OK;
  PDN_CHECK_ARG(x && cos_t && sin_t && y, "pdn_rope_rows_f32: null operand");
  PDN_CHECK_ARG(L > 0 && heads > 0 && head_dim > 0 && head_dim % 2 == 0 && x_row_stride % 2 == 0 && y_row_stride % 2 == 0,
                "pdn_rope_rows_f32: bad dims");
  PDN_CHECK_ARG((((uintptr_t)x | (uintptr_t)y) & 7) == 0, "pdn_rope_rows_f32: 8B alignment");
  const int64_t total = rows * heads * (head_dim / 2);
  hipLaunchKernelGGL(rope_rows_kernel, dim3(stream_grid(total / 2 + 1)), dim3(256), 0, (hipStream_t)stream, x, cos_t, sin_t, y,
                     rows, L, heads, head_dim / 2, x_row_stride, y_row_stride, backward ? -1.f : 1.f);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

extern "C" int pdn_rope_f32(const float* x, const float* cos_t, const float* sin_t, float* y,
                            int64_t rows, int L, int heads, int head_dim, int backward,
                            void* stream) {
  if (rows == 0) return PDN_OK;
  PDN_CHECK_ARG(x && cos_t && sin_t && y, "pdn_rope_f32: null operand");
  PDN_CHECK_ARG(L > 0 && heads > 0 && head_dim > 0 && head_dim % 2 == 0, "pdn_rope_f32: bad dims");
  PDN_CHECK_ARG((((uintptr_t)x | (uintptr_t)y) & 7) == 0, "pdn_rope_f32: 8B alignment");
  const int64_t total = rows * heads * (head_dim / 2);
  hipLaunchKernelGGL(rope_kernel, dim3(stream_grid(total / 2 + 1)), dim3(256), 0, (hipStream_t)stream, x,
                     cos_t, sin_t, y, rows, L, heads, head_dim / 2, backward ? -1.f : 1.f);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

// ======================================================================================
// Embedding: gather rows / scatter back.
//  gather:          out[n, :] = W[ids[n], :]                          (bit-exact copy)
//  scatter-assign:  dW[ids[n], :] (+)= g[n, :] for the LAST n holding each id, matching
//                   NumPy `full[key] = grad` (tensor.py:939: last write wins on duplicates)
//  scatter-add:     dW[ids[n], :] += g[n, :] for every n (atomic; torch semantics, opt-in)
// ======================================================================================
__global__ void gather_rows_kernel(const float* __restrict__ W, const int64_t* __restrict__ ids,
                                   float* __restrict__ out, int64_t n, int D, int64_t w_rs,
                                   int64_t V, int* __restrict__ err) {
  const int lanes = blockDim.x;  // one workgroup per row
  for (int64_t r = blockIdx.x; r < n; r += gridDim.x) {
    int64_t id = ids[r];
    if (id < 0) id += V;
    if (id < 0 || id >= V) { if (threadIdx.x == 0) *err = 1; continue; }
    const float* src = W + id * w_rs;
    float* dst = out + r * (int64_t)D;
    for (int c = threadIdx.x; c < D; c += lanes) dst[c] = src[c];
  }
}
__global__ void last_occurrence_kernel(const int64_t* __restrict__ ids, int64_t n, int64_t V,
                                       int* __restrict__ last) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n; r += stride) {
    int64_t id = ids[r];
    if (id < 0) id += V;
    if (id >= 0 && id < V) atomicMax(&last[id], (int)r);
  }
}
__global__ void scatter_rows_kernel(const float* __restrict__ g, const int64_t* __restrict__ ids,
                                    float* __restrict__ dW, int64_t n, int D, int64_t V,
                                    const int* __restrict__ last, int mode,
                                    const float* __restrict__ row_owner, float owner_tag) {
  // mode 0: assign-last into zero (dW = g), 1: assign-last accumulating (dW += g), 2: atomic add
  for (int64_t r = blockIdx.x; r < n; r += gridDim.x) {
    int64_t id = ids[r];
    if (id < 0) id += V;
    if (id < 0 || id >= V) continue;
    if (mode != 2 && last[id] != (int)r) continue;
    if (row_owner && row_owner[id] != owner_tag) continue;   // a later data-parallel rank holds the last occurrence
    const float* src = g + r * (int64_t)D;
    float* dst = dW + id * (int64_t)D;
    if (mode == 0) for (int c = threadIdx.x; c < D; c += blockDim.x) dst[c] = src[c];
    else if (mode == 1) for (int c = threadIdx.x; c < D; c += blockDim.x) dst[c] += src[c];
    else for (int c = threadIdx.x; c < D; c += blockDim.x) atomicAdd(&dst[c], src[c]);
  }
}

// err_flag: device int set to 1 on an out-of-range id (checked lazily by the host shim).
extern "C" int pdn_embedding_gather_f32(const float* W, int64_t V, int D, int64_t w_row_stride,
                                        const int64_t* ids, int64_t n, float* out, int* err_flag,
                                        void* stream) {
  if (n == 0 || D == 0) return PDN_OK;
  PDN_CHECK_ARG(W && ids && out && err_flag && V > 0, "pdn_embedding_gather_f32: bad arguments");
  const int threads = D >= 256 ? 256 : (D >= 128 ? 128 : 64);
  const int g = (int)(n < 65535 ? n : 65535);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(g), dim3(threads), 0, (hipStream_t)stream, W, ids, out,
                     n, D, w_row_stride, V, err_flag);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

extern "C" int64_t pdn_embedding_scatter_workspace_bytes(int64_t V) { return V * 4; }

// mode: 0 assign (dW rows overwritten), 1 assign-last accumulated into dW (the engine's
// `grad += full_grad`), 2 atomic scatter-add.  dW is (V, D) contiguous.  row_owner (V floats, may be
// NULL): rows whose entry differs from owner_tag are skipped -- data parallel: only the rank holding
// the globally last occurrence of a token id contributes its row.
extern "C" int pdn_embedding_scatter_f32(const float* g, const int64_t* ids, int64_t n, float* dW,
                                         int64_t V, int D, int mode, const float* row_owner,
                                         float owner_tag, void* workspace,
                                         int64_t workspace_bytes, void* stream) {
  if (n == 0 || D == 0) return PDN_OK;
  PDN_CHECK_ARG(g && ids && dW && V > 0 && mode >= 0 && mode <= 2, "pdn_embedding_scatter_f32: bad arguments");
  PDN_CHECK_ARG(n <= 2147483647ll, "pdn_embedding_scatter_f32: too many rows");
  hipStream_t st = (hipStream_t)stream;
  int* last = (int*)workspace;
  if (mode != 2) {
    if (!workspace || workspace_bytes < V * 4) {
      pdn_set_error("pdn_embedding_scatter_f32: workspace too small");
      return PDN_EWORKSPACE;
    }
    PDN_HIP(hipMemsetAsync(last, 0xff, V * 4, st));  // -1
    hipLaunchKernelGGL(last_occurrence_kernel, dim3(stream_grid(n)), dim3(256), 0, st, ids, n, V, last);
    PDN_LAUNCH_CHECK();
  }
  const int threads = D >= 256 ? 256 : (D >= 128 ? 128 : 64);
  const int grid = (int)(n < 65535 ? n : 65535);
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid), dim3(threads), 0, st, g, ids, dW, n, D, V, last, mode,
                     row_owner, owner_tag);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

// Pick / un-pick one column per row:  out[n] = x[n, idx[n]]   (CE's neg_log_sm[range(N), y])
__global__ void take_cols_kernel(const float* __restrict__ x, const int64_t* __restrict__ idx,
                                 float* __restrict__ out, int64_t n, int64_t C, int64_t x_rs,
                                 int* __restrict__ err) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n; r += stride) {
    int64_t c = idx[r];
    if (c < 0) c += C;
    if (c < 0 || c >= C) { *err = 1; continue; }
    out[r] = x[r * x_rs + c];
  }
}
__global__ void put_cols_kernel(const float* __restrict__ g, const int64_t* __restrict__ idx,
                                float* __restrict__ dx, int64_t n, int64_t C) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n; r += stride) {
    int64_t c = idx[r];
    if (c < 0) c += C;
    if (c >= 0 && c < C) dx[r * C + c] = g[r];
  }
}
extern "C" int pdn_take_cols_f32(const float* x, int64_t n, int64_t C, int64_t x_row_stride,
                                 const int64_t* idx, float* out, int* err_flag, void* stream) {
  if (n == 0) return PDN_OK;
  PDN_CHECK_ARG(x && idx && out && err_flag && C > 0, "pdn_take_cols_f32: bad arguments");
  hipLaunchKernelGGL(take_cols_kernel, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream, x, idx, out, n, C, x_row_stride, err_flag);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}
// dx (n, C) contiguous must be pre-zeroed by the caller; rows are distinct so no races.
extern "C" int pdn_put_cols_f32(const float* g, const int64_t* idx, float* dx, int64_t n, int64_t C, void* stream) {
  if (n == 0) return PDN_OK;
  PDN_CHECK_ARG(g && idx && dx && C > 0, "pdn_put_cols_f32: bad arguments");
  hipLaunchKernelGGL(put_cols_kernel, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream, g, idx, dx, n, C);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

// ======================================================================================
// Cross entropy with integer targets (functional.py:364-381):
//   loss_n = log(sum_v exp(x[n,v] - m)) + m - x[n, t_n]      (shift m: row max; the
//   reference shifts by the global max -- the value is shift-invariant)
//   dx[n,v] = (exp(x[n,v] - lse_n) - [v == t_n]) * gscale
// One workgroup per row; the row (V = 32000 -> 125 KB) is re-read from L2 for pass 2.
// ======================================================================================
__global__ void ce_fwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ tgt,
                              float* __restrict__ loss_row, float* __restrict__ lse_row,
                              int64_t rows, int V, int* __restrict__ err) {
  __shared__ float red[16];
  for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
    const float* xr = x + row * (int64_t)V;
    const int n4 = (((uintptr_t)xr & 15) == 0) ? V >> 2 : 0;
    float m = -INFINITY;
    for (int i = threadIdx.x; i < n4; i += blockDim.x) {
      const float4 v = reinterpret_cast<const float4*>(xr)[i];
      m = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
    }
    for (int c = n4 * 4 + threadIdx.x; c < V; c += blockDim.x) m = fmaxf(m, xr[c]);
    m = block_max(m, red);
    float s = 0.f;
    for (int i = threadIdx.x; i < n4; i += blockDim.x) {
      const float4 v = reinterpret_cast<const float4*>(xr)[i];
      s += (expf(v.x - m) + expf(v.y - m)) + (expf(v.z - m) + expf(v.w - m));
    }
    for (int c = n4 * 4 + threadIdx.x; c < V; c += blockDim.x) s += expf(xr[c] - m);
    s = block_sum(s, red);
    if (threadIdx.x == 0) {
      const float lse = logf(s) + m;
      int64_t t = tgt[row];
      if (t < 0) t += V;
      if (t < 0 || t >= V) { *err = 1; t = 0; }
      lse_row[row] = lse;
      loss_row[row] = lse - xr[t];
    }
    __syncthreads();
  }
}
__global__ void ce_bwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ tgt,
                              const float* __restrict__ lse_row, const float* __restrict__ gscale_dev,
                              float gscale, float* __restrict__ dx, int64_t rows, int V) {
  const float gs = gscale_dev ? gscale * gscale_dev[0] : gscale;
  for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
    const float* xr = x + row * (int64_t)V;
    float* dr = dx + row * (int64_t)V;
    const float lse = lse_row[row];
    int64_t t = tgt[row];
    if (t < 0) t += V;
    const int n4 = ((((uintptr_t)xr | (uintptr_t)dr) & 15) == 0) ? V >> 2 : 0;
    for (int i = threadIdx.x; i < n4; i += blockDim.x) {
      const float4 v = reinterpret_cast<const float4*>(xr)[i];
      float4 r;
      r.x = expf(v.x - lse); r.y = expf(v.y - lse); r.z = expf(v.z - lse); r.w = expf(v.w - lse);
      const int c = 4 * i;
      if (t >= c && t < c + 4) {
        if (t == c) r.x -= 1.f; else if (t == c + 1) r.y -= 1.f;
        else if (t == c + 2) r.z -= 1.f; else r.w -= 1.f;
      }
      r.x *= gs; r.y *= gs; r.z *= gs; r.w *= gs;
      reinterpret_cast<float4*>(dr)[i] = r;
    }
    for (int c = n4 * 4 + threadIdx.x; c < V; c += blockDim.x)
      dr[c] = (expf(xr[c] - lse) - (c == t ? 1.f : 0.f)) * gs;
  }
}
// sum (or mean) of the per-row losses, one workgroup, fixed order.
__global__ void ce_reduce_kernel(const float* __restrict__ loss_row, int64_t rows, float scale,
                                 float* __restrict__ out) {
  __shared__ float red[16];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < rows; i += blockDim.x) s += loss_row[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[0] = s * scale;
}

// loss_row / lse_row: (rows,) scratch kept for backward.  loss_out: 1 float =
// (mean ? 1/rows : 1) * sum(loss_row).
template <bool COLSUM, bool WRITE = true>
__global__ void ce_fwd_bwd_reg_kernel(const float* __restrict__ x, const int64_t* __restrict__ tgt,
                                      float* __restrict__ loss_row, float* __restrict__ lse_row,
                                      float* __restrict__ dx, float gscale, int64_t rows, int V,
                                      int* __restrict__ err, float* __restrict__ colsum_part);

#define CE_REG_MAX_V 32768      // 1024 threads x 8 float4 held in registers
static inline bool ce_reg_row_ok(int V, const void* a, const void* b) {
  return V >= 4096 && V % 4 == 0 && V <= CE_REG_MAX_V && ((((uintptr_t)a | (uintptr_t)b) & 15) == 0);
}
static inline int ce_reg_grid(int64_t rows) { return (int)(rows < 256 ? rows : 256); }

extern "C" int pdn_cross_entropy_fwd_f32(const float* logits, const int64_t* targets, int64_t rows,
                                         int V, int mean, float* loss_row, float* lse_row,
                                         float* loss_out, int* err_flag, void* stream) {
  PDN_CHECK_ARG(rows > 0 && V > 0, "pdn_cross_entropy_fwd_f32: empty input");
  PDN_CHECK_ARG(logits && targets && loss_row && lse_row && loss_out && err_flag, "pdn_cross_entropy_fwd_f32: null operand");
  hipStream_t st = (hipStream_t)stream;
  // long rows: few, fat workgroups so the rows in flight (grid x V x 4 B) stay within L2 for the re-read pass
  const int ce_threads = V >= 4096 ? 1024 : 256;
  const int g = (int)(V >= 4096 ? (rows < 512 ? rows : 512) : (rows < 65535 ? rows : 65535));
  if (ce_reg_row_ok(V, logits, logits))            // one pass, the row held in registers
    hipLaunchKernelGGL((ce_fwd_bwd_reg_kernel<false, false>), dim3(ce_reg_grid(rows)), dim3(1024), 0, st, logits, targets,
                       loss_row, lse_row, (float*)nullptr, 1.f, rows, V, err_flag, (float*)nullptr);
  else
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(g), dim3(ce_threads), 0, st, logits, targets, loss_row, lse_row, rows, V, err_flag);
  PDN_LAUNCH_CHECK();
  hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(1024), 0, st, loss_row, rows,
                     mean ? 1.f / (float)rows : 1.f, loss_out);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}
// loss from row statistics that already exist (the vocabulary projection left lse[row], csrc/gemm_rowres.hip EPI 4):
// loss_row[r] = lse[r] - logits[r][target[r]], loss_out = (mean ? 1 / rows : 1) * sum -- one 4-byte gather per row
// instead of a pass over the logits.
__global__ void ce_rows_from_lse_kernel(const float* __restrict__ logits, int64_t ldl, const float* __restrict__ lse,
                                        const int64_t* __restrict__ tgt, int64_t rows, int V, float* __restrict__ loss_row,
                                        int* __restrict__ err) {
  const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r >= rows) return;
  int64_t t = tgt[r];
  if (t < 0 || t >= V) { *err = 1; t = 0; }
  loss_row[r] = lse[r] - logits[r * ldl + t];
}
extern "C" int pdn_cross_entropy_from_lse_f32(const float* logits, int64_t ldl, const float* lse, const int64_t* targets,
                                              int64_t rows, int V, int mean, float* loss_row, float* loss_out,
                                              int* err_flag, void* stream) {
  PDN_CHECK_ARG(rows > 0 && V > 0, "pdn_cross_entropy_from_lse_f32: empty input");
  PDN_CHECK_ARG(logits && lse && targets && loss_row && loss_out && err_flag, "pdn_cross_entropy_from_lse_f32: null operand");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ce_rows_from_lse_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, logits, ldl, lse,
                     targets, rows, V, loss_row, err_flag);
  PDN_LAUNCH_CHECK();
  hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(1024), 0, st, loss_row, rows, mean ? 1.f / (float)rows : 1.f, loss_out);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}
// dlogits = (softmax(logits) - onehot) * gscale * (upstream ? upstream[0] : 1).
// `dlogits` may alias `logits` (in-place).  gscale is 1/rows for reduction='mean'.
extern "C" int pdn_cross_entropy_bwd_f32(const float* logits, const int64_t* targets,
                                         const float* lse_row, const float* upstream, float gscale,
                                         float* dlogits, int64_t rows, int V, void* stream) {
  if (rows == 0) return PDN_OK;
  PDN_CHECK_ARG(logits && targets && lse_row && dlogits && V > 0, "pdn_cross_entropy_bwd_f32: bad arguments");
  const int g = (int)(rows < 65535 ? rows : 65535);
  hipLaunchKernelGGL(ce_bwd_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, logits, targets, lse_row, upstream, gscale, dlogits, rows, V);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

// ======================================================================================
// Adam, multi-tensor (optim/optimizer.py:185-196), one launch for all parameters.
// table: device int64[nchunks][5] = {p, g, m, v (addresses), n (elements in chunk)}.
//   g' = g*grad_scale + wd*p;  m = b1*m + (1-b1)*g';  v = b2*v + (1-b2)*g'^2;
//   p -= step * m / (sqrt(v) + eps)        step = lr * sqrt(1-b2^t)/(1-b1^t)  (host scalar;
//   eps is added to sqrt(v) WITHOUT bias correction -- reference semantics, not PyTorch's)
// ======================================================================================
__global__ void adam_multi_kernel(const int64_t* __restrict__ table, float step, float b1,
                                  float b2, float one_m_b1, float one_m_b2, float eps, float wd,
                                  float grad_scale, const float* __restrict__ step_dev) {
  if (step_dev) step = *step_dev;            // replayed from a hipGraph: the step size lives on the device
  const int64_t* e = table + (int64_t)blockIdx.x * 5;
  float* p = (float*)e[0]; const float* g = (const float*)e[1];
  float* m = (float*)e[2]; float* v = (float*)e[3];
  const int n = (int)e[4];
  const bool al = ((e[0] | e[1] | e[2] | e[3]) & 15) == 0;
  const int n4 = al ? n >> 2 : 0;
  for (int i = threadIdx.x; i < n4; i += blockDim.x) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    const float4 gv0 = reinterpret_cast<const float4*>(g)[i];
    float4 mv = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
#define ADAM1(P, G, M, V)                                 \
  {                                                       \
    const float gg = G * grad_scale + wd * P;             \
    M = M * b1 + one_m_b1 * gg;                           \
    V = V * b2 + one_m_b2 * (gg * gg);                    \
    P -= step * M / (sqrtf(V) + eps);                     \
  }
    ADAM1(pv.x, gv0.x, mv.x, vv.x) ADAM1(pv.y, gv0.y, mv.y, vv.y)
    ADAM1(pv.z, gv0.z, mv.z, vv.z) ADAM1(pv.w, gv0.w, mv.w, vv.w)
    reinterpret_cast<float4*>(p)[i] = pv;
    reinterpret_cast<float4*>(m)[i] = mv;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  for (int i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) {
    float P = p[i], M = m[i], V = v[i];
    const float G = g[i];
    ADAM1(P, G, M, V)
    p[i] = P; m[i] = M; v[i] = V;
  }
}
extern "C" int pdn_adam_multi_f32(const int64_t* chunk_table_dev, int nchunks, float step,
                                  float beta1, float beta2, float one_minus_beta1,
                                  float one_minus_beta2, float eps, float weight_decay,
                                  float grad_scale, void* stream) {
  if (nchunks == 0) return PDN_OK;
  PDN_CHECK_ARG(chunk_table_dev && nchunks > 0, "pdn_adam_multi_f32: bad arguments");
  hipLaunchKernelGGL(adam_multi_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream,
                     chunk_table_dev, step, beta1, beta2, one_minus_beta1, one_minus_beta2, eps,
                     weight_decay, grad_scale, (const float*)nullptr);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

// Step counter on the device, for optimizer steps replayed from a hipGraph (no host value can change
// between replays): state = {t, lr} as doubles; each launch writes step_out = lr * sqrt(1-b2^t)/(1-b1^t)
// (optimizer.py:191, the reference's host scalar a_t) and advances t.
__global__ void adam_tick_kernel(double* __restrict__ state, float* __restrict__ step_out, double b1, double b2) {
  const double t = state[0], lr = state[1];
  *step_out = (float)(lr * sqrt(1.0 - pow(b2, t)) / (1.0 - pow(b1, t)));
  state[0] = t + 1.0;
}
extern "C" int pdn_adam_multi_tick_f32(const int64_t* chunk_table_dev, int nchunks, double* state_dev,
                                       float* step_dev, float beta1, float beta2, float eps, float weight_decay,
                                       float grad_scale, void* stream) {
  PDN_CHECK_ARG(state_dev && step_dev, "pdn_adam_multi_tick_f32: null state");
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state_dev, step_dev, (double)beta1,
                     (double)beta2);
  PDN_LAUNCH_CHECK();
  if (nchunks == 0) return PDN_OK;
  PDN_CHECK_ARG(chunk_table_dev && nchunks > 0, "pdn_adam_multi_tick_f32: bad arguments");
  hipLaunchKernelGGL(adam_multi_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, chunk_table_dev, 0.f,
                     beta1, beta2, 1.f - beta1, 1.f - beta2, eps, weight_decay, grad_scale, (const float*)step_dev);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

// ======================================================================================
// Cross entropy, forward and backward in ONE pass over HBM.  The gradient of the loss with
// respect to the logits does not depend on anything produced later, so it is written while the
// row is still in L2: pass 1 max, pass 2 sum-exp, pass 3 dlogits = (softmax - onehot) * gscale.
// Backward then only has to apply the upstream scalar, which is 1 for `loss.backward()`:
// pdn_scale_by_device_scalar_f32 reads it ON THE DEVICE and returns without touching the
// buffer when it is exactly 1 (no host sync, no extra pass).
// ======================================================================================
__global__ void ce_fwd_bwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ tgt,
                                  float* __restrict__ loss_row, float* __restrict__ lse_row,
                                  float* __restrict__ dx, float gscale, int64_t rows, int V,
                                  int* __restrict__ err) {
  __shared__ float red[16];
  for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
    const float* xr = x + row * (int64_t)V;
    float* dr = dx + row * (int64_t)V;
    const int n4 = ((((uintptr_t)xr | (uintptr_t)dr) & 15) == 0) ? V >> 2 : 0;
    float m = -INFINITY;
    for (int i = threadIdx.x; i < n4; i += blockDim.x) {
      const float4 v = reinterpret_cast<const float4*>(xr)[i];
      m = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
    }
    for (int c = n4 * 4 + threadIdx.x; c < V; c += blockDim.x) m = fmaxf(m, xr[c]);
    m = block_max(m, red);
    float s = 0.f;
    for (int i = threadIdx.x; i < n4; i += blockDim.x) {
      const float4 v = reinterpret_cast<const float4*>(xr)[i];
      s += (expf(v.x - m) + expf(v.y - m)) + (expf(v.z - m) + expf(v.w - m));
    }
    for (int c = n4 * 4 + threadIdx.x; c < V; c += blockDim.x) s += expf(xr[c] - m);
    s = block_sum(s, red);
    const float lse = logf(s) + m;
    int64_t t = tgt[row];
    if (t < 0) t += V;
    if (t < 0 || t >= V) { if (threadIdx.x == 0) *err = 1; t = 0; }
    if (threadIdx.x == 0) {
      lse_row[row] = lse;
      loss_row[row] = lse - xr[t];
    }
    for (int i = threadIdx.x; i < n4; i += blockDim.x) {
      const float4 v = reinterpret_cast<const float4*>(xr)[i];
      float4 r;
      r.x = expf(v.x - lse); r.y = expf(v.y - lse); r.z = expf(v.z - lse); r.w = expf(v.w - lse);
      const int c = 4 * i;
      if (t >= c && t < c + 4) {
        if (t == c) r.x -= 1.f; else if (t == c + 1) r.y -= 1.f;
        else if (t == c + 2) r.z -= 1.f; else r.w -= 1.f;
      }
      r.x *= gscale; r.y *= gscale; r.z *= gscale; r.w *= gscale;
      reinterpret_cast<float4*>(dr)[i] = r;
    }
    for (int c = n4 * 4 + threadIdx.x; c < V; c += blockDim.x)
      dr[c] = (expf(xr[c] - lse) - (c == t ? 1.f : 0.f)) * gscale;
    __syncthreads();
  }
}


// A handful of classes (the 10 digits of examples/pydynet/mnist.py): one THREAD per row -- a workgroup per 40-byte row
// spends its time in three block reductions (150 us for 65536 x 10; this: the 5 MB at memory rate).
template <int VMAX>
__global__ __launch_bounds__(256) void ce_small_kernel(const float* __restrict__ x, const int64_t* __restrict__ tgt,
                                                       float* __restrict__ loss_row, float* __restrict__ lse_row,
                                                       float* __restrict__ dx, float gscale, int64_t rows, int V,
                                                       int* __restrict__ err) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  const float* xr = x + row * V;
  float v[VMAX];
  float m = -INFINITY;
#pragma unroll
  for (int c = 0; c < VMAX; ++c) {
    v[c] = c < V ? xr[c] : -INFINITY;
    m = fmaxf(m, v[c]);
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < VMAX; ++c) s += c < V ? expf(v[c] - m) : 0.f;
  const float lse = logf(s) + m;
  int64_t t = tgt[row];
  if (t < 0) t += V;
  if (t < 0 || t >= V) { *err = 1; t = 0; }
  lse_row[row] = lse;
  float xt = 0.f;
#pragma unroll
  for (int c = 0; c < VMAX; ++c) xt = (c == (int)t) ? v[c] : xt;
  loss_row[row] = lse - xt;
  if (dx) {
    float* dr = dx + row * V;
#pragma unroll
    for (int c = 0; c < VMAX; ++c)
      if (c < V) dr[c] = (expf(v[c] - lse) - (c == (int)t ? 1.f : 0.f)) * gscale;
  }
}

static bool ce_small_launch(const float* logits, const int64_t* targets, int64_t rows, int V, float gscale, float* loss_row,
                            float* lse_row, float* dlogits, int* err_flag, hipStream_t st) {
  if (V > 32 || rows < 1024 || rows > (1ll << 30)) return false;
  const dim3 g((unsigned)((rows + 255) / 256));
  if (V <= 16)
    hipLaunchKernelGGL((ce_small_kernel<16>), g, dim3(256), 0, st, logits, targets, loss_row, lse_row, dlogits, gscale, rows, V, err_flag);
  else
    hipLaunchKernelGGL((ce_small_kernel<32>), g, dim3(256), 0, st, logits, targets, loss_row, lse_row, dlogits, gscale, rows, V, err_flag);
  pdn_count(PDN_CNT_CE_SMALL);
  return true;
}

// Bytes of workspace needed for the fused column sums of dlogits (the bias gradient of the layer
// that produced the logits); 0 when this shape takes the generic path, which has no such fusion.
extern "C" int64_t pdn_cross_entropy_colsum_workspace_bytes(int64_t rows, int V) {
  if (!(V >= 4096 && V % 4 == 0 && V <= CE_REG_MAX_V) || rows <= 0) return 0;
  return (int64_t)ce_reg_grid(rows) * V * 4;
}

extern "C" int pdn_cross_entropy_fwd_bwd_f32(const float* logits, const int64_t* targets, int64_t rows,
                                             int V, int mean, float gscale, float* loss_row,
                                             float* lse_row, float* loss_out, float* dlogits,
                                             float* dlogits_colsum, void* workspace,
                                             int64_t workspace_bytes, int* err_flag, void* stream) {
  PDN_CHECK_ARG(rows > 0 && V > 0, "pdn_cross_entropy_fwd_bwd_f32: empty input");
  PDN_CHECK_ARG(logits && targets && loss_row && lse_row && loss_out && dlogits && err_flag,
                "pdn_cross_entropy_fwd_bwd_f32: null operand");
  hipStream_t st = (hipStream_t)stream;
  const int ce_threads = V >= 4096 ? 1024 : 256;   // see pdn_cross_entropy_fwd_f32
  const int g = (int)(V >= 4096 ? (rows < 512 ? rows : 512) : (rows < 65535 ? rows : 65535));
  const bool reg_row = ce_reg_row_ok(V, logits, dlogits);
  if (dlogits_colsum) {
    if (!reg_row) {
      pdn_set_error("pdn_cross_entropy_fwd_bwd_f32: fused column sums need 4096 <= V <= %d, V %% 4 == 0, aligned rows", CE_REG_MAX_V);
      return PDN_EUNSUPPORTED;
    }
    if (!workspace || workspace_bytes < pdn_cross_entropy_colsum_workspace_bytes(rows, V)) {
      pdn_set_error("pdn_cross_entropy_fwd_bwd_f32: workspace too small");
      return PDN_EWORKSPACE;
    }
  }
  if (reg_row) {
    const int gl = ce_reg_grid(rows);
    if (dlogits_colsum)
      hipLaunchKernelGGL((ce_fwd_bwd_reg_kernel<true>), dim3(gl), dim3(1024), 0, st, logits, targets, loss_row,
                         lse_row, dlogits, gscale, rows, V, err_flag, (float*)workspace);
    else
      hipLaunchKernelGGL((ce_fwd_bwd_reg_kernel<false>), dim3(gl), dim3(1024), 0, st, logits, targets, loss_row,
                         lse_row, dlogits, gscale, rows, V, err_flag, (float*)nullptr);
    PDN_LAUNCH_CHECK();
    if (dlogits_colsum) {
      hipLaunchKernelGGL(colsum_partials_kernel, dim3((V + COLSUM_COLS - 1) / COLSUM_COLS), dim3(1024), 0, st,
                         (const float*)workspace, gl, V, dlogits_colsum, 0);
      PDN_LAUNCH_CHECK();
    }
  } else if (ce_small_launch(logits, targets, rows, V, gscale, loss_row, lse_row, dlogits, err_flag, st)) {
    PDN_LAUNCH_CHECK();
  } else {
    hipLaunchKernelGGL(ce_fwd_bwd_kernel, dim3(g), dim3(ce_threads), 0, st, logits, targets, loss_row,
                       lse_row, dlogits, gscale, rows, V, err_flag);
    PDN_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(1024), 0, st, loss_row, rows,
                     mean ? 1.f / (float)rows : 1.f, loss_out);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

__global__ void scale_by_device_scalar_kernel(float* __restrict__ x, int64_t n,
                                              const float* __restrict__ scalar) {
  const float s = scalar[0];
  if (s == 1.0f) return;                                   // the common case: nothing to do
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride) x[i] *= s;
}

// x[i] *= scalar_dev[0]; a no-op pass (blocks exit after one load) when the scalar is exactly 1.
extern "C" int pdn_scale_by_device_scalar_f32(float* x, int64_t n, const float* scalar_dev, void* stream) {
  if (n == 0) return PDN_OK;
  PDN_CHECK_ARG(x && scalar_dev && n > 0, "pdn_scale_by_device_scalar_f32: bad arguments");
  hipLaunchKernelGGL(scale_by_device_scalar_kernel, dim3(stream_grid(n / 4 + 1)), dim3(256), 0,
                     (hipStream_t)stream, x, n, scalar_dev);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

// ---- cross entropy fwd+bwd with the row held in registers ---------------------------------------
// One 1024-thread workgroup per CU walks rows; a vocabulary row of up to 32768 floats lives in the
// registers of the workgroup (8 float4 per thread), so it is read from HBM exactly once and never
// touches LDS: max -> exp (kept in place) -> sum -> dlogits = e / sum - onehot.  The NEXT row is
// fetched into a second register set while the current one is reduced and written, so the HBM
// read stream overlaps the exp / store work.  Algorithmic traffic = 4 B read + 4 B written per logit.
// COLSUM: per-thread column sums of dlogits over the workgroup's rows -> one partial row per
// workgroup (the bias gradient of the vocabulary projection, summed by colsum_partials_kernel).
// WRITE = false: row statistics only (lse, loss) -- the forward of the fused linear + cross-entropy node, whose
// backward forms the gradient inside the two GEMMs: one read of the logits, nothing written back.
template <bool COLSUM, bool WRITE>
__global__ __launch_bounds__(1024) void ce_fwd_bwd_reg_kernel(
    const float* __restrict__ x, const int64_t* __restrict__ tgt, float* __restrict__ loss_row,
    float* __restrict__ lse_row, float* __restrict__ dx, float gscale, int64_t rows, int V,
    int* __restrict__ err, float* __restrict__ colsum_part) {
  constexpr int NV = CE_REG_MAX_V / 4 / 1024;     // float4 per thread
  __shared__ float red[16];
  __shared__ float xt_s;
  const int n4 = V >> 2, tid = threadIdx.x;
  float4 cur[NV], nxt[NV], cs[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    cs[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    nxt[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  int64_t row = blockIdx.x;
  if (row < rows) {
    const float4* xr = reinterpret_cast<const float4*>(x + row * (int64_t)V);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int i = tid + 1024 * j;
      if (i < n4) nxt[j] = xr[i];
    }
  }
  for (; row < rows; row += gridDim.x) {
    int64_t t = tgt[row];
    if (t < 0) t += V;
    if (t < 0 || t >= V) { if (tid == 0) *err = 1; t = 0; }
    const int t4 = (int)(t >> 2), tc = (int)(t & 3);
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int i = tid + 1024 * j;
      cur[j] = nxt[j];
      if (i < n4) {
        const float4 v = cur[j];
        m = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
        if (i == t4) xt_s = tc == 0 ? v.x : (tc == 1 ? v.y : (tc == 2 ? v.z : v.w));
      }
    }
    const int64_t nrow = row + gridDim.x;
    if (nrow < rows) {
      const float4* xn = reinterpret_cast<const float4*>(x + nrow * (int64_t)V);
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int i = tid + 1024 * j;
        if (i < n4) nxt[j] = xn[i];
      }
    }
    m = block_max(m, red);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int i = tid + 1024 * j;
      if (i < n4) {
        float4 e;
        if (WRITE) {
          e.x = expf(cur[j].x - m); e.y = expf(cur[j].y - m); e.z = expf(cur[j].z - m); e.w = expf(cur[j].w - m);
        } else {      // statistics only: the hardware exponential (the pass is VALU-bound with expf: 2.1 G elements)
          e.x = __expf(cur[j].x - m); e.y = __expf(cur[j].y - m); e.z = __expf(cur[j].z - m); e.w = __expf(cur[j].w - m);
        }
        cur[j] = e;
        s += (e.x + e.y) + (e.z + e.w);
      }
    }
    s = block_sum(s, red);            // (its barriers also publish xt_s)
    const float lse = logf(s) + m;
    const float inv = 1.f / s;
    if (tid == 0) {
      lse_row[row] = lse;
      loss_row[row] = lse - xt_s;
    }
    float4* dr = reinterpret_cast<float4*>(dx + row * (int64_t)V);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int i = tid + 1024 * j;
      if (WRITE && i < n4) {
        float4 r;
        r.x = cur[j].x * inv; r.y = cur[j].y * inv; r.z = cur[j].z * inv; r.w = cur[j].w * inv;
        if (i == t4) {
          if (tc == 0) r.x -= 1.f; else if (tc == 1) r.y -= 1.f;
          else if (tc == 2) r.z -= 1.f; else r.w -= 1.f;
        }
        r.x *= gscale; r.y *= gscale; r.z *= gscale; r.w *= gscale;
        dr[i] = r;
        if (COLSUM) { cs[j].x += r.x; cs[j].y += r.y; cs[j].z += r.z; cs[j].w += r.w; }
      }
    }
    __syncthreads();                  // xt_s / red are rewritten by the next row
  }
  if (COLSUM) {
    float4* part = reinterpret_cast<float4*>(colsum_part + (int64_t)blockIdx.x * V);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int i = tid + 1024 * j;
      if (i < n4) part[i] = cs[j];
    }
  }
}

// ======================================================================================
// GRU cell gate algebra (nn/modules/rnn.py:537-544):
//   [z, r] = sigmoid(x Wx1 + h Wh1 + b1);  n = tanh(x Wx2 + (r*h) Wh2 + b2);  h' = (1-z) h + z n
// The four matrix products run on the GEMM kernel; these kernels are everything in between, one
// pass each (the reference spends ~20 tape nodes per step on them).  sigmoid / tanh use the
// reference's overflow-safe piecewise forms (core/tensor.py:999-1003, 1012-1016).
// ======================================================================================
__device__ __forceinline__ float ref_sigmoid(float x) {
  return x > 0.f ? 1.f / (1.f + expf(-x)) : 1.f - 1.f / (1.f + expf(x));
}
__device__ __forceinline__ float ref_tanh(float x) {
  return x > 0.f ? 2.f / (1.f + expf(-2.f * x)) - 1.f : 1.f - 2.f / (1.f + expf(2.f * x));
}

__global__ void gru_gates_fwd_kernel(const float* __restrict__ g1, const float* __restrict__ h,
                                     float* __restrict__ z, float* __restrict__ r,
                                     float* __restrict__ rh, int64_t n, int H) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / H;
    const int j = (int)(i - b * H);
    const float zz = ref_sigmoid(g1[b * 2 * H + j]), rr = ref_sigmoid(g1[b * 2 * H + H + j]);
    z[i] = zz; r[i] = rr; rh[i] = rr * h[i];
  }
}

__global__ void gru_out_fwd_kernel(const float* __restrict__ g2, const float* __restrict__ z,
                                   const float* __restrict__ h, float* __restrict__ nn,
                                   float* __restrict__ hnew, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float t = ref_tanh(g2[i]), zz = z[i];
    nn[i] = t;
    hnew[i] = (1.f - zz) * h[i] + zz * t;
  }
}

// dG2 = dh' z (1 - n^2);  dG1[:, :H] = dh' (n - h) z (1 - z);  dh = dh' (1 - z)
__global__ void gru_out_bwd_kernel(const float* __restrict__ dhn, const float* __restrict__ z,
                                   const float* __restrict__ nn, const float* __restrict__ h,
                                   float* __restrict__ dg2, float* __restrict__ dg1,
                                   float* __restrict__ dh, int64_t n, int H) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / H;
    const int j = (int)(i - b * H);
    const float g = dhn[i], zz = z[i], t = nn[i];
    dg2[i] = (1.f - t * t) * (g * zz);
    dg1[b * 2 * H + j] = zz * (1.f - zz) * (g * (t - h[i]));
    dh[i] = g * (1.f - zz);
  }
}

// drh = dG2 Wh2^T arrives from the GEMM:  dG1[:, H:] = drh h r (1 - r);  dh += drh r
__global__ void gru_gates_bwd_kernel(const float* __restrict__ drh, const float* __restrict__ r,
                                     const float* __restrict__ h, float* __restrict__ dg1,
                                     float* __restrict__ dh, int64_t n, int H) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / H;
    const int j = (int)(i - b * H);
    const float d = drh[i], rr = r[i];
    dg1[b * 2 * H + H + j] = rr * (1.f - rr) * (d * h[i]);
    dh[i] += d * rr;
  }
}

extern "C" int pdn_gru_gates_fwd_f32(const float* g1, const float* h, float* z, float* r, float* rh,
                                     int64_t B, int H, void* stream) {
  if (B == 0 || H == 0) return PDN_OK;
  PDN_CHECK_ARG(g1 && h && z && r && rh && B > 0 && H > 0, "pdn_gru_gates_fwd_f32: bad arguments");
  hipLaunchKernelGGL(gru_gates_fwd_kernel, dim3(stream_grid(B * H)), dim3(256), 0, (hipStream_t)stream, g1, h,
                     z, r, rh, B * H, H);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

extern "C" int pdn_gru_out_fwd_f32(const float* g2, const float* z, const float* h, float* n,
                                   float* hnew, int64_t B, int H, void* stream) {
  if (B == 0 || H == 0) return PDN_OK;
  PDN_CHECK_ARG(g2 && z && h && n && hnew && B > 0 && H > 0, "pdn_gru_out_fwd_f32: bad arguments");
  hipLaunchKernelGGL(gru_out_fwd_kernel, dim3(stream_grid(B * H)), dim3(256), 0, (hipStream_t)stream, g2, z, h,
                     n, hnew, B * H);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

extern "C" int pdn_gru_out_bwd_f32(const float* dhnew, const float* z, const float* n, const float* h,
                                   float* dg2, float* dg1, float* dh, int64_t B, int H, void* stream) {
  if (B == 0 || H == 0) return PDN_OK;
  PDN_CHECK_ARG(dhnew && z && n && h && dg2 && dg1 && dh && B > 0 && H > 0, "pdn_gru_out_bwd_f32: bad arguments");
  hipLaunchKernelGGL(gru_out_bwd_kernel, dim3(stream_grid(B * H)), dim3(256), 0, (hipStream_t)stream, dhnew, z,
                     n, h, dg2, dg1, dh, B * H, H);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

extern "C" int pdn_gru_gates_bwd_f32(const float* drh, const float* r, const float* h, float* dg1,
                                     float* dh, int64_t B, int H, void* stream) {
  if (B == 0 || H == 0) return PDN_OK;
  PDN_CHECK_ARG(drh && r && h && dg1 && dh && B > 0 && H > 0, "pdn_gru_gates_bwd_f32: bad arguments");
  hipLaunchKernelGGL(gru_gates_bwd_kernel, dim3(stream_grid(B * H)), dim3(256), 0, (hipStream_t)stream, drh, r,
                     h, dg1, dh, B * H, H);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

// ======================================================================================
// Column-statistics normalisation: the reference's LayerNorm (nn/modules/norm.py:203-218 -- its
// statistics run over the LEADING axes, i.e. per feature over all tokens, with running averages)
// and BatchNorm1d on (N, C) (norm.py:60-74).  x is (rows, cols) row-major.
//   mean_c = sum_r x / R;  var_c = sum_r (x - mean_c)^2 / R;  y = (x - mean_c) / sqrt(var_c + eps) * w + b
//   running = (1 - m) * running + m * stat                       (norm.py:211-214)
// backward:  db = sum_r dy;  dw = sum_r dy * xhat;  dx = w * rstd * (dy - db/R - xhat * dw/R)
// Statistics: per 256-row chunk shifted sums (shift = first row of the chunk, so the squares do
// not cancel), chunks merged in a fixed order with the pairwise-variance formula.
// ======================================================================================
#define CN_ROWS 256      // rows per chunk
#define CN_COLS 32       // columns per workgroup (x 8 row lanes)

__global__ __launch_bounds__(256) void colnorm_stat_partial_kernel(const float* __restrict__ x, int64_t rows, int cols,
                                                                   float* __restrict__ part) {
  __shared__ float sm[8][CN_COLS], sq[8][CN_COLS];
  const int tx = threadIdx.x & (CN_COLS - 1), ty = threadIdx.x / CN_COLS;
  const int c = blockIdx.x * CN_COLS + tx;
  const int64_t r0 = (int64_t)blockIdx.y * CN_ROWS;
  const int64_t r1 = r0 + CN_ROWS < rows ? r0 + CN_ROWS : rows;
  float s = 0.f, q = 0.f, shift = 0.f;
  if (c < cols) {
    shift = x[r0 * cols + c];
    for (int64_t r = r0 + ty; r < r1; r += 8) {
      const float d = x[r * cols + c] - shift;
      s += d; q += d * d;
    }
  }
  sm[ty][tx] = s; sq[ty][tx] = q;
  __syncthreads();
  if (ty == 0 && c < cols) {
    float S = sm[0][tx], Q = sq[0][tx];
#pragma unroll
    for (int k = 1; k < 8; ++k) { S += sm[k][tx]; Q += sq[k][tx]; }
    const float n = (float)(r1 - r0);
    float* dst = part + ((int64_t)blockIdx.y * cols + c) * 2;
    dst[0] = shift + S / n;            // chunk mean
    dst[1] = Q - S * S / n;            // chunk sum of squared deviations
  }
}

__global__ void colnorm_stat_finish_kernel(const float* __restrict__ part, int64_t rows, int cols, int nchunks,
                                           float eps, float momentum, float* __restrict__ mean,
                                           float* __restrict__ rstd, float* __restrict__ run_mean,
                                           float* __restrict__ run_var) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  float n = 0.f, mu = 0.f, m2 = 0.f;
  for (int k = 0; k < nchunks; ++k) {
    const int64_t r0 = (int64_t)k * CN_ROWS;
    const float nb = (float)((r0 + CN_ROWS < rows ? r0 + CN_ROWS : rows) - r0);
    const float mb = part[((int64_t)k * cols + c) * 2], qb = part[((int64_t)k * cols + c) * 2 + 1];
    const float tot = n + nb, d = mb - mu;
    m2 += qb + d * d * (n * nb / tot);
    mu += d * (nb / tot);
    n = tot;
  }
  const float var = m2 / n;
  mean[c] = mu;
  rstd[c] = 1.f / sqrtf(var + eps);
  if (run_mean) run_mean[c] = run_mean[c] * (1.f - momentum) + momentum * mu;
  if (run_var) run_var[c] = run_var[c] * (1.f - momentum) + momentum * var;
}

__global__ void colnorm_apply_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                     const float* __restrict__ b, const float* __restrict__ mean,
                                     const float* __restrict__ rstd, float* __restrict__ y, int64_t n, int cols) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cols);
    y[i] = (x[i] - mean[c]) * rstd[c] * w[c] + b[c];
  }
}

// partial column sums of dy and dy * xhat per 256-row chunk: part[chunk][{0,1}][cols]
__global__ __launch_bounds__(256) void colnorm_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                  const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                  int64_t rows, int cols, float* __restrict__ part) {
  __shared__ float sa[8][CN_COLS], sb[8][CN_COLS];
  const int tx = threadIdx.x & (CN_COLS - 1), ty = threadIdx.x / CN_COLS;
  const int c = blockIdx.x * CN_COLS + tx;
  const int64_t r0 = (int64_t)blockIdx.y * CN_ROWS;
  const int64_t r1 = r0 + CN_ROWS < rows ? r0 + CN_ROWS : rows;
  float a = 0.f, bsum = 0.f;
  if (c < cols) {
    const float mu = mean[c], rs = rstd[c];
    for (int64_t r = r0 + ty; r < r1; r += 8) {
      const float g = dy[r * cols + c];
      a += g; bsum += g * ((x[r * cols + c] - mu) * rs);
    }
  }
  sa[ty][tx] = a; sb[ty][tx] = bsum;
  __syncthreads();
  if (ty == 0 && c < cols) {
    float A = sa[0][tx], Bv = sb[0][tx];
#pragma unroll
    for (int k = 1; k < 8; ++k) { A += sa[k][tx]; Bv += sb[k][tx]; }
    part[((int64_t)blockIdx.y * 2) * cols + c] = A;
    part[((int64_t)blockIdx.y * 2 + 1) * cols + c] = Bv;
  }
}

// sums[0][c] = sum_chunks part[k][0][c] (db), sums[1][c] = ... (dw), fixed order
__global__ void colnorm_bwd_finish_kernel(const float* __restrict__ part, int cols, int nchunks,
                                          float* __restrict__ sums) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  float a = 0.f, b = 0.f;
  for (int k = 0; k < nchunks; ++k) {
    a += part[((int64_t)k * 2) * cols + c];
    b += part[((int64_t)k * 2 + 1) * cols + c];
  }
  sums[c] = a; sums[cols + c] = b;
}

__global__ void colnorm_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                         const float* __restrict__ dy, const float* __restrict__ sums,
                                         float* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db,
                                         int accumulate, int64_t rows, int cols) {
  const int64_t n = rows * cols;
  const float inv_r = 1.f / (float)rows;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cols);
    const float rs = rstd[c], xh = (x[i] - mean[c]) * rs;
    if (dx) dx[i] = w[c] * rs * (dy[i] - sums[c] * inv_r - xh * (sums[cols + c] * inv_r));
    if (i < cols) {                         // the first row of threads also publishes the parameter gradients
      if (db) db[c] = accumulate ? db[c] + sums[c] : sums[c];
      if (dw) dw[c] = accumulate ? dw[c] + sums[cols + c] : sums[cols + c];
    }
  }
}

extern "C" int64_t pdn_colnorm_workspace_bytes(int64_t rows, int cols) {
  const int64_t chunks = (rows + CN_ROWS - 1) / CN_ROWS;
  return (chunks * 2 + 2) * (int64_t)cols * 4;
}

extern "C" int pdn_colnorm_fwd_f32(const float* x, const float* w, const float* b, float* y, float* mean,
                                   float* rstd, float* running_mean, float* running_var, float momentum,
                                   float eps, int64_t rows, int cols, void* workspace,
                                   int64_t workspace_bytes, void* stream) {
  PDN_CHECK_ARG(x && w && b && y && mean && rstd && rows > 0 && cols > 0, "pdn_colnorm_fwd_f32: bad arguments");
  if (!workspace || workspace_bytes < pdn_colnorm_workspace_bytes(rows, cols)) {
    pdn_set_error("pdn_colnorm_fwd_f32: workspace too small");
    return PDN_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int chunks = (int)((rows + CN_ROWS - 1) / CN_ROWS);
  PDN_CHECK_ARG(chunks <= 65535, "pdn_colnorm_fwd_f32: too many rows (%lld)", (long long)rows);
  float* part = (float*)workspace;
  hipLaunchKernelGGL(colnorm_stat_partial_kernel, dim3((cols + CN_COLS - 1) / CN_COLS, chunks), dim3(256), 0, st, x,
                     rows, cols, part);
  PDN_LAUNCH_CHECK();
  hipLaunchKernelGGL(colnorm_stat_finish_kernel, dim3((cols + 255) / 256), dim3(256), 0, st, part, rows, cols, chunks,
                     eps, momentum, mean, rstd, running_mean, running_var);
  PDN_LAUNCH_CHECK();
  hipLaunchKernelGGL(colnorm_apply_kernel, dim3(stream_grid(rows * cols)), dim3(256), 0, st, x, w, b, mean, rstd, y,
                     rows * cols, cols);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

// dx / dw / db are each optional; dw, db: (cols,), (+)= when accumulate.
extern "C" int pdn_colnorm_bwd_f32(const float* x, const float* w, const float* mean, const float* rstd,
                                   const float* dy, float* dx, float* dw, float* db, int accumulate,
                                   int64_t rows, int cols, void* workspace, int64_t workspace_bytes,
                                   void* stream) {
  PDN_CHECK_ARG(x && w && mean && rstd && dy && rows > 0 && cols > 0, "pdn_colnorm_bwd_f32: bad arguments");
  if (!workspace || workspace_bytes < pdn_colnorm_workspace_bytes(rows, cols)) {
    pdn_set_error("pdn_colnorm_bwd_f32: workspace too small");
    return PDN_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int chunks = (int)((rows + CN_ROWS - 1) / CN_ROWS);
  PDN_CHECK_ARG(chunks <= 65535, "pdn_colnorm_bwd_f32: too many rows (%lld)", (long long)rows);
  float* part = (float*)workspace;
  float* sums = part + (int64_t)chunks * 2 * cols;
  hipLaunchKernelGGL(colnorm_bwd_partial_kernel, dim3((cols + CN_COLS - 1) / CN_COLS, chunks), dim3(256), 0, st, x, dy,
                     mean, rstd, rows, cols, part);
  PDN_LAUNCH_CHECK();
  hipLaunchKernelGGL(colnorm_bwd_finish_kernel, dim3((cols + 255) / 256), dim3(256), 0, st, part, cols, chunks, sums);
  PDN_LAUNCH_CHECK();
  hipLaunchKernelGGL(colnorm_bwd_apply_kernel, dim3(stream_grid(rows * cols)), dim3(256), 0, st, x, w, mean, rstd, dy,
                     sums, dx, dw, db, accumulate, rows, cols);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}
