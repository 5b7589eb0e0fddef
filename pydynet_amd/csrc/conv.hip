// im2col / col2im / pooling kernels for the Conv2d path (gfx950), fp32.
//
// Reference: pydynet/nn/functional.py:194-339.  The reference pads (xp.pad, :241-245), builds a
// strided window view and copies it (`as_strided(...).copy()`, :211-222) into the layout
// (N, C, kh, kw, oh, ow); the backward is `xp.add.at` on the overlapping view (:224-232).
// Here padding is folded into the gather (values are identical: bit-exact `col`), and the
// backward is a GATHER per input pixel over the <= k*k windows that cover it, so it needs no
// atomics and is deterministic.  These are pure HBM streams: every lane walks the fastest
// (ow / W) index so loads and stores coalesce.
#include "common.h"

// col[n][c][i][j][oy][ox] = x[n][c][oy*s + i - p][ox*s + j - p]   (0 outside)
// One workgroup per (image, channel): it writes the k*k rows of that channel (consecutive threads ->
// consecutive output pixels, so both the shifted-window reads and the writes are coalesced; the
// 4 KB input plane is served by L1).  Rows >= C*k*k pad the contraction to `rows` (a multiple of 4
// for the GEMM's 16-byte path) and are written by the extra workgroup c == C; row C*k*k holds ones
// when `ones_row` (it pairs with a bias column in the packed weight, so the bias and its gradient
// ride inside the GEMMs).
__global__ void im2col2d_kernel(const float* __restrict__ x, float* __restrict__ col, int C, int H,
                                int W, int k, int s, int p, int oh, int ow, int rows, int ones_row) {
  const int M = oh * ow, c = blockIdx.x, n = blockIdx.y, kk = k * k, ckk = C * kk;
  if (c == C) {                                   // padding rows
    float* dst = col + ((int64_t)n * rows + ckk) * M;
    const int total = (rows - ckk) * M;
    for (int e = threadIdx.x; e < total; e += blockDim.x) dst[e] = (ones_row && e < M) ? 1.f : 0.f;
    return;
  }
  const float* src = x + ((int64_t)n * C + c) * H * W;
  float* dst = col + ((int64_t)n * rows + (int64_t)c * kk) * M;
  const int total = kk * M;
  for (int e = threadIdx.x; e < total; e += blockDim.x) {
    const int r = e / M, m = e - r * M;
    const int i = r / k, j = r - i * k;
    const int oy = m / ow, ox = m - oy * ow;
    const int y = oy * s + i - p, xx = ox * s + j - p;
    dst[e] = (y >= 0 && y < H && xx >= 0 && xx < W) ? src[y * W + xx] : 0.f;
  }
}

// dx[n][c][y][x] = sum over (i,j) with (y+p-i) % s == 0, (x+p-j) % s == 0 of dcol[n][c][i][j][oy][ox]
__global__ void col2im2d_kernel(const float* __restrict__ dcol, float* __restrict__ dx, int N, int C,
                                int H, int W, int k, int s, int p, int oh, int ow, int rows) {
  const int64_t total = (int64_t)N * C * H * W;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t t = idx;
    const int xx = (int)(t % W); t /= W;
    const int y = (int)(t % H); t /= H;      // t = n*C + c
    const int n = (int)(t / C), c = (int)(t - (int64_t)n * C);
    const float* base = dcol + ((int64_t)n * rows + (int64_t)c * k * k) * oh * ow;
    float acc = 0.f;
    for (int i = 0; i < k; ++i) {
      const int ny = y + p - i;
      if (ny < 0 || ny % s) continue;
      const int oy = ny / s;
      if (oy >= oh) continue;
      for (int j = 0; j < k; ++j) {
        const int nx = xx + p - j;
        if (nx < 0 || nx % s) continue;
        const int ox = nx / s;
        if (ox >= ow) continue;
        acc += base[((int64_t)(i * k + j) * oh + oy) * ow + ox];
      }
    }
    dx[idx] = acc;
  }
}

// Pooling over k x k windows of the zero-padded input (the pad value 0 takes part in max,
// exactly as in the reference where pooling runs on the padded array).  mode 0 = max, 1 = avg.
// Output is NCHW contiguous.
__global__ void pool2d_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int NC, int H,
                                  int W, int k, int s, int p, int oh, int ow, int mode) {
  const int64_t total = (int64_t)NC * oh * ow;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t t = idx;
    const int ox = (int)(t % ow); t /= ow;
    const int oy = (int)(t % oh); t /= oh;
    const float* xp = x + t * (int64_t)H * W;
    float acc = mode == 0 ? -INFINITY : 0.f;
    for (int i = 0; i < k; ++i)
      for (int j = 0; j < k; ++j) {
        const int yy = oy * s + i - p, xx = ox * s + j - p;
        const float v = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? xp[yy * W + xx] : 0.f;
        acc = mode == 0 ? fmaxf(acc, v) : acc + v;
      }
    y[idx] = mode == 0 ? acc : acc / (float)(k * k);
  }
}

// max: every position equal to its window's max receives that window's gradient (ties all get
// it: tensor.py:744-750); avg: g / k^2.  Gather formulation over the windows covering a pixel.
__global__ void pool2d_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                  const float* __restrict__ dy, float* __restrict__ dx, int NC, int H,
                                  int W, int k, int s, int p, int oh, int ow, int mode) {
  // one workgroup per (n, c) plane: 32-bit index math only
  const int HW = H * W, OM = oh * ow;
  for (int64_t plane = blockIdx.x; plane < NC; plane += gridDim.x) {
    const float* xp = x + plane * HW;
    const float* yp = y + plane * OM;
    const float* gp = dy + plane * OM;
    float* dp = dx + plane * HW;
    for (int e = threadIdx.x; e < HW; e += blockDim.x) {
      const int yy = e / W, xx = e - yy * W;
      const float v = xp[e];
      float acc = 0.f;
      if (k == s) {                                // non-overlapping windows: exactly one covers a pixel
        const int oy = (yy + p) / k, ox = (xx + p) / k;
        if (oy < oh && ox < ow) {
          const float g = gp[oy * ow + ox];
          acc = mode == 0 ? ((yp[oy * ow + ox] == v) ? g : 0.f) : g / (float)(k * k);
        }
      } else {
        for (int i = 0; i < k; ++i) {
          const int ny = yy + p - i;
          if (ny < 0 || ny % s) continue;
          const int oy = ny / s;
          if (oy >= oh) continue;
          for (int j = 0; j < k; ++j) {
            const int nx = xx + p - j;
            if (nx < 0 || nx % s) continue;
            const int ox = nx / s;
            if (ox >= ow) continue;
            const float g = gp[oy * ow + ox];
            if (mode == 0) acc += (yp[oy * ow + ox] == v) ? g : 0.f;
            else acc += g / (float)(k * k);
          }
        }
      }
      dp[e] = acc;
    }
  }
}

static inline int grid1d(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (int)b;
}

#define CONV_ARGS_OK(name)                                                                   \
  PDN_CHECK_ARG(N >= 0 && C > 0 && H > 0 && W > 0 && k > 0 && stride > 0 && pad >= 0, name ": bad dims"); \
  const int oh = (H + 2 * pad - k) / stride + 1, ow = (W + 2 * pad - k) / stride + 1;         \
  PDN_CHECK_ARG(oh > 0 && ow > 0, name ": kernel larger than padded input");

// col: (N, col_rows, oh*ow) with col_rows >= C*k*k (+1 when ones_row); see im2col2d_kernel.
extern "C" int pdn_im2col2d_f32(const float* x, int N, int C, int H, int W, int k, int stride,
                                int pad, float* col, int col_rows, int ones_row, void* stream) {
  CONV_ARGS_OK("pdn_im2col2d_f32")
  if (N == 0) return PDN_OK;
  PDN_CHECK_ARG(x && col, "pdn_im2col2d_f32: null operand");
  PDN_CHECK_ARG(col_rows >= C * k * k + (ones_row ? 1 : 0) && col_rows <= 65535 && N <= 65535,
                "pdn_im2col2d_f32: col_rows=%d too small for C*k*k=%d (or grid limit)", col_rows, C * k * k);
  PDN_CHECK_ARG((int64_t)k * k * oh * ow < (1ll << 31), "pdn_im2col2d_f32: plane too large");
  hipLaunchKernelGGL(im2col2d_kernel, dim3(C + (col_rows > C * k * k ? 1 : 0), N), dim3(256), 0,
                     (hipStream_t)stream, x, col, C, H, W, k, stride, pad, oh, ow, col_rows, ones_row);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

extern "C" int pdn_col2im2d_f32(const float* dcol, int N, int C, int H, int W, int k, int stride,
                                int pad, float* dx, int col_rows, void* stream) {
  CONV_ARGS_OK("pdn_col2im2d_f32")
  if (N == 0) return PDN_OK;
  PDN_CHECK_ARG(dcol && dx, "pdn_col2im2d_f32: null operand");
  PDN_CHECK_ARG(col_rows >= C * k * k, "pdn_col2im2d_f32: col_rows=%d < C*k*k=%d", col_rows, C * k * k);
  hipLaunchKernelGGL(col2im2d_kernel, dim3(grid1d((int64_t)N * C * H * W)), dim3(256), 0,
                     (hipStream_t)stream, dcol, dx, N, C, H, W, k, stride, pad, oh, ow, col_rows);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

extern "C" int pdn_pool2d_fwd_f32(const float* x, int N, int C, int H, int W, int k, int stride,
                                  int pad, int mode, float* y, void* stream) {
  CONV_ARGS_OK("pdn_pool2d_fwd_f32")
  if (N == 0) return PDN_OK;
  PDN_CHECK_ARG(x && y && (mode == 0 || mode == 1), "pdn_pool2d_fwd_f32: bad arguments");
  hipLaunchKernelGGL(pool2d_fwd_kernel, dim3(grid1d((int64_t)N * C * oh * ow)), dim3(256), 0,
                     (hipStream_t)stream, x, y, N * C, H, W, k, stride, pad, oh, ow, mode);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

extern "C" int pdn_pool2d_bwd_f32(const float* x, const float* y, const float* dy, int N, int C,
                                  int H, int W, int k, int stride, int pad, int mode, float* dx,
                                  void* stream) {
  CONV_ARGS_OK("pdn_pool2d_bwd_f32")
  if (N == 0) return PDN_OK;
  PDN_CHECK_ARG(x && y && dy && dx && (mode == 0 || mode == 1), "pdn_pool2d_bwd_f32: bad arguments");
  PDN_CHECK_ARG((int64_t)H * W < (1ll << 31), "pdn_pool2d_bwd_f32: plane too large");
  const int64_t planes = (int64_t)N * C;
  hipLaunchKernelGGL(pool2d_bwd_kernel, dim3((unsigned)(planes < (1 << 20) ? planes : (1 << 20))), dim3(256), 0,
                     (hipStream_t)stream, x, y, dy, dx, N * C, H, W, k, stride, pad, oh, ow, mode);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}
