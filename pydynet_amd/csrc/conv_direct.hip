// Direct (implicit-GEMM) Conv2d for gfx950, fp32 MFMA: forward, data gradient, weight gradient.
//
// Replaces, for small-image / small-channel convolutions (LeNet-class: the whole zero-padded
// image fits in LDS), the reference's explicit path pydynet/nn/functional.py:254-281
//   pad -> as_strided(...).copy() [im2col, (N,C,kh,kw,oh,ow)] -> transpose+reshape copy -> GEMM
//   -> NHWC->NCHW view (+ bias), and np.add.at col2im in backward (:224-232).
// The explicit im2col moves ~10x the algorithmic bytes (9 copies of every input pixel, twice in
// forward, twice in backward).  Here the im2col matrix exists only as LDS ADDRESSES: the padded
// image is staged once per workgroup, every MFMA B-operand is one ds_read_b32 at
// image[c][oy*s + kh][ox*s + kw], the weights sit in LDS transposed so that A-operands are
// conflict-free, and the result leaves the accumulators straight into the NCHW output (bias added
// on the way).  HBM traffic = read x once + write y once (+ the weights once per workgroup).
//
// MFMA mapping (v_mfma_f32_32x32x2_f32, exact fp32): D[32 out-channels][32 positions] +=
// A[32 oc][2] * B[2][32 pos]; the contraction index runs tap-major / channel-minor so that the two
// half-waves take channels c and c+1 of one tap (channels padded to an even count with zero
// weights).  Any contraction order is legal: results differ from the reference's BLAS by summation
// order only (fp32 tolerance stated in tests/test_conv_direct_gpu.py).
//
//   conv_direct_kernel<OT, CH>   forward (mode 0) and data gradient (mode 1: input = dy, weights
//                                read flipped and with in/out channels swapped, pad = k-1-pad)
//   conv_wgrad_kernel<WT, PS>    dW[o][c][kh][kw] (+ db[o]) partial sums per workgroup over its
//                                images: A = dy[o][pos] from LDS, B = image gather with a per-lane
//                                column offset; contraction over positions; deterministic two-stage
//                                reduction (partials in a workspace, fixed combine order)
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvGeom {
  int N, Cin, Hin, Win, Cout, k, stride, pad, OH, OW;
  int Cp;      // Cin rounded up to even
  int PH, PW;  // padded image extent held in LDS
  int OPAD;    // Cout rounded up to 32
  int mode;    // 0 forward, 1 data gradient
  int wC, wO;  // dims of the weight tensor (O, C, k, k) as stored
};

__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// Stage one image (Cin, Hin, Win) into the interior of the zero-haloed LDS image.
__device__ __forceinline__ void stage_image(const float* __restrict__ src, float* __restrict__ img,
                                            const ConvGeom& g, int pad) {
  const int plane = g.Hin * g.Win, total = g.Cin * plane;
  if ((g.Win & 3) == 0) {
    for (int e = threadIdx.x * 4; e < total; e += blockDim.x * 4) {
      const float4 v = *reinterpret_cast<const float4*>(src + e);
      const int c = e / plane, rem = e - c * plane;
      const int y = rem / g.Win, x = rem - y * g.Win;
      float* d = img + (c * g.PH + y + pad) * g.PW + x + pad;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
  } else {
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
      const int c = e / plane, rem = e - c * plane;
      const int y = rem / g.Win, x = rem - y * g.Win;
      img[(c * g.PH + y + pad) * g.PW + x + pad] = src[e];
    }
  }
}

template <int OT, int CH>
__global__ __launch_bounds__(256) void conv_direct_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ w,
                                                           const float* __restrict__ bias,
                                                           float* __restrict__ y, ConvGeom g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int taps = g.k * g.k;
  float* wt = lds;                                   // [taps][Cp][OPAD]
  float* img = wt + taps * g.Cp * g.OPAD;            // [Cp][PH][PW]
  float* bs = img + g.Cp * g.PH * g.PW;              // [OPAD]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
  const int M = g.OH * g.OW, img_elems = g.Cp * g.PH * g.PW;

  // weights, transposed to [tap][cin][cout] (cout fastest: A-operand reads are conflict-free)
  for (int e = threadIdx.x; e < taps * g.Cp * g.OPAD; e += blockDim.x) {
    const int co = e % g.OPAD, t2 = e / g.OPAD, ci = t2 % g.Cp, tap = t2 / g.Cp;
    float v = 0.f;
    if (co < g.Cout && ci < g.Cin)
      v = g.mode == 0 ? w[((int64_t)co * g.wC + ci) * taps + tap]
                      : w[((int64_t)ci * g.wC + co) * taps + (taps - 1 - tap)];
    wt[e] = v;
  }
  for (int e = threadIdx.x; e < g.OPAD; e += blockDim.x) bs[e] = (bias && e < g.Cout) ? bias[e] : 0.f;
  for (int e = threadIdx.x; e < img_elems; e += blockDim.x) img[e] = 0.f;   // halo + padded channel stay 0
  __syncthreads();

  const int chunks = (M + 31) / 32, per_pass = 4 * CH, passes = (chunks + per_pass - 1) / per_pass;
  const int plane = g.PH * g.PW;
  for (int n = blockIdx.x; n < g.N; n += gridDim.x) {
    stage_image(x + (int64_t)n * g.Cin * g.Hin * g.Win, img, g, g.pad);
    __syncthreads();
    float* yn = y + (int64_t)n * g.Cout * M;
    for (int pass = 0; pass < passes; ++pass) {
      int poff[CH], pos[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int chunk = (pass * 4 + wave) * CH + c;
        pos[c] = chunk * 32 + l31;
        const int pc = pos[c] < M ? pos[c] : M - 1;
        const int oy = pc / g.OW, ox = pc - oy * g.OW;
        poff[c] = oy * g.stride * g.PW + ox * g.stride + half * plane;   // half-wave h reads channel c0 + h
      }
      if ((pass * 4 + wave) * CH * 32 >= M) continue;                    // wave-uniform: nothing to do
      f32x16 acc[OT][CH];
#pragma unroll
      for (int o = 0; o < OT; ++o)
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[o][c][r] = 0.f;
      const float* wrow = wt + half * g.OPAD + l31;
      for (int tap = 0; tap < taps; ++tap) {
        const int kh = tap / g.k, kw = tap - kh * g.k;
        const float* ib = img + kh * g.PW + kw;
        const float* wb = wrow + tap * g.Cp * g.OPAD;
        for (int c2 = 0; c2 < g.Cp; c2 += 2) {
          float a[OT], b[CH];
#pragma unroll
          for (int o = 0; o < OT; ++o) a[o] = wb[c2 * g.OPAD + o * 32];
#pragma unroll
          for (int c = 0; c < CH; ++c) b[c] = ib[c2 * plane + poff[c]];
#pragma unroll
          for (int o = 0; o < OT; ++o)
#pragma unroll
            for (int c = 0; c < CH; ++c)
              acc[o][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[o], b[c], acc[o][c], 0, 0, 0);
        }
      }
#pragma unroll
      for (int o = 0; o < OT; ++o)
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          if (pos[c] >= M) continue;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int oc = o * 32 + acc_row(r, half);
            if (oc < g.Cout) yn[(int64_t)oc * M + pos[c]] = acc[o][c][r] + bs[oc];
          }
        }
    }
    __syncthreads();                                  // everyone is done with this image
  }
}

struct WgradGeom {
  int N, C, H, W, O, k, stride, pad, OH, OW;
  int PH, PW, OPAD, K1, KCOLS;   // K1 = C*k*k + 1 (bias column), KCOLS = K1 rounded up to 32
  int DYS;                       // LDS row stride of the dy tile (odd)
  int per_block;                 // images per workgroup
};

// WT = tiles per wave; PS = 1: waves share all tiles and split the positions, 0: waves split tiles.
template <int WT, int PS>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ dy,
                                                          float* __restrict__ partial, WgradGeom g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* img = lds;                                   // [C][PH][PW]
  float* dyl = img + g.C * g.PH * g.PW;               // [OPAD][DYS]
  int* ptab = reinterpret_cast<int*>(dyl + g.OPAD * g.DYS);   // [M] image offset of each output position
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
  const int M = g.OH * g.OW, taps = g.k * g.k, KT = g.KCOLS / 32, T = (g.OPAD / 32) * KT;
  const int plane = g.PH * g.PW;

  for (int e = threadIdx.x; e < g.C * plane; e += blockDim.x) img[e] = 0.f;
  for (int e = threadIdx.x; e < g.OPAD * g.DYS; e += blockDim.x) dyl[e] = 0.f;
  for (int e = threadIdx.x; e < M; e += blockDim.x) {
    const int oy = e / g.OW, ox = e - oy * g.OW;
    ptab[e] = oy * g.stride * g.PW + ox * g.stride;
  }
  // per-lane column of every owned tile: j = c*taps + tap (the layout of the (O, C, k, k) weight)
  int coff[WT], row0[WT], kind[WT];                   // kind 0 = gather, 1 = ones (bias column), 2 = zero
#pragma unroll
  for (int i = 0; i < WT; ++i) {
    const int t = PS ? i : wave + 4 * i;
    const int ot = t / KT, kt = t - ot * KT;
    const int j = kt * 32 + l31;
    row0[i] = ot * 32;
    kind[i] = (t >= T || j >= g.K1) ? 2 : (j == g.K1 - 1 ? 1 : 0);
    const int c = j / taps, tap = j - c * taps, kh = tap / g.k, kw = tap - kh * g.k;
    coff[i] = kind[i] == 0 ? c * plane + kh * g.PW + kw : 0;
  }
  f32x16 acc[WT];
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  __syncthreads();

  const int n0 = blockIdx.x * g.per_block, n1 = min(g.N, n0 + g.per_block);
  const int pairs = (M + 1) / 2;
  for (int n = n0; n < n1; ++n) {
    // image into the padded LDS frame, dy rows into [o][pos]
    {
      ConvGeom cg;
      cg.Cin = g.C; cg.Hin = g.H; cg.Win = g.W; cg.PH = g.PH; cg.PW = g.PW;
      stage_image(x + (int64_t)n * g.C * g.H * g.W, img, cg, g.pad);
    }
    const float* dyn = dy + (int64_t)n * g.O * M;
    if ((M & 3) == 0) {
      for (int e = threadIdx.x * 4; e < g.O * M; e += blockDim.x * 4) {
        const float4 v = *reinterpret_cast<const float4*>(dyn + e);
        const int o = e / M, p = e - o * M;
        float* d = dyl + o * g.DYS + p;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      }
    } else {
      for (int e = threadIdx.x; e < g.O * M; e += blockDim.x) {
        const int o = e / M, p = e - o * M;
        dyl[o * g.DYS + p] = dyn[e];
      }
    }
    __syncthreads();
    for (int pp = PS ? wave : 0; pp < pairs; pp += PS ? 4 : 1) {
      const int pos = 2 * pp + half;
      const bool valid = pos < M;
      const int pc = valid ? pos : M - 1;
      const int po = ptab[pc];
#pragma unroll
      for (int i = 0; i < WT; ++i) {
        float a = dyl[(row0[i] + l31) * g.DYS + pc];
        float b = kind[i] == 0 ? img[coff[i] + po] : (kind[i] == 1 ? 1.f : 0.f);
        a = valid ? a : 0.f;
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // partial sums: slab per (workgroup[, wave]); layout [OPAD][KCOLS]
  const int slab = PS ? blockIdx.x * 4 + wave : blockIdx.x;
  float* out = partial + (int64_t)slab * g.OPAD * g.KCOLS;
#pragma unroll
  for (int i = 0; i < WT; ++i) {
    const int t = PS ? i : wave + 4 * i;
    if (t >= T) continue;
    const int ot = t / KT, kt = t - ot * KT;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      out[(int64_t)(ot * 32 + acc_row(r, half)) * g.KCOLS + kt * 32 + l31] = acc[i][r];
  }
}

// dw[o][j] (+)= sum_slabs partial[slab][o][j] (j < K), db[o] (+)= column K; fixed order: deterministic
__global__ void conv_wgrad_reduce_kernel(const float* __restrict__ partial, int slabs, int OPAD, int KCOLS,
                                         int O, int K, float* __restrict__ dw, float* __restrict__ db,
                                         int accumulate) {
  const int total = O * (K + 1);
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int o = e / (K + 1), j = e - o * (K + 1);
    const float* p = partial + (int64_t)o * KCOLS + j;
    float s = 0.f;
    for (int b = 0; b < slabs; ++b) s += p[(int64_t)b * OPAD * KCOLS];
    if (j < K) {
      if (dw) dw[(int64_t)o * K + j] = accumulate ? dw[(int64_t)o * K + j] + s : s;
    } else if (db) {
      db[o] = accumulate ? db[o] + s : s;
    }
  }
}

namespace {
const int kMaxLds = 150 * 1024;

bool fwd_geom(ConvGeom& g, int N, int Cin, int Hin, int Win, int Cout, int k, int stride, int pad, int mode,
              int wC, int wO) {
  g.N = N; g.Cin = Cin; g.Hin = Hin; g.Win = Win; g.Cout = Cout; g.k = k; g.stride = stride; g.pad = pad;
  g.OH = (Hin + 2 * pad - k) / stride + 1;
  g.OW = (Win + 2 * pad - k) / stride + 1;
  g.Cp = (Cin + 1) & ~1;
  g.PH = Hin + 2 * pad; g.PW = Win + 2 * pad;
  g.OPAD = (Cout + 31) / 32 * 32;
  g.mode = mode; g.wC = wC; g.wO = wO;
  return g.OH > 0 && g.OW > 0;
}
int64_t fwd_lds(const ConvGeom& g) {
  return 4ll * ((int64_t)g.k * g.k * g.Cp * g.OPAD + (int64_t)g.Cp * g.PH * g.PW + g.OPAD) + 64;
}
bool fwd_ok(const ConvGeom& g) { return g.OPAD <= 64 && g.pad >= 0 && fwd_lds(g) <= kMaxLds; }

int launch_direct(const float* x, const float* w, const float* bias, float* y, const ConvGeom& g, hipStream_t st) {
  const int64_t lds = fwd_lds(g);
  const int per_cu = (int)(kMaxLds / lds) < 1 ? 1 : (int)((160 * 1024) / lds);
  int grid = 256 * (per_cu > 4 ? 4 : per_cu);
  if (grid > g.N) grid = g.N;
  const int M = g.OH * g.OW, chunks = (M + 31) / 32;
#define PDN_CONV_LAUNCH(OT, CH)                                                                        \
  do {                                                                                                 \
    auto kern = conv_direct_kernel<OT, CH>;                                                            \
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                       (int)lds);                                                      \
    if (e != hipSuccess) { pdn_set_error("conv: LDS attribute: %s", hipGetErrorString(e)); return (int)e; } \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, x, w, bias, y, g);                        \
  } while (0)
  if (g.OPAD == 64) {
    if (chunks >= 8) PDN_CONV_LAUNCH(2, 2); else PDN_CONV_LAUNCH(2, 1);
  } else {
    if (chunks >= 16) PDN_CONV_LAUNCH(1, 4); else if (chunks >= 8) PDN_CONV_LAUNCH(1, 2); else PDN_CONV_LAUNCH(1, 1);
  }
#undef PDN_CONV_LAUNCH
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

bool wgrad_geom(WgradGeom& g, int N, int C, int H, int W, int O, int k, int stride, int pad) {
  g.N = N; g.C = C; g.H = H; g.W = W; g.O = O; g.k = k; g.stride = stride; g.pad = pad;
  g.OH = (H + 2 * pad - k) / stride + 1;
  g.OW = (W + 2 * pad - k) / stride + 1;
  g.PH = H + 2 * pad; g.PW = W + 2 * pad;
  g.OPAD = (O + 31) / 32 * 32;
  g.K1 = C * k * k + 1;
  g.KCOLS = (g.K1 + 31) / 32 * 32;
  g.DYS = (g.OH * g.OW) | 1;
  return g.OH > 0 && g.OW > 0;
}
int64_t wgrad_lds(const WgradGeom& g) {
  return 4ll * ((int64_t)g.C * g.PH * g.PW + (int64_t)g.OPAD * g.DYS + (int64_t)g.OH * g.OW) + 64;
}
int wgrad_tiles(const WgradGeom& g) { return (g.OPAD / 32) * (g.KCOLS / 32); }
bool wgrad_ok(const WgradGeom& g) { return wgrad_tiles(g) <= 16 && wgrad_lds(g) <= kMaxLds; }
int wgrad_blocks(const WgradGeom& g) {
  const int64_t lds = wgrad_lds(g);
  int per_cu = (int)((160 * 1024) / lds);
  if (per_cu > 2) per_cu = 2;
  int blocks = 256 * per_cu;
  if (blocks > g.N) blocks = g.N;
  return blocks;
}
}  // namespace

extern "C" {

/* bitmask of the directions the direct kernels serve for this shape: 1 forward, 2 data gradient,
 * 4 weight gradient (0 = use the im2col + GEMM path) */
int pdn_conv2d_direct_supported(int C, int H, int W, int O, int k, int stride, int pad) {
  if (C <= 0 || O <= 0 || k <= 0 || stride <= 0 || pad < 0 || H + 2 * pad < k || W + 2 * pad < k) return 0;
  int mask = 0;
  ConvGeom f;
  if (fwd_geom(f, 1, C, H, W, O, k, stride, pad, 0, C, O) && fwd_ok(f)) mask |= 1;
  if (stride == 1 && k - 1 - pad >= 0) {
    ConvGeom d;
    const int OH = H + 2 * pad - k + 1, OW = W + 2 * pad - k + 1;
    if (fwd_geom(d, 1, O, OH, OW, C, k, 1, k - 1 - pad, 1, C, O) && fwd_ok(d) && d.OH == H && d.OW == W) mask |= 2;
  }
  WgradGeom g;
  if (wgrad_geom(g, 1, C, H, W, O, k, stride, pad) && wgrad_ok(g)) mask |= 4;
  return mask;
}

/* y (N, O, OH, OW) = conv(x (N, C, H, W), w (O, C, k, k)) + bias[O]   (nn/functional.py:254-281) */
int pdn_conv2d_fwd_f32(const float* x, const float* w, const float* bias, float* y, int N, int C, int H,
                       int W, int O, int k, int stride, int pad, void* stream) {
  if (N == 0) return PDN_OK;
  PDN_CHECK_ARG(x && w && y, "pdn_conv2d_fwd_f32: null operand");
  ConvGeom g;
  if (!fwd_geom(g, N, C, H, W, O, k, stride, pad, 0, C, O) || !fwd_ok(g)) {
    pdn_set_error("pdn_conv2d_fwd_f32: shape outside the direct kernel (use im2col + GEMM)");
    return PDN_EUNSUPPORTED;
  }
  return launch_direct(x, w, bias, y, g, (hipStream_t)stream);
}

/* dx (N, C, H, W) = conv_transpose(dy (N, O, OH, OW), w): stride 1 only */
int pdn_conv2d_bwd_data_f32(const float* dy, const float* w, float* dx, int N, int C, int H, int W, int O,
                            int k, int stride, int pad, void* stream) {
  if (N == 0) return PDN_OK;
  PDN_CHECK_ARG(dy && w && dx, "pdn_conv2d_bwd_data_f32: null operand");
  const int OH = H + 2 * pad - k + 1, OW = W + 2 * pad - k + 1;
  ConvGeom g;
  if (stride != 1 || k - 1 - pad < 0 || !fwd_geom(g, N, O, OH, OW, C, k, 1, k - 1 - pad, 1, C, O) || !fwd_ok(g) ||
      g.OH != H || g.OW != W) {
    pdn_set_error("pdn_conv2d_bwd_data_f32: shape outside the direct kernel (use GEMM + col2im)");
    return PDN_EUNSUPPORTED;
  }
  return launch_direct(dy, w, nullptr, dx, g, (hipStream_t)stream);
}

int64_t pdn_conv2d_bwd_weight_workspace_bytes(int N, int C, int H, int W, int O, int k, int stride, int pad) {
  WgradGeom g;
  if (!wgrad_geom(g, N, C, H, W, O, k, stride, pad) || !wgrad_ok(g)) return 0;
  return 4ll * wgrad_blocks(g) * 4 * g.OPAD * g.KCOLS;
}

/* dw (O, C, k, k) and db (O) [either may be NULL]; accumulate != 0 adds into them */
int pdn_conv2d_bwd_weight_f32(const float* x, const float* dy, float* dw, float* db, int accumulate, int N,
                              int C, int H, int W, int O, int k, int stride, int pad, void* workspace,
                              int64_t workspace_bytes, void* stream) {
  if (N == 0) return PDN_OK;
  PDN_CHECK_ARG(x && dy && (dw || db), "pdn_conv2d_bwd_weight_f32: null operand");
  WgradGeom g;
  if (!wgrad_geom(g, N, C, H, W, O, k, stride, pad) || !wgrad_ok(g)) {
    pdn_set_error("pdn_conv2d_bwd_weight_f32: shape outside the direct kernel (use the GEMM path)");
    return PDN_EUNSUPPORTED;
  }
  const int blocks = wgrad_blocks(g);
  g.per_block = (N + blocks - 1) / blocks;
  const int used = (N + g.per_block - 1) / g.per_block;
  const int T = wgrad_tiles(g);
  const int ps = T < 4 ? 1 : 0;
  const int slabs = used * (ps ? 4 : 1);
  if (!workspace || workspace_bytes < 4ll * slabs * g.OPAD * g.KCOLS) {
    pdn_set_error("pdn_conv2d_bwd_weight_f32: workspace too small");
    return PDN_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int64_t lds = wgrad_lds(g);
  float* partial = (float*)workspace;
#define PDN_WGRAD_LAUNCH(WT, PS)                                                                       \
  do {                                                                                                 \
    auto kern = conv_wgrad_kernel<WT, PS>;                                                             \
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                       (int)lds);                                                      \
    if (e != hipSuccess) { pdn_set_error("conv: LDS attribute: %s", hipGetErrorString(e)); return (int)e; } \
    hipLaunchKernelGGL(kern, dim3(used), dim3(256), lds, st, x, dy, partial, g);                       \
  } while (0)
  if (ps) {
    if (T == 1) PDN_WGRAD_LAUNCH(1, 1); else if (T == 2) PDN_WGRAD_LAUNCH(2, 1); else PDN_WGRAD_LAUNCH(3, 1);
  } else {
    const int wt = (T + 3) / 4;
    if (wt == 1) PDN_WGRAD_LAUNCH(1, 0); else if (wt == 2) PDN_WGRAD_LAUNCH(2, 0);
    else if (wt == 3) PDN_WGRAD_LAUNCH(3, 0); else PDN_WGRAD_LAUNCH(4, 0);
  }
#undef PDN_WGRAD_LAUNCH
  PDN_LAUNCH_CHECK();
  const int total = O * g.K1;
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, st, partial, slabs,
                     g.OPAD, g.KCOLS, O, g.K1 - 1, dw, db, accumulate);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

}  // extern "C"
