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
//                                EP = 1: bias + ReLU + 2x2 / stride-2 max-pool in the epilogue (the conv ->
//                                relu -> max_pool chain of examples/pydynet/mnist.py:92-95): only the pooled map
//                                and a 4-bit mask per pooled element (which window positions receive the
//                                gradient) are written; SRC = 1: the input image is the EXPANSION of a pooled
//                                gradient through such a mask, formed while it is staged into LDS -- the
//                                full-resolution conv output and its gradient never exist in HBM
//   conv_wgrad_kernel<WT, PS>    dW[o][c][kh][kw] (+ db[o]) partial sums per workgroup over its
//                                images: A = dy[o][pos] from LDS, B = image gather with a per-lane
//                                column offset; contraction over positions; deterministic two-stage
//                                reduction (partials in a workspace, fixed combine order)
#include "common.h"
#include "conv_quad.h"
#include <type_traits>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Division by a launch-time constant without the ~40-instruction integer-division sequence: the staging code turns a
// flat element index into (channel, row, column) several times per element, and in the first version those divisions
// were most of what the kernels executed (counters: 14 VALU instructions per LDS instruction).  Powers of two are a
// shift; anything else is one multiply-high + shift (exact for 0 <= x < 2^31).
struct FastDiv {
  unsigned mul;                 // 0: shift only
  int sh;
};
static inline FastDiv make_fastdiv(int d) {
  FastDiv f{0u, 0};
  if (d <= 1) return f;
  int s = 0;
  while ((1 << (s + 1)) <= d) ++s;              // floor(log2 d)
  f.sh = s;
  if ((d & (d - 1)) != 0) f.mul = (unsigned)(((1ull << (32 + s)) + (unsigned)d - 1) / (unsigned)d);
  return f;
}
__device__ __forceinline__ int fdiv(int x, const FastDiv& f) {
  return f.mul ? (int)(__umulhi((unsigned)x, f.mul) >> f.sh) : (x >> f.sh);
}

struct ConvGeom {
  FastDiv fw, fplane, fow, fopad, fcp;   // Win, Hin * Win, OW, OPAD, Cp
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
      const int c = fdiv(e, g.fplane), rem = e - c * plane;
      const int y = fdiv(rem, g.fw), x = rem - y * g.Win;
      float* d = img + (c * g.PH + y + pad) * g.PW + x + pad;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
  } else {
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
      const int c = fdiv(e, g.fplane), rem = e - c * plane;
      const int y = fdiv(rem, g.fw), x = rem - y * g.Win;
      img[(c * g.PH + y + pad) * g.PW + x + pad] = src[e];
    }
  }
}

// Register prefetch of a (C, H, W) block whose rows are float4-aligned: `issue` starts the global loads
// (NV float4 per thread cover 256 * NV * 4 floats), `commit` writes them into the padded LDS frame.
// Issued one image ahead of the MFMA phase, the HBM latency hides behind compute instead of in front.
__device__ __forceinline__ void lds_store4(float* __restrict__ d, const float4& v) {
  // (rotating the component order per lane group to dodge the 4-way bank conflict of these stores was
  // measured SLOWER: the selects cost more than the conflicts -- the stores are not on the critical path)
  d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
}

// HO: the (c, y, x) -> frame offset of every piece a thread stages is the same for every image: worked out once
// (`init`) and kept in registers instead of redone per image (kernels with registers to spare).
template <int NV, bool HO = false>
struct TilePrefetch {
  float4 v[NV > 0 ? NV : 1];
  int foff[(HO && NV > 0) ? NV : 1];                  // frame offset of piece i, -1: not this thread's
  __device__ __forceinline__ void init(int total, int H, int W, int PH, int PW, int pad, const FastDiv& fplane,
                                       const FastDiv& fw) {
    if constexpr (HO) {
      const int plane = H * W;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int e = (threadIdx.x + i * 256) * 4;
        const int c = fdiv(e, fplane), rem = e - c * plane;
        const int y = fdiv(rem, fw), x = rem - y * W;
        foff[i] = e < total ? (c * PH + y + pad) * PW + x + pad : -1;
      }
    }
  }
  __device__ __forceinline__ void issue(const float* __restrict__ src, int total) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      // (unconditional: a load under `if (e < total)` is followed by a copy of its registers -- the phi with the
      //  untaken path -- that WAITS for it, and the "prefetch" becomes NV serialised round trips; threads past the
      //  end re-read piece 0 and never commit it)
      const int e = (threadIdx.x + i * 256) * 4;
      v[i] = *reinterpret_cast<const float4*>(src + (e < total ? e : 0));
    }
  }
  // (c, y, x) of element e in a (C, H, W) block -> frame[(c * PH + y + pad) * PW + x + pad]
  __device__ __forceinline__ void commit_image(float* __restrict__ frame, int total, int H, int W, int PH, int PW,
                                               int pad, const FastDiv& fplane, const FastDiv& fw) {
    const int plane = H * W;
    if constexpr (HO) {
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if (foff[i] >= 0) lds_store4(frame + foff[i], v[i]);
      return;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = (threadIdx.x + i * 256) * 4;
      if (e < total) {
        const int c = fdiv(e, fplane), rem = e - c * plane;
        const int y = fdiv(rem, fw), x = rem - y * W;
        lds_store4(frame + (c * PH + y + pad) * PW + x + pad, v[i]);
      }
    }
  }
};

// The same block read through a 2x2 max-pool hit map: element (c, y, x) of the (C, H, W) block is
//   dp[c][y / 2][x / 2]  if bit ((y * W + x) & 31) of hit[c][(y * W + x) >> 5] is set, else 0
// (reference semantics of relu -> max_pool backward, tensor.py:808-815 + functional.py:284-339: every position
// that equals the window maximum AND passes relu'(y) = [y >= 0] receives the pooled gradient; one bit per conv
// output position, 32 positions per word = one ballot of the forward epilogue).  A float4 piece = two pooled
// elements: one float2 + one hit word per piece are what travels from HBM.
__device__ __forceinline__ float4 expand_pooled(const float2& v, unsigned bits) {
  return make_float4((bits & 1u) ? v.x : 0.f, (bits & 2u) ? v.x : 0.f, (bits & 4u) ? v.y : 0.f, (bits & 8u) ? v.y : 0.f);
}

template <int NV, bool HO = false>
struct PooledPrefetch {
  float2 v[NV > 0 ? NV : 1];
  unsigned m[NV > 0 ? NV : 1];
  int foff[(HO && NV > 0) ? NV : 1], soff[(HO && NV > 0) ? NV : 1];   // frame offset (-1: none) / pooled source offset
  __device__ __forceinline__ void init(int total, int H, int W, int PH, int PW, int pad, const FastDiv& fplane,
                                       const FastDiv& fw) {
    if constexpr (HO) {
      const int plane = H * W, hw = W >> 1;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int e0 = (threadIdx.x + i * 256) * 4, e = e0 < total ? e0 : 0;
        const int c = fdiv(e, fplane), rem = e - c * plane;
        const int y = fdiv(rem, fw), x = rem - y * W;
        soff[i] = (c * (H >> 1) + (y >> 1)) * hw + (x >> 1);
        foff[i] = e0 < total ? (c * PH + y + pad) * PW + x + pad : -1;
      }
    }
  }
  // dp of ONE image: (C, H / 2, W / 2); hit words of one image: (C, H * W / 32); piece e covers (c, y, x .. x + 3)
  __device__ __forceinline__ void issue(const float* __restrict__ dp, const unsigned* __restrict__ hit, int total,
                                        int H, int W, const FastDiv& fplane, const FastDiv& fw) {
    const int plane = H * W, hw = W >> 1;
    if constexpr (HO) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int e0 = (threadIdx.x + i * 256) * 4, e = e0 < total ? e0 : 0;
        v[i] = *reinterpret_cast<const float2*>(dp + (unsigned)soff[i]);
        m[i] = hit[(unsigned)(e >> 5)];
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e0 = (threadIdx.x + i * 256) * 4, e = e0 < total ? e0 : 0;      // (unconditional loads: see TilePrefetch)
      const int c = fdiv(e, fplane), rem = e - c * plane;
      const int y = fdiv(rem, fw), x = rem - y * W;
      v[i] = *reinterpret_cast<const float2*>(dp + (c * (H >> 1) + (y >> 1)) * hw + (x >> 1));
      m[i] = hit[e >> 5];
    }
  }
  __device__ __forceinline__ void commit_image(float* __restrict__ frame, int total, int H, int W, int PH, int PW,
                                               int pad, const FastDiv& fplane, const FastDiv& fw) {
    if constexpr (HO) {
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if (foff[i] >= 0)
          lds_store4(frame + foff[i], expand_pooled(v[i], (m[i] >> (((threadIdx.x + i * 256) * 4) & 31)) & 15u));
      return;
    }
    const int plane = H * W;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = (threadIdx.x + i * 256) * 4;
      if (e < total) {
        const int c = fdiv(e, fplane), rem = e - c * plane;
        const int y = fdiv(rem, fw), x = rem - y * W;
        lds_store4(frame + (c * PH + y + pad) * PW + x + pad, expand_pooled(v[i], (m[i] >> (e & 31)) & 15u));
      }
    }
  }
};

// KS = compile-time tap extent (1, 3, 5) or 0 for a runtime extent.  With KS known the taps*(OT+CH)
// LDS reads of one channel pair are issued as a block ahead of the MFMAs that consume them, and the
// block of the NEXT channel pair is in flight while the current one is multiplied (one wave per SIMD
// has no partner to hide the ~100-cycle ds_read latency behind, so the prefetch is explicit).
template <int OT, int CH, int KS, int NV, int EP = 0, int SRC = 0>
__global__ __launch_bounds__(256, 2) void conv_direct_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ w,
                                                           const float* __restrict__ bias,
                                                           float* __restrict__ y, ConvGeom g,
                                                           unsigned* __restrict__ pmask) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int taps = g.k * g.k;
  float* wt = lds;                                   // [taps][Cp][OPAD]
  float* img = wt + taps * g.Cp * g.OPAD;            // [Cp][PH][PW]
  float* bs = img + g.Cp * g.PH * g.PW;              // [OPAD]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
  const int M = g.OH * g.OW, img_elems = g.Cp * g.PH * g.PW;

  // weights, transposed to [tap][cin][cout] (cout fastest: A-operand reads are conflict-free)
  for (int e = threadIdx.x; e < taps * g.Cp * g.OPAD; e += blockDim.x) {
    const int t2 = fdiv(e, g.fopad), co = e - t2 * g.OPAD, tap = fdiv(t2, g.fcp), ci = t2 - tap * g.Cp;
    float v = 0.f;
    if (co < g.Cout && ci < g.Cin)
      v = g.mode == 0 ? w[((int64_t)co * g.wC + ci) * taps + tap]
                      : w[((int64_t)ci * g.wC + co) * taps + (taps - 1 - tap)];
    wt[e] = v;
  }
  for (int e = threadIdx.x; e < g.OPAD; e += blockDim.x) bs[e] = (bias && e < g.Cout) ? bias[e] : 0.f;
  for (int e = threadIdx.x; e < img_elems; e += blockDim.x) img[e] = 0.f;   // halo + padded channel stay 0
  __syncthreads();

  // (the bias is re-read from LDS in the epilogue -- sixteen reads per stored tile: keeping it in registers cost the
  //  data-gradient form, which has none, sixteen registers it needs for its prefetch)
  unsigned rowmask[OT];
#pragma unroll
  for (int o = 0; o < OT; ++o) {
    rowmask[o] = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int oc = o * 32 + acc_row(r, half);
      rowmask[o] |= (oc < g.Cout ? 1u : 0u) << r;
    }
  }
  const int chunks = (M + 31) / 32, per_pass = 4 * CH, passes = (chunks + per_pass - 1) / per_pass;
  const int plane = g.PH * g.PW, npair = g.Cp / 2, wstep = 2 * g.OPAD, tapw = g.Cp * g.OPAD;
  const int in_elems = g.Cin * g.Hin * g.Win;
  constexpr bool HO = NV > 0 && (NV <= 8 || OT * CH <= 2);   // staging offsets kept in registers where there is room
  typename std::conditional<SRC == 1, PooledPrefetch<NV, HO>, TilePrefetch<NV, HO>>::type pf;
  pf.init(in_elems, g.Hin, g.Win, g.PH, g.PW, g.pad, g.fplane, g.fw);
  auto pf_issue = [&](int n) {
    if constexpr (SRC == 1)
      pf.issue(x + (int64_t)n * (in_elems >> 2), pmask + (int64_t)n * (in_elems >> 5), in_elems, g.Hin, g.Win, g.fplane, g.fw);
    else
      pf.issue(x + (int64_t)n * in_elems, in_elems);
  };
  if (NV > 0 && blockIdx.x < g.N) pf_issue(blockIdx.x);
  for (int n = blockIdx.x; n < g.N; n += gridDim.x) {
    if (NV > 0) {
      pf.commit_image(img, in_elems, g.Hin, g.Win, g.PH, g.PW, g.pad, g.fplane, g.fw);
      __syncthreads();
      if (n + (int)gridDim.x < g.N) pf_issue(n + gridDim.x);
    } else {
      stage_image(x + (int64_t)n * in_elems, img, g, g.pad);
      __syncthreads();
    }
    float* yn = y + (int64_t)n * g.Cout * M;
    for (int pass = 0; pass < passes; ++pass) {
      int poff[CH], pos[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int chunk = (pass * 4 + wave) * CH + c;
        pos[c] = chunk * 32 + l31;
        const int pc = pos[c] < M ? pos[c] : M - 1;
        const int oy = fdiv(pc, g.fow), ox = pc - oy * g.OW;
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
      if constexpr (KS > 0) {
        // software pipeline over steps s = (channel pair, tap row): the KS*(OT+CH) ds_reads of step s+1 are
        // in flight while step s is multiplied; <= 15 newer LDS operations are ever outstanding, so the
        // lgkmcnt waits in front of the MFMAs stay exact
        float a0[KS][OT], b0[KS][CH], a1[KS][OT], b1[KS][CH];
        int c2l = 0, khl = 0;                           // position of the NEXT load
        auto load = [&](float (&a)[KS][OT], float (&b)[KS][CH]) {
          const float* wp = wrow + khl * KS * tapw + c2l * wstep;
          const float* ip = img + c2l * 2 * plane + khl * g.PW;
#pragma unroll
          for (int t = 0; t < KS; ++t) {
#pragma unroll
            for (int o = 0; o < OT; ++o) a[t][o] = wp[t * tapw + o * 32];
#pragma unroll
            for (int c = 0; c < CH; ++c) b[t][c] = ip[t + poff[c]];
          }
          if (++khl == KS) { khl = 0; ++c2l; }
        };
        auto mul = [&](float (&a)[KS][OT], float (&b)[KS][CH]) {
#pragma unroll
          for (int t = 0; t < KS; ++t)
#pragma unroll
            for (int o = 0; o < OT; ++o)
#pragma unroll
              for (int c = 0; c < CH; ++c)
                acc[o][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][o], b[t][c], acc[o][c], 0, 0, 0);
        };
        const int steps = npair * KS;
        load(a0, b0);
        int st = 0;
        for (; st + 2 <= steps; st += 2) {
          load(a1, b1);
          mul(a0, b0);
          if (st + 2 < steps) load(a0, b0);
          mul(a1, b1);
        }
        if (st < steps) mul(a0, b0);
      } else {
        for (int tap = 0; tap < taps; ++tap) {
          const int kh = tap / g.k, kw = tap - kh * g.k;
          const float* ib = img + kh * g.PW + kw;
          const float* wb = wrow + tap * tapw;
          for (int c2 = 0; c2 < npair; ++c2) {
            float a[OT], b[CH];
#pragma unroll
            for (int o = 0; o < OT; ++o) a[o] = wb[c2 * wstep + o * 32];
#pragma unroll
            for (int c = 0; c < CH; ++c) b[c] = ib[c2 * 2 * plane + poff[c]];
#pragma unroll
            for (int o = 0; o < OT; ++o)
#pragma unroll
              for (int c = 0; c < CH; ++c)
                acc[o][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[o], b[c], acc[o][c], 0, 0, 0);
          }
        }
      }
      // epilogue: rows of this lane's accumulator registers are fixed -> bias and row validity come
      // from registers set up once per kernel (no LDS read, no branch per store)
      if constexpr (EP == 1) {
        // bias + relu + 2x2 / stride-2 max-pool on the accumulators (lane = output position, register = channel):
        // the window's other column is lane ^ 1, its other row is the wave's next chunk (OW = 32: a chunk is one
        // output row, CH is even) or lane ^ OW (OW = 8 / 16: a chunk holds 32 / OW rows).  Written: the pooled
        // value and the HIT MAP, one bit per conv output position = "this position receives the pooled gradient"
        // = relu(v) == window max  AND  v >= 0  (ties all pass; relu'(0) = 1: the reference's maximum(0., x)
        // quirk).  A chunk is 32 positions, so the map of (channel, chunk) is one half of a wave-wide ballot:
        // one v_cmp and ONE dword store by lane 0 of the half-wave per accumulator register.
        const int OW = g.OW, HW = OW >> 1, PM = (g.OH >> 1) * HW, MW = M >> 5;
        float* pn = y + (int64_t)n * g.Cout * PM;
        unsigned* mn = pmask + (int64_t)n * g.Cout * MW;
        // (the 16 hit words of a tile -- one per accumulator register = per channel -- are uniform values; lane r
        //  of each half-wave keeps the word of register r, so they leave in ONE store instruction per tile)
        const int myrow = acc_row(l31 & 15, half);
        if (OW == 32) {
#pragma unroll
          for (int o = 0; o < OT; ++o)
#pragma unroll
            for (int c = 0; c + 1 < CH; c += 2) {
              const bool live = pos[c] < M;
              const int chunk = pos[c] >> 5;
              const int pidx = (chunk >> 1) * HW + (l31 >> 1);
              unsigned w0 = 0, w1 = 0;
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const float bb = bs[o * 32 + acc_row(r, half)];
                const float v0 = acc[o][c][r] + bb, v1 = acc[o][c + 1][r] + bb;
                const float r0 = fmaxf(v0, 0.f), r1 = fmaxf(v1, 0.f);
                const float mv = fmaxf(r0, r1);
                const float m = fmaxf(mv, __shfl_xor(mv, 1, 64));
                const unsigned long long h0 = __ballot(r0 == m && v0 >= 0.f), h1 = __ballot(r1 == m && v1 >= 0.f);
                if ((l31 & 15) == r) { w0 = (unsigned)(h0 >> (32 * half)); w1 = (unsigned)(h1 >> (32 * half)); }
                if (live && !(l31 & 1) && (rowmask[o] >> r & 1)) pn[(int64_t)(o * 32 + acc_row(r, half)) * PM + pidx] = m;
              }
              if (live && l31 < 16 && o * 32 + myrow < g.Cout)
                *reinterpret_cast<uint2*>(mn + (int64_t)(o * 32 + myrow) * MW + chunk) = make_uint2(w0, w1);
            }
        } else {
#pragma unroll
          for (int o = 0; o < OT; ++o)
#pragma unroll
            for (int c = 0; c < CH; ++c) {
              const bool live = pos[c] < M;
              const int oy = pos[c] / OW, ox = pos[c] - oy * OW;
              const int pidx = (oy >> 1) * HW + (ox >> 1);
              unsigned w0 = 0;
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const float v = acc[o][c][r] + bs[o * 32 + acc_row(r, half)];
                const float rr = fmaxf(v, 0.f);
                const float up = OW == 16 ? __shfl_xor(rr, 16, 64) : __shfl_xor(rr, 8, 64);
                const float m1 = fmaxf(rr, up);
                const float m = fmaxf(m1, __shfl_xor(m1, 1, 64));
                const unsigned long long h = __ballot(rr == m && v >= 0.f);
                if ((l31 & 15) == r) w0 = (unsigned)(h >> (32 * half));
                if (live && !(l31 & 1) && !(l31 & OW) && (rowmask[o] >> r & 1))
                  pn[(int64_t)(o * 32 + acc_row(r, half)) * PM + pidx] = m;
              }
              if (live && l31 < 16 && o * 32 + myrow < g.Cout) mn[(int64_t)(o * 32 + myrow) * MW + (pos[c] >> 5)] = w0;
            }
        }
      } else {
#pragma unroll
      for (int o = 0; o < OT; ++o)
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          if (pos[c] >= M) continue;
          float* yp = yn + pos[c];
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (rowmask[o] >> r & 1) yp[(int64_t)(o * 32 + acc_row(r, half)) * M] = acc[o][c][r] + bs[o * 32 + acc_row(r, half)];
        }
      }
    }
    __syncthreads();                                  // everyone is done with this image
  }
}

struct WgradGeom {
  FastDiv fw, fplane, fow, fq;   // W, H * W, OW, MB / 4
  int N, C, H, W, O, k, stride, pad, OH, OW;
  int PH, PW, OPAD, K1, KCOLS;   // K1 = C*k*k + 1 (bias column), KCOLS = K1 rounded up to 32
  int MB;                        // output positions staged per pass (a multiple of 8, or all of them)
  int DYS;                       // LDS row stride of the dy tile (odd)
  int per_block;                 // images per workgroup
};

// WT = tiles per wave; PS = 1: waves share all tiles and split the positions, 0: waves split tiles.
// The dy tile holds O + 1 rows (row O stays zero and serves the padded output channels).
// SRC = 1: `dy` is a POOLED gradient (O, OH / 2, OW / 2) expanded through `dmask` while it is staged (see
// PooledPrefetch): the relu -> max_pool backward of the fused forward epilogue.
template <int WT, int PS, int NVX, int NVD, int SRC = 0>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ dy,
                                                          float* __restrict__ partial, WgradGeom g,
                                                          const unsigned* __restrict__ dmask) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* img = lds;                                   // [C][PH][PW]
  float* dyl = img + g.C * g.PH * g.PW;               // [O + 1][DYS]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
  const int M = g.OH * g.OW, taps = g.k * g.k, KT = g.KCOLS / 32, T = (g.OPAD / 32) * KT;
  const int plane = g.PH * g.PW;
  constexpr int G = WT >= 3 ? 2 : (WT == 2 ? 3 : 4);  // position pairs per prefetch group (<= 14 ds_reads)

  for (int e = threadIdx.x; e < g.C * plane; e += blockDim.x) img[e] = 0.f;
  for (int e = threadIdx.x; e < (g.O + 1) * g.DYS; e += blockDim.x) dyl[e] = 0.f;
  // per-lane column of every owned tile: j = c*taps + tap (the layout of the (O, C, k, k) weight)
  int coff[WT], arow[WT], kind[WT];                   // kind 0 = gather, 1 = ones (bias column), 2 = zero
#pragma unroll
  for (int i = 0; i < WT; ++i) {
    const int t = PS ? i : wave * WT + i;
    const int ot = t / KT, kt = t - ot * KT;
    const int j = kt * 32 + l31;
    const int o = ot * 32 + l31;
    arow[i] = (o < g.O ? o : g.O) * g.DYS;
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
  const int pstep = PS ? 4 : 1;                       // pairs between this wave's consecutive iterations
  const int nblk = (M + g.MB - 1) / g.MB, items = (n1 - n0) * nblk, x_elems = g.C * g.H * g.W;
  TilePrefetch<NVX> px;
  float4 dv[(NVD > 0 && SRC == 0) ? NVD : 1];
  float2 dpv[(NVD > 0 && SRC == 1) ? NVD : 1];
  unsigned dpm[(NVD > 0 && SRC == 1) ? NVD : 1];
  // dy[:, m0 : m0 + mb] of image n as float4 pieces: piece e -> row e / q, columns 4 * (e % q)
  auto issue = [&](int it) {
    const int n = n0 + it / nblk, m0 = (it % nblk) * g.MB, mb = min(g.MB, M - m0), q = mb >> 2;
    if (m0 == 0) px.issue(x + (int64_t)n * x_elems, x_elems);
    const float* dyn = dy + (int64_t)n * g.O * (SRC == 1 ? (M >> 2) : M) + (SRC == 1 ? 0 : m0);
#pragma unroll
    for (int i = 0; i < NVD; ++i) {
      const int e0 = threadIdx.x + i * 256, e = e0 < g.O * q ? e0 : 0;         // (unconditional loads: see TilePrefetch)
      const int o = mb == g.MB ? fdiv(e, g.fq) : e / q, p = (e - o * q) * 4;
      if constexpr (SRC == 1) {
        const int gp = m0 + p, oy = fdiv(gp, g.fow), ox = gp - oy * g.OW;
        const int pi = (o * (g.OH >> 1) + (oy >> 1)) * (g.OW >> 1) + (ox >> 1);
        dpv[i] = *reinterpret_cast<const float2*>(dyn + pi);
        dpm[i] = dmask[((int64_t)(n * g.O + o) * M + gp) >> 5];
      } else {
        dv[i] = *reinterpret_cast<const float4*>(dyn + (int64_t)o * M + p);
      }
    }
  };
  auto commit = [&](int it) {
    const int m0 = (it % nblk) * g.MB, mb = min(g.MB, M - m0), q = mb >> 2;
    if (m0 == 0) px.commit_image(img, x_elems, g.H, g.W, g.PH, g.PW, g.pad, g.fplane, g.fw);
#pragma unroll
    for (int i = 0; i < NVD; ++i) {
      const int e = threadIdx.x + i * 256;
      if (e < g.O * q) {
        const int o = mb == g.MB ? fdiv(e, g.fq) : e / q, p = (e - o * q) * 4;
        if constexpr (SRC == 1)
          lds_store4(dyl + o * g.DYS + p, expand_pooled(dpv[i], (dpm[i] >> ((m0 + p) & 31)) & 15u));
        else
          lds_store4(dyl + o * g.DYS + p, dv[i]);
      }
    }
  };
  if (NVD > 0 && items > 0) issue(0);
  for (int it = 0; it < items; ++it) {
    const int n = n0 + it / nblk, m0 = (it % nblk) * g.MB, mb = min(g.MB, M - m0);
    if (it) __syncthreads();                          // previous item fully consumed
    if (NVD > 0) {
      commit(it);
      __syncthreads();
      if (it + 1 < items) issue(it + 1);              // in flight during this item's MFMAs
    } else {
      if (m0 == 0) {
        ConvGeom cg;
        cg.Cin = g.C; cg.Hin = g.H; cg.Win = g.W; cg.PH = g.PH; cg.PW = g.PW;
        cg.fw = g.fw; cg.fplane = g.fplane;
        stage_image(x + (int64_t)n * x_elems, img, cg, g.pad);
      }
      const float* dyn = dy + (int64_t)n * g.O * M;
      // a ragged last block leaves stale columns that are never read
      for (int e = threadIdx.x; e < g.O * mb; e += blockDim.x) {
        const int o = e / mb, p = e - o * mb;
        dyl[o * g.DYS + p] = dyn[(int64_t)o * M + m0 + p];
      }
      __syncthreads();
    }
    {
      const int pairs = (mb + 1) / 2;
      // this wave's position walk: pair index pp -> local position 2*pp + half -> (oy, ox) of m0 + that
      int pp = PS ? wave : 0;
      int lp = 2 * pp + half;                         // local position (may run past mb: masked)
      int gp = m0 + lp;
      int oy = gp / g.OW, ox = gp - oy * g.OW;
      float av[2][G][WT], bv[2][G][WT];
      auto load = [&](float (&a)[G][WT], float (&b)[G][WT]) {
#pragma unroll
        for (int q = 0; q < G; ++q) {
          const bool valid = lp < mb;
          const int lpc = valid ? lp : 0;
          const int po = valid ? oy * g.stride * g.PW + ox * g.stride : 0;
#pragma unroll
          for (int i = 0; i < WT; ++i) {
            const float av_ = dyl[arow[i] + lpc];
            a[q][i] = valid ? av_ : 0.f;
            b[q][i] = kind[i] == 0 ? img[coff[i] + po] : (kind[i] == 1 ? 1.f : 0.f);
          }
          lp += 2 * pstep;
          ox += 2 * pstep;
          while (ox >= g.OW) { ox -= g.OW; ++oy; }
        }
      };
      auto mul = [&](float (&a)[G][WT], float (&b)[G][WT]) {
#pragma unroll
        for (int q = 0; q < G; ++q)
#pragma unroll
          for (int i = 0; i < WT; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][i], b[q][i], acc[i], 0, 0, 0);
      };
      const int iters = (pairs - (PS ? wave : 0) + pstep - 1) / pstep;      // this wave's pair count
      const int groups = (iters + G - 1) / G;
      if (groups > 0) load(av[0], bv[0]);
      for (int gi = 0; gi < groups; gi += 2) {
        if (gi + 1 < groups) load(av[1], bv[1]);
        mul(av[0], bv[0]);
        if (gi + 2 < groups) load(av[0], bv[0]);
        if (gi + 1 < groups) mul(av[1], bv[1]);
      }
    }
  }
  __syncthreads();
  // partial sums: slab per workgroup; with PS the four waves' slabs are summed through LDS first
  float* out = partial + (int64_t)blockIdx.x * g.OPAD * g.KCOLS;
  if (PS) {
    float* red = lds;                                 // [4][WT][64][16] floats, reuses the image space
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((wave * WT + i) * 16 + r) * 64 + lane] = acc[i][r];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float s = 0.f;
#pragma unroll
          for (int wv = 0; wv < 4; ++wv) s += red[((wv * WT + i) * 16 + r) * 64 + lane];
          acc[i][r] = s;
        }
    } else {
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < WT; ++i) {
    const int t = PS ? i : wave * WT + i;
    if (t >= T) continue;
    const int ot = t / KT, kt = t - ot * KT;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      out[(int64_t)(ot * 32 + acc_row(r, half)) * g.KCOLS + kt * 32 + l31] = acc[i][r];
  }
}

// ---------------------------------------------------------------------------------------------------------
// Lean weight gradient (round 4).  Same product as conv_wgrad_kernel -- dW[o][j] = sum over images and positions of
// dy[o][pos] * col[pos][j], j = (c, kh, kw), + the bias column -- restructured around what the counters of the first
// version showed (MFMA pipe 0.22 / 0.44 busy, 14 VALU per LDS instruction, as many bank-conflict cycles as LDS cycles):
//   * a wave owns ONE 32-row tile of output channels and ALL column tiles of it (<= 6 accumulator tiles): the dy
//     fragment of a position pair is read once for up to six MFMAs (1 + KT LDS reads per KT MFMAs instead of 2 KT);
//     the 4 / OTN waves that share a row tile split the position pairs and are summed through LDS at the end;
//   * the COLUMN ORDER inside the tiles is a permutation chosen on the host so that the 32 gather addresses of a
//     tile -- c * plane + kh * PW + kw (+ a lane-uniform position offset) -- fall into 32 different banks (natural
//     order: 2-3 lanes per bank); the reduce kernel undoes it (`WgradPerm`: tile lane -> weight column);
//   * the bias column and the padding lanes read a plane of ones / of zeros behind the image planes with the SAME
//     address arithmetic as a gather: no per-lane select in the loop;
//   * the position walk is wave-uniform (scalar registers): per pair one vector add per LDS read and nothing else.
// ---------------------------------------------------------------------------------------------------------
struct WgradPerm {
  unsigned short col[6 * 32];                        // tile lane -> weight column j (c * taps + tap), K = bias, 0xFFFF = padding
};

template <int KT, int OTN, int NVX, int NVD, int SRC>
__global__ __launch_bounds__(256, 2) void conv_wgrad_lean_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               float* __restrict__ partial, WgradGeom g, WgradPerm perm,
                                                               const unsigned* __restrict__ dmask) {
  constexpr int NPS = 4 / OTN;                        // waves sharing a row tile = slices of the position pairs
  constexpr int G = KT >= 5 ? 1 : 2;                  // position pairs per prefetch group
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int plane = g.PH * g.PW, taps = g.k * g.k, K = g.C * taps;
  float* img = lds;                                   // [C + 2][PH][PW]: image planes, a plane of ones, a plane of zeros
  float* dyl = img + (g.C + 2) * plane;               // [O + 1][DYS], columns >= MB stay zero
  const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ot = wave % OTN, ps = wave / OTN;
  const int M = g.OH * g.OW;
  for (int e = threadIdx.x; e < (g.C + 2) * plane; e += blockDim.x) img[e] = (e >= g.C * plane && e < (g.C + 1) * plane) ? 1.f : 0.f;
  for (int e = threadIdx.x; e < (g.O + 1) * g.DYS; e += blockDim.x) dyl[e] = 0.f;
  int cvo[KT];                                        // gather offset of this lane's column in every tile + its half's position
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const int j = perm.col[kt * 32 + l31];
    int off;
    if (j < K) {
      const int c = j / taps, tap = j - c * taps, kh = tap / g.k, kw = tap - kh * g.k;
      off = c * plane + kh * g.PW + kw;
    } else {
      off = (j == K ? g.C : g.C + 1) * plane;         // ones (the bias column) / zeros
    }
    cvo[kt] = off + half * g.stride;
  }
  const int o = ot * 32 + l31;
  const int arow = (o < g.O ? o : g.O) * g.DYS + half;
  f32x16 acc[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[kt][r] = 0.f;
  __syncthreads();

  const int n0 = blockIdx.x * g.per_block, n1 = min(g.N, n0 + g.per_block);
  const int nblk = (M + g.MB - 1) / g.MB, items = (n1 - n0) * nblk, x_elems = g.C * g.H * g.W;
  const int zcol = g.DYS - 2;                         // an even column pair of the dy tile that is never written: zeros
  TilePrefetch<NVX, (KT <= 3)> px;
  px.init(x_elems, g.H, g.W, g.PH, g.PW, g.pad, g.fplane, g.fw);
  float4 dv[SRC == 0 ? NVD : 1];
  float2 dpv[SRC == 1 ? NVD : 1];
  unsigned dpm[SRC == 1 ? NVD : 1];
  // Which element of the dy block a thread stages, where it comes from and where it goes, depend on the thread and
  // the piece index only (every block has MB positions, MB a multiple of two output rows and of 32): worked out ONCE
  // here -- in the first version this index arithmetic, redone per piece and item with 64-bit addresses, was most
  // of the kernel's instruction stream.  Per item only two scalar bases change.
  const int q = g.MB >> 2;
  int src_off[NVD], msk_off[SRC == 1 ? NVD : 1], dst_sh[NVD];
  unsigned vbits = 0;
#pragma unroll
  for (int i = 0; i < NVD; ++i) {
    const int e0 = threadIdx.x + i * 256;
    const bool ok = e0 < g.O * q;
    const int e = ok ? e0 : 0;                        // (threads past the end re-read piece 0 and never commit it)
    const int oo = fdiv(e, g.fq), p = (e - oo * q) * 4;
    if constexpr (SRC == 1) {
      const int oy = fdiv(p, g.fow), ox = p - oy * g.OW;
      src_off[i] = (oo * (g.OH >> 1) + (oy >> 1)) * (g.OW >> 1) + (ox >> 1);
      msk_off[i] = oo * (M >> 5) + (p >> 5);
      dst_sh[i] = (oo * g.DYS + p) | ((p & 31) << 24);
    } else {
      src_off[i] = oo * M + p;
      dst_sh[i] = oo * g.DYS + p;
    }
    vbits |= (ok ? 1u : 0u) << i;
  }
  auto issue = [&](int it) {
    const int n = n0 + it / nblk, m0 = (it % nblk) * g.MB;
    if (m0 == 0) px.issue(x + (int64_t)n * x_elems, x_elems);
    if constexpr (SRC == 1) {
      const float* dp_base = dy + (int64_t)n * g.O * (M >> 2) + (m0 >> 2);
      const unsigned* m_base = dmask + (((int64_t)n * g.O * M + m0) >> 5);
#pragma unroll
      for (int i = 0; i < NVD; ++i) {
        dpv[i] = *reinterpret_cast<const float2*>(dp_base + (unsigned)src_off[i]);
        dpm[i] = m_base[(unsigned)msk_off[i]];
      }
    } else {
      const float* d_base = dy + (int64_t)n * g.O * M + m0;
#pragma unroll
      for (int i = 0; i < NVD; ++i) dv[i] = *reinterpret_cast<const float4*>(d_base + (unsigned)src_off[i]);
    }
  };
  auto commit = [&](int it) {
    const int m0 = (it % nblk) * g.MB;
    if (m0 == 0) px.commit_image(img, x_elems, g.H, g.W, g.PH, g.PW, g.pad, g.fplane, g.fw);
#pragma unroll
    for (int i = 0; i < NVD; ++i) {
      if ((vbits >> i) & 1u) {
        if constexpr (SRC == 1)
          lds_store4(dyl + (dst_sh[i] & 0xFFFFFF), expand_pooled(dpv[i], (dpm[i] >> (dst_sh[i] >> 24)) & 15u));
        else
          lds_store4(dyl + dst_sh[i], dv[i]);
      }
    }
  };
  if (items > 0) issue(0);
  for (int it = 0; it < items; ++it) {
    const int m0 = (it % nblk) * g.MB, mb = min(g.MB, M - m0);
    if (it) __syncthreads();                          // previous item fully consumed
    commit(it);
    __syncthreads();
    if (it + 1 < items) issue(it + 1);                // in flight during this item's MFMAs
    // this wave's pairs: pp = ps, ps + NPS, ...; positions 2 pp (+ half); all of it in scalar registers
    const int pairs = mb >> 1;                        // (mb is a multiple of 4)
    int pp = ps;
    int gp = m0 + 2 * pp;
    int oy = fdiv(gp, g.fow), ox = gp - oy * g.OW;
    float a[2][G], b[2][G][KT];
    auto load = [&](int buf) {
#pragma unroll
      for (int q = 0; q < G; ++q) {
        const bool live = pp < pairs;                 // (uniform) past the end: the zero columns of the dy tile
        const int col = live ? 2 * pp : zcol;
        const int po = live ? (oy * g.PW + ox) * g.stride : 0;
        a[buf][q] = dyl[arow + col];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) b[buf][q][kt] = img[cvo[kt] + po];
        pp += NPS;
        ox += 2 * NPS;
        if (ox >= g.OW) { ox -= g.OW; ++oy; }
        if (ox >= g.OW) { ox -= g.OW; ++oy; }
      }
    };
    auto mul = [&](int buf) {
#pragma unroll
      for (int q = 0; q < G; ++q)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
          acc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[buf][q], b[buf][q][kt], acc[kt], 0, 0, 0);
    };
    const int iters = (pairs - ps + NPS - 1) / NPS;
    const int groups = (iters + G - 1) / G;
    if (groups > 0) load(0);
    for (int gi = 0; gi < groups; gi += 2) {
      if (gi + 1 < groups) load(1);
      mul(0);
      if (gi + 2 < groups) load(0);
      if (gi + 1 < groups) mul(1);
    }
  }
  __syncthreads();
  // the NPS waves of a row tile: summed through LDS (the image space is free now), slice 0 stores
  float* out = partial + (int64_t)blockIdx.x * g.OPAD * g.KCOLS;
  if (NPS > 1) {
    float* red = lds;                                 // [NPS - 1 slices][OTN][KT][16][64]: slice 0 keeps its own in registers
    if (ps != 0) {
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((((ps - 1) * OTN + ot) * KT + kt) * 16 + r) * 64 + lane] = acc[kt][r];
    }
    __syncthreads();
    if (ps != 0) return;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float sum = acc[kt][r];
#pragma unroll
        for (int sl = 1; sl < NPS; ++sl) sum += red[((((sl - 1) * OTN + ot) * KT + kt) * 16 + r) * 64 + lane];
        acc[kt][r] = sum;
      }
  }
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      out[(int64_t)(ot * 32 + acc_row(r, half)) * g.KCOLS + kt * 32 + l31] = acc[kt][r];
}

// the reduce of the lean kernel: slab column p holds weight column perm[p]
__global__ __launch_bounds__(256) void conv_wgrad_reduce_perm_kernel(const float* __restrict__ partial, int slabs, int OPAD,
                                                                      int KCOLS, int O, int K, WgradPerm perm,
                                                                      float* __restrict__ dw, float* __restrict__ db,
                                                                      int accumulate) {
  __shared__ float red[8][32];
  const int total = O * KCOLS;
  const int e = blockIdx.x * 32 + (threadIdx.x & 31), grp = threadIdx.x >> 5;
  float s = 0.f;
  int o = 0, j = 0xFFFF;
  if (e < total) {
    o = e / KCOLS;
    const int pcol = e - o * KCOLS;
    j = perm.col[pcol];
    if (j <= K) {
      const float* p = partial + (int64_t)o * KCOLS + pcol;
      const int64_t stride = (int64_t)OPAD * KCOLS;
      for (int b = grp; b < slabs; b += 8) s += p[b * stride];
    }
  }
  red[grp][threadIdx.x & 31] = s;
  __syncthreads();
  if (grp == 0 && e < total && j <= K) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += red[q][threadIdx.x];
    if (j < K) {
      if (dw) dw[(int64_t)o * K + j] = accumulate ? dw[(int64_t)o * K + j] + t : t;
    } else if (db) {
      db[o] = accumulate ? db[o] + t : t;
    }
  }
}

// dw[o][j] (+)= sum_slabs partial[slab][o][j] (j < K), db[o] (+)= column K.  A workgroup owns 32
// consecutive elements; its 8 thread groups sum interleaved slabs and combine through LDS in a fixed
// order: deterministic, and 8x the memory parallelism of one thread per element.
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ partial, int slabs,
                                                                 int OPAD, int KCOLS, int O, int K,
                                                                 float* __restrict__ dw, float* __restrict__ db,
                                                                 int accumulate) {
  __shared__ float red[8][32];
  const int total = O * (K + 1);
  const int e = blockIdx.x * 32 + (threadIdx.x & 31), grp = threadIdx.x >> 5;
  float s = 0.f;
  int o = 0, j = 0;
  if (e < total) {
    o = e / (K + 1); j = e - o * (K + 1);
    const float* p = partial + (int64_t)o * KCOLS + j;
    const int64_t stride = (int64_t)OPAD * KCOLS;
    for (int b = grp; b < slabs; b += 8) s += p[b * stride];
  }
  red[grp][threadIdx.x & 31] = s;
  __syncthreads();
  if (grp == 0 && e < total) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += red[q][threadIdx.x];
    if (j < K) {
      if (dw) dw[(int64_t)o * K + j] = accumulate ? dw[(int64_t)o * K + j] + t : t;
    } else if (db) {
      db[o] = accumulate ? db[o] + t : t;
    }
  }
}

__global__ __launch_bounds__(256) void pool_mask_expand_kernel(const float* __restrict__ dp,
                                                               const unsigned* __restrict__ hit,
                                                               float* __restrict__ dy, int64_t n, int PH, int PW) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;          // pooled element
  if (i >= n) return;
  const int px = (int)(i % PW);
  const int64_t t = i / PW;
  const int py = (int)(t % PH);
  const int64_t row = t / PH;
  const float v = dp[i];
  const int64_t p0 = (row * (2 * PH) + 2 * py) * (int64_t)(2 * PW) + 2 * px, p1 = p0 + 2 * PW;   // flat positions
  const unsigned b0 = (hit[p0 >> 5] >> (p0 & 31)) & 3u, b1 = (hit[p1 >> 5] >> (p1 & 31)) & 3u;
  *reinterpret_cast<float2*>(dy + p0) = make_float2((b0 & 1u) ? v : 0.f, (b0 & 2u) ? v : 0.f);
  *reinterpret_cast<float2*>(dy + p1) = make_float2((b1 & 1u) ? v : 0.f, (b1 & 2u) ? v : 0.f);
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
  g.fw = make_fastdiv(Win); g.fplane = make_fastdiv(Hin * Win); g.fow = make_fastdiv(g.OW > 0 ? g.OW : 1);
  g.fopad = make_fastdiv(g.OPAD); g.fcp = make_fastdiv(g.Cp);
  return g.OH > 0 && g.OW > 0;
}
int64_t fwd_lds(const ConvGeom& g) {
  return 4ll * ((int64_t)g.k * g.k * g.Cp * g.OPAD + (int64_t)g.Cp * g.PH * g.PW + g.OPAD) + 64;
}
bool fwd_ok(const ConvGeom& g) { return g.OPAD <= 64 && g.pad >= 0 && fwd_lds(g) <= kMaxLds; }

// ep = 1: fused bias + relu + 2x2 max-pool epilogue (y = pooled map, pmask = its 4-bit masks);
// src = 1: x is a pooled gradient expanded through pmask while staged.
int launch_direct(const float* x, const float* w, const float* bias, float* y, const ConvGeom& g, hipStream_t st,
                  int ep = 0, int src = 0, unsigned* pmask = nullptr) {
  const int64_t lds = fwd_lds(g);
  const int per_cu = (int)(kMaxLds / lds) < 1 ? 1 : (int)((160 * 1024) / lds);
  int grid = 256 * (per_cu > 4 ? 4 : per_cu);
  if (grid > g.N) grid = g.N;
  const int M = g.OH * g.OW, chunks = (M + 31) / 32;
  // float4 per thread that hold one input image in registers (0: rows not float4-aligned / too large)
  const int in_elems = g.Cin * g.Hin * g.Win;
  int nv = ((g.Win & 3) == 0 && in_elems <= 16 * 1024) ? (in_elems / 4 + 255) / 256 : 0;
#define PDN_CONV_LAUNCH_KNE(OT, CH, KS, NV, EP, SRC)                                                   \
  do {                                                                                                 \
    auto kern = conv_direct_kernel<OT, CH, KS, NV, EP, SRC>;                                           \
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                       (int)lds);                                                      \
    if (e != hipSuccess) { pdn_set_error("conv: LDS attribute: %s", hipGetErrorString(e)); return (int)e; } \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, x, w, bias, y, g, pmask);                 \
  } while (0)
#define PDN_CONV_LAUNCH_KN(OT, CH, KS, NV) PDN_CONV_LAUNCH_KNE(OT, CH, KS, NV, 0, 0)
  /* the fused forms exist for 3x3 kernels on the register-prefetch path (what fused_ok() admits) */
#define PDN_CONV_LAUNCH_KF(OT, CH, NV)                                                                 \
  do {                                                                                                 \
    if (ep) PDN_CONV_LAUNCH_KNE(OT, CH, 3, NV, 1, 0);                                                  \
    else if (src) PDN_CONV_LAUNCH_KNE(OT, CH, 3, NV, 0, 1);                                            \
    else PDN_CONV_LAUNCH_KNE(OT, CH, 3, NV, 0, 0);                                                     \
  } while (0)
#define PDN_CONV_LAUNCH_K(OT, CH, KS)                                                                  \
  do {                                                                                                 \
    if (nv == 0 || KS != 3) PDN_CONV_LAUNCH_KN(OT, CH, KS, 0);                                         \
    else if (nv <= 4) PDN_CONV_LAUNCH_KF(OT, CH, 4);                                                   \
    else if (nv <= 8) PDN_CONV_LAUNCH_KF(OT, CH, 8);                                                   \
    else PDN_CONV_LAUNCH_KF(OT, CH, 16);                                                               \
  } while (0)
#define PDN_CONV_LAUNCH(OT, CH)                                                                        \
  do {                                                                                                 \
    if (g.k == 3) PDN_CONV_LAUNCH_K(OT, CH, 3);                                                        \
    else if (g.k == 5 && OT * CH <= 2) PDN_CONV_LAUNCH_K(OT, CH, 5);                                   \
    else if (g.k == 1) PDN_CONV_LAUNCH_K(OT, CH, 1);                                                   \
    else PDN_CONV_LAUNCH_K(OT, CH, 0);                                                                 \
  } while (0)
  if (g.OPAD == 64) {
    // (measured equal with the fused epilogue: four accumulator tiles with a few spilled registers vs two tiles)
    if (chunks >= 8) PDN_CONV_LAUNCH(2, 2); else PDN_CONV_LAUNCH(2, 1);
  } else {
    if (chunks >= 16) PDN_CONV_LAUNCH(1, 4); else if (chunks >= 8) PDN_CONV_LAUNCH(1, 2); else PDN_CONV_LAUNCH(1, 1);
  }
#undef PDN_CONV_LAUNCH
#undef PDN_CONV_LAUNCH_K
#undef PDN_CONV_LAUNCH_KF
#undef PDN_CONV_LAUNCH_KN
#undef PDN_CONV_LAUNCH_KNE
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
  // stage all positions of an image at once when two workgroups per CU still fit, else blocks of 512
  const int M = g.OH * g.OW;
  g.MB = M;
  const int64_t img_b = 4ll * C * g.PH * g.PW;
  if (img_b + 4ll * (O + 1) * (M | 1) > 78 * 1024 && M > 512) g.MB = 512;
  g.DYS = g.MB | 1;
  g.fw = make_fastdiv(W); g.fplane = make_fastdiv(H * W); g.fow = make_fastdiv(g.OW > 0 ? g.OW : 1);
  g.fq = make_fastdiv(g.MB / 4 > 0 ? g.MB / 4 : 1);
  return g.OH > 0 && g.OW > 0;
}
int64_t wgrad_lds(const WgradGeom& g) {
  int64_t b = 4ll * ((int64_t)g.C * g.PH * g.PW + (int64_t)(g.O + 1) * g.DYS) + 64;
  const int64_t red = 4ll * 4 * 3 * 16 * 64;          // cross-wave reduction scratch of the position-split form
  return b > red ? b : red;
}
int wgrad_tiles(const WgradGeom& g) { return (g.OPAD / 32) * (g.KCOLS / 32); }
bool wgrad_prefetch(const WgradGeom& g) {     // operands of one item fit the register prefetch (float4-aligned rows)
  const int M = g.OH * g.OW;
  return (g.W & 3) == 0 && (M & 3) == 0 && (g.MB & 3) == 0 && g.C * g.H * g.W <= 8 * 1024 &&
         g.O * (g.MB < M ? g.MB : M) <= 16 * 1024;
}
// Column order of the lean kernel: tile lane -> weight column, chosen so that the gather addresses of a tile's 32 lanes
// (c * plane + kh * PW + kw) occupy 32 different LDS banks wherever the columns allow it (greedy colouring; the bias
// column reads the ones plane, padding lanes the zeros plane: all padding lanes share ONE address = a broadcast).
bool wgrad_lean_geom(const WgradGeom& g, int& KT, int& OTN) {
  KT = g.KCOLS / 32; OTN = g.OPAD / 32;
  return KT >= 1 && KT <= 6 && (OTN == 1 || OTN == 2) && (g.MB & 3) == 0 && g.C * g.k * g.k + 1 <= 6 * 32;
}
int64_t wgrad_lean_lds(const WgradGeom& g) {
  const int KT = g.KCOLS / 32;
  int64_t b = 4ll * ((int64_t)(g.C + 2) * g.PH * g.PW + (int64_t)(g.O + 1) * g.DYS) + 64;
  const int OTN = g.OPAD / 32;
  const int64_t red = 4ll * (4 / OTN - 1) * OTN * KT * 16 * 64;
  return b > red ? b : red;
}
void wgrad_build_perm(const WgradGeom& g, WgradPerm& pm) {
  const int KT = g.KCOLS / 32, taps = g.k * g.k, K = g.C * taps, plane = g.PH * g.PW;
  int fill[6] = {0, 0, 0, 0, 0, 0};
  unsigned used[6] = {0, 0, 0, 0, 0, 0};             // banks taken per tile
  for (int i = 0; i < 6 * 32; ++i) pm.col[i] = 0xFFFF;
  auto bank_of = [&](int j) {
    if (j == K) return (g.C * plane) & 31;
    const int c = j / taps, tap = j - c * taps, kh = tap / g.k, kw = tap - kh * g.k;
    return (c * plane + kh * g.PW + kw) & 31;
  };
  // two passes: conflict-free placements first, whatever is left goes where there is room (the padding lanes that
  // fill the tiles up afterwards share one address: at worst one two-way conflict per tile)
  unsigned char placed[6 * 32 + 1] = {0};
  for (int pass = 0; pass < 2; ++pass)
    for (int j = 0; j <= K; ++j) {
      if (placed[j]) continue;
      const unsigned bit = 1u << bank_of(j);
      for (int t = 0; t < KT; ++t) {
        const unsigned busy = used[t];
        if (fill[t] < 32 && (pass == 1 || !(busy & bit))) {
          pm.col[t * 32 + fill[t]++] = (unsigned short)j;
          used[t] |= bit;
          placed[j] = 1;
          break;
        }
      }
    }
}
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
  return 4ll * wgrad_blocks(g) * g.OPAD * g.KCOLS;
}

}  // extern "C"

namespace {
int launch_wgrad(const float* x, const float* dy, const unsigned* dmask, float* dw, float* db, int accumulate, int N,
                 int C, int H, int W, int O, int k, int stride, int pad, void* workspace, int64_t workspace_bytes,
                 void* stream, const char* who) {
  WgradGeom g;
  if (!wgrad_geom(g, N, C, H, W, O, k, stride, pad) || !wgrad_ok(g)) {
    pdn_set_error("%s: shape outside the direct kernel (use the GEMM path)", who);
    return PDN_EUNSUPPORTED;
  }
  const int blocks = wgrad_blocks(g);
  g.per_block = (N + blocks - 1) / blocks;
  const int used = (N + g.per_block - 1) / g.per_block;
  const int T = wgrad_tiles(g);
  {
    // the lean kernel (one row tile and all column tiles per wave, conflict-free column order): LeNet-class shapes
    static const int lean_env = getenv("PDN_CONV_WGRAD_LEAN") ? atoi(getenv("PDN_CONV_WGRAD_LEAN")) : 1;
    int KT = 0, OTN = 0;
    const int M_ = g.OH * g.OW;
    const int mb_keep = g.MB;
    g.DYS = (g.MB + 2) | 1;                           // two zero columns behind the staged positions
    // half the positions per item when two workgroups would not fit a CU, or when the wide (>= 4 column tiles) forms
    // would need more than 8 staging pieces per thread (their accumulators leave no registers for 16)
    if ((wgrad_lean_lds(g) > 80 * 1024 || (g.KCOLS >= 128 && O * (g.MB / 4) > 8 * 256)) && (g.MB & 7) == 0 && g.MB > 128) {
      g.MB >>= 1;
      g.DYS = (g.MB + 2) | 1;
    }
    g.fq = make_fastdiv(g.MB / 4);
    const int need = (O * (g.MB / 4) + 255) / 256;    // 16-byte pieces of the dy block per thread
    const bool lean = lean_env && wgrad_lean_geom(g, KT, OTN) && wgrad_prefetch(g) && wgrad_lean_lds(g) <= 80 * 1024 &&
                      M_ % g.MB == 0 && g.MB % 32 == 0 && g.MB % (2 * g.OW) == 0 && need <= (KT >= 4 ? 8 : 16) &&
                      (!dmask || ((g.OW & 3) == 0 && (g.OH & 1) == 0 && (M_ & 31) == 0)) && (g.OW & 1) == 0 &&
                      workspace && workspace_bytes >= 4ll * used * g.OPAD * g.KCOLS;
    if (getenv("PDN_CONV_DEBUG"))
      fprintf(stderr, "wgrad %s: lean=%d geom=%d (KT %d OTN %d) prefetch=%d lds=%lld MB=%d M=%d need=%d ws=%lld/%lld\n", who, (int)lean,
              (int)wgrad_lean_geom(g, KT, OTN), KT, OTN, (int)wgrad_prefetch(g), (long long)wgrad_lean_lds(g), g.MB, M_, need,
              (long long)workspace_bytes, (long long)(4ll * used * g.OPAD * g.KCOLS));
    if (lean) {
      WgradPerm pm;
      wgrad_build_perm(g, pm);
      const int64_t lds = wgrad_lean_lds(g);
      hipStream_t st = (hipStream_t)stream;
      float* partial = (float*)workspace;
#define PDN_LEAN_GO(KT_, OTN_, SRC_)                                                                       \
  do {                                                                                                    \
    auto kern8 = conv_wgrad_lean_kernel<KT_, OTN_, 8, 8, SRC_>;                                           \
    auto kern16 = conv_wgrad_lean_kernel<KT_, OTN_, 8, 16, SRC_>;                                         \
    auto kern = need <= 8 ? kern8 : kern16;                                                               \
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e != hipSuccess) { pdn_set_error("conv: LDS attribute: %s", hipGetErrorString(e)); return (int)e; } \
    hipLaunchKernelGGL(kern, dim3(used), dim3(256), lds, st, x, dy, partial, g, pm, dmask);               \
  } while (0)
#define PDN_LEAN_KT(OTN_, SRC_)                                                                           \
  switch (KT) {                                                                                           \
    case 1: PDN_LEAN_GO(1, OTN_, SRC_); break; case 2: PDN_LEAN_GO(2, OTN_, SRC_); break;                 \
    case 3: PDN_LEAN_GO(3, OTN_, SRC_); break; case 4: PDN_LEAN_GO(4, OTN_, SRC_); break;                 \
    case 5: PDN_LEAN_GO(5, OTN_, SRC_); break; default: PDN_LEAN_GO(6, OTN_, SRC_); break;                \
  }
      if (OTN == 1) { if (dmask) { PDN_LEAN_KT(1, 1) } else { PDN_LEAN_KT(1, 0) } }
      else { if (dmask) { PDN_LEAN_KT(2, 1) } else { PDN_LEAN_KT(2, 0) } }
#undef PDN_LEAN_KT
#undef PDN_LEAN_GO
      PDN_LAUNCH_CHECK();
      const int total = O * g.KCOLS;
      hipLaunchKernelGGL(conv_wgrad_reduce_perm_kernel, dim3((total + 31) / 32), dim3(256), 0, st, partial, used, g.OPAD,
                         g.KCOLS, O, g.K1 - 1, pm, dw, db, accumulate);
      PDN_LAUNCH_CHECK();
      return PDN_OK;
    }
    g.MB = mb_keep;
    g.DYS = g.MB | 1;
    g.fq = make_fastdiv(g.MB / 4);
  }
  const int ps = T < 4 ? 1 : 0;
  const int slabs = used;
  if (!workspace || workspace_bytes < 4ll * slabs * g.OPAD * g.KCOLS) {
    pdn_set_error("%s: workspace too small", who);
    return PDN_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int64_t lds = wgrad_lds(g);
  float* partial = (float*)workspace;
  const bool pre = wgrad_prefetch(g);
  if (dmask && !(pre && (g.OW & 3) == 0 && (g.OH & 1) == 0 && ((g.OH * g.OW) & 31) == 0)) {
    pdn_set_error("%s: pooled source outside the prefetching kernel", who);
    return PDN_EUNSUPPORTED;
  }
#define PDN_WGRAD_LAUNCH(WT, PS)                                                                       \
  do {                                                                                                 \
    if (dmask) PDN_WGRAD_LAUNCH_N(WT, PS, 8, 16, 1);                                                   \
    else if (pre) PDN_WGRAD_LAUNCH_N(WT, PS, 8, 16, 0);                                                \
    else PDN_WGRAD_LAUNCH_N(WT, PS, 0, 0, 0);                                                          \
  } while (0)
#define PDN_WGRAD_LAUNCH_N(WT, PS, NVX, NVD, SRC)                                                      \
  do {                                                                                                 \
    auto kern = conv_wgrad_kernel<WT, PS, NVX, NVD, SRC>;                                              \
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                       (int)lds);                                                      \
    if (e != hipSuccess) { pdn_set_error("conv: LDS attribute: %s", hipGetErrorString(e)); return (int)e; } \
    hipLaunchKernelGGL(kern, dim3(used), dim3(256), lds, st, x, dy, partial, g, dmask);                \
  } while (0)
  if (ps) {
    if (T == 1) PDN_WGRAD_LAUNCH(1, 1); else if (T == 2) PDN_WGRAD_LAUNCH(2, 1); else PDN_WGRAD_LAUNCH(3, 1);
  } else {
    const int wt = (T + 3) / 4;
    if (wt == 1) PDN_WGRAD_LAUNCH(1, 0); else if (wt == 2) PDN_WGRAD_LAUNCH(2, 0);
    else if (wt == 3) PDN_WGRAD_LAUNCH(3, 0); else PDN_WGRAD_LAUNCH(4, 0);
  }
#undef PDN_WGRAD_LAUNCH
#undef PDN_WGRAD_LAUNCH_N
  PDN_LAUNCH_CHECK();
  const int total = O * g.K1;
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((total + 31) / 32), dim3(256), 0, st, partial, slabs,
                     g.OPAD, g.KCOLS, O, g.K1 - 1, dw, db, accumulate);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

// geometry of the fused conv -> relu -> 2x2 max-pool forms: what the epilogue / the expanding loads can take
bool fused_fwd_ok(const ConvGeom& g) {
  const int M = g.OH * g.OW, in_elems = g.Cin * g.Hin * g.Win;
  if (!fwd_ok(g) || g.k != 3 || (g.Win & 3) || in_elems > 16 * 1024) return false;
  if (!(g.OW == 8 || g.OW == 16 || g.OW == 32) || (g.OH & 1) || (M & 31)) return false;
  return g.OW != 32 || M / 32 >= 8;                  // (OW = 32 pairs two chunks of one wave: CH must be even)
}
}  // namespace

extern "C" {

/* dw (O, C, k, k) and db (O) [either may be NULL]; accumulate != 0 adds into them */
int pdn_conv2d_bwd_weight_f32(const float* x, const float* dy, float* dw, float* db, int accumulate, int N,
                              int C, int H, int W, int O, int k, int stride, int pad, void* workspace,
                              int64_t workspace_bytes, void* stream) {
  if (N == 0) return PDN_OK;
  PDN_CHECK_ARG(x && dy && (dw || db), "pdn_conv2d_bwd_weight_f32: null operand");
  return launch_wgrad(x, dy, nullptr, dw, db, accumulate, N, C, H, W, O, k, stride, pad, workspace, workspace_bytes, stream,
                      "pdn_conv2d_bwd_weight_f32");
}

/* ---- conv -> relu -> max_pool(2, 2) as ONE node (examples/pydynet/mnist.py:92-95) ------------------------------
 * bitmask: 1 forward, 2 data gradient, 4 weight gradient from the pooled gradient (2 / 4 only with 1). */
int pdn_conv2d_relu_pool_supported(int C, int H, int W, int O, int k, int stride, int pad) {
  if (C <= 0 || O <= 0 || k <= 0 || stride <= 0 || pad < 0 || H + 2 * pad < k || W + 2 * pad < k) return 0;
  ConvGeom f;
  if (!fwd_geom(f, 1, C, H, W, O, k, stride, pad, 0, C, O) || !fused_fwd_ok(f)) return 0;
  int mask = 1;
  if (stride == 1 && k - 1 - pad >= 0) {
    ConvGeom d;
    if (fwd_geom(d, 1, O, f.OH, f.OW, C, k, 1, k - 1 - pad, 1, C, O) && fwd_ok(d) && d.OH == H && d.OW == W &&
        (f.OW & 3) == 0 && O * f.OH * f.OW <= 16 * 1024)
      mask |= 2;
  }
  WgradGeom g;
  if (wgrad_geom(g, 1, C, H, W, O, k, stride, pad) && wgrad_ok(g) && wgrad_prefetch(g) && (g.OW & 3) == 0) mask |= 4;
  return mask;
}

/* pooled (N, O, OH / 2, OW / 2) = max_pool2x2(relu(conv(x, w) + bias)); mask (N, O, OH * OW / 32) words: bit
 * (p & 31) of word p >> 5 set when conv output position p = oy * OW + ox receives the pooled gradient (its relu equals
 * the window maximum and y >= 0) */
int pdn_conv2d_relu_pool_fwd_f32(const float* x, const float* w, const float* bias, float* pooled, unsigned* mask,
                                 int N, int C, int H, int W, int O, int k, int stride, int pad, void* stream) {
  if (N == 0) return PDN_OK;
  PDN_CHECK_ARG(x && w && pooled && mask, "pdn_conv2d_relu_pool_fwd_f32: null operand");
  ConvGeom g;
  if (!fwd_geom(g, N, C, H, W, O, k, stride, pad, 0, C, O) || !fused_fwd_ok(g)) {
    pdn_set_error("pdn_conv2d_relu_pool_fwd_f32: shape outside the fused kernel");
    return PDN_EUNSUPPORTED;
  }
  // the LeNet shapes: channel-innermost LDS image, four k-steps per ds_read_b128 (csrc/conv_quad.hip)
  if (conv_quad_fwd_supported(C, H, W, O, k, stride, pad))
    return conv_quad_relu_pool_fwd(x, w, bias, pooled, mask, N, C, H, W, O, stream);
  return launch_direct(x, w, bias, pooled, g, (hipStream_t)stream, 1, 0, mask);
}

/* dx = conv_transpose(expand(dpooled, mask), w) */
int pdn_conv2d_relu_pool_bwd_data_f32(const float* dpooled, const unsigned* mask, const float* w, float* dx, int N,
                                      int C, int H, int W, int O, int k, int stride, int pad, void* stream) {
  if (N == 0) return PDN_OK;
  PDN_CHECK_ARG(dpooled && mask && w && dx, "pdn_conv2d_relu_pool_bwd_data_f32: null operand");
  if (!(pdn_conv2d_relu_pool_supported(C, H, W, O, k, stride, pad) & 2)) {
    pdn_set_error("pdn_conv2d_relu_pool_bwd_data_f32: shape outside the fused kernel");
    return PDN_EUNSUPPORTED;
  }
  // LeNet's second layer: col2im-style product, accumulators scattered with ds_add_f32 (csrc/conv_quad.hip)
  if (conv_quad_dgrad_supported(C, H, W, O, k, stride, pad))
    return conv_quad_relu_pool_bwd_data(dpooled, mask, w, dx, N, C, H, W, O, stream);
  const int OH = H + 2 * pad - k + 1, OW = W + 2 * pad - k + 1;
  ConvGeom g;
  fwd_geom(g, N, O, OH, OW, C, k, 1, k - 1 - pad, 1, C, O);
  return launch_direct(dpooled, w, nullptr, dx, g, (hipStream_t)stream, 0, 1, const_cast<unsigned*>(mask));
}

/* dw, db from x and expand(dpooled, mask); workspace as pdn_conv2d_bwd_weight_workspace_bytes */
int pdn_conv2d_relu_pool_bwd_weight_f32(const float* x, const float* dpooled, const unsigned* mask, float* dw,
                                        float* db, int accumulate, int N, int C, int H, int W, int O, int k, int stride,
                                        int pad, void* workspace, int64_t workspace_bytes, void* stream) {
  if (N == 0) return PDN_OK;
  PDN_CHECK_ARG(x && dpooled && mask && (dw || db), "pdn_conv2d_relu_pool_bwd_weight_f32: null operand");
  // LeNet's second layer: shifted image copies in LDS, dy expanded in registers (csrc/conv_quad.hip)
  if (conv_quad_wgrad_supported(C, H, W, O, k, stride, pad))
    return conv_quad_relu_pool_bwd_weight(x, dpooled, mask, dw, db, accumulate, N, C, H, W, O, workspace, workspace_bytes,
                                          stream);
  return launch_wgrad(x, dpooled, mask, dw, db, accumulate, N, C, H, W, O, k, stride, pad, workspace, workspace_bytes,
                      stream, "pdn_conv2d_relu_pool_bwd_weight_f32");
}

/* dy (rows, OH, OW) = expand(dpooled (rows, OH / 2, OW / 2), mask): the generic fallback when a fused backward form
 * does not take the shape (then the plain conv kernels run on dy) */
int pdn_pool_mask_expand_f32(const float* dpooled, const unsigned* mask, float* dy, int64_t rows, int OH, int OW,
                             void* stream) {
  if (rows == 0) return PDN_OK;
  PDN_CHECK_ARG(dpooled && mask && dy && OH > 0 && OW > 0 && !(OH & 1) && !(OW & 1) && ((rows * OH * OW) & 31) == 0,
                "pdn_pool_mask_expand_f32: bad arguments (even extents, rows * OH * OW a multiple of 32)");
  const int64_t n = rows * (OH / 2) * (OW / 2);
  hipLaunchKernelGGL(pool_mask_expand_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     dpooled, mask, dy, n, OH / 2, OW / 2);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

}  // extern "C"
