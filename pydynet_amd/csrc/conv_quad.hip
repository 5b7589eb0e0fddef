// 3 x 3 / stride 1 / pad 1 convolutions of compile-time geometry on fp32 MFMA with FOUR k-steps per LDS read (gfx950).
//
// Same mathematics as csrc/conv_direct.hip (the reference's pad -> im2col -> GEMM -> NCHW view of
// pydynet/nn/functional.py:194-281 with the im2col matrix existing only as LDS addresses), different operand layout.
// conv_direct.hip feeds every v_mfma_f32_32x32x2_f32 with one ds_read_b32 per operand at a gather address worked out
// per MFMA: its counters read 5.3 VALU + 3.3 SALU + 1.1 LDS instructions per MFMA and 26-50 % of the LDS cycles are
// bank conflicts (profiles/r05_lenet_b4096_pmc.txt) -- the kernels are bound by their own instruction streams.  Here
//   * the staged image is CHANNEL-INNERMOST, [y][x][c] with c padded to a multiple of four: the four floats a lane reads
//     with ONE ds_read_b128 are four channels of one tap at its position, i.e. four contraction indices;
//   * the contraction order is chosen to fit that read: MFMA j of a group of four contracts channel 4 q + j of
//     "unit" (tap, q) from the lanes 0-31 and channel 4 q' + j of unit (tap', q') from the lanes 32-63 -- any order is
//     legal (results differ from a BLAS by summation order only) and the weights are staged in the same order,
//     [out channel][pair][half][4], so their fragment is one ds_read_b128 too;
//   * every address is lane base + compile-time constant: no integer arithmetic in the loop; 2 + 2 reads feed 16 MFMAs
//     of a 64-position x 64-channel wave tile (0.25 LDS instructions per MFMA);
//   * bank conflicts: a ds_read_b128 is served in 16-lane groups {0-3, 12-15, 20-27}, ... (MI355X_MICROARCH.md, LDS);
//     a 32-lane tile is one image row of 32 positions, or the SAME row of TWO images (W = 16) whose LDS frames lie a
//     multiple of 256 B apart: each group then reads 16 different 16-byte slots; weight rows have an odd slot count.
// One 8-wave workgroup per CU, persistent over images; the next image (pair) is prefetched into registers during the
// MFMA phase and written to the other LDS buffer: one barrier per image (pair).
//
//   conv_quad_fwd_kernel<G>   conv + bias + relu + max_pool(2, 2) (examples/pydynet/mnist.py:92-95): writes the pooled
//                             map and the hit map of csrc/conv_direct.hip (one bit per conv output position)
#include "common.h"
#include "conv_quad.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

template <int C_, int H_, int W_, int O_>
struct QuadGeom {
  static constexpr int C = C_, H = H_, W = W_, O = O_;
  static constexpr int CQ = (C + 3) / 4, CP = 4 * CQ;          // channel quads; floats per pixel in LDS
  static constexpr int PH = H + 2, PW = W + 2;
  static constexpr int UNITS = 9 * CQ, NP = (UNITS + 1) / 2;   // (tap, quad) units; pairs (one unit per half-wave)
  static constexpr int OT = (O + 31) / 32;
  static constexpr int WS = NP * 8 + 4;                        // weight row stride: an ODD number of 16-byte slots
  static constexpr int IPT = 32 / W;                           // images per 32-lane tile
  static constexpr int IMGS = ((PH * PW * CP + 63) / 64) * 64; // frame stride: a multiple of 256 B
  static constexpr int RP = H / 2, PASSES = (RP + 7) / 8;      // row pairs per image; per wave
  static constexpr int F4 = IPT * C * H * W / 4;               // 16-byte pieces of one tile group
  static constexpr int NV = (F4 + 511) / 512;
  static constexpr int LDS_FLOATS = OT * 32 * WS + OT * 32 + 2 * IPT * IMGS;
  // workgroups per CU: a second one fills the barrier / epilogue gaps of the first where LDS and registers (one channel
  // tile: 32 accumulators) allow it
  static constexpr int WGS = (OT == 1 && LDS_FLOATS * 4 * 2 <= 160 * 1024) ? 2 : 1;
  static_assert(W == 16 || W == 32, "a 32-lane tile is one row of 32 positions or one row of two images");
  static_assert((H & 1) == 0 && (W & 3) == 0, "2 x 2 pooling, float4 rows");
  // frame offset (floats) of unit u relative to a lane's pixel
  static __host__ __device__ constexpr int unit_off(int u) {
    return (((u / CQ) / 3) * PW + ((u / CQ) % 3)) * CP + (u % CQ) * 4;
  }
};

__device__ __forceinline__ int qacc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// lane ^ 1 through DPP (quad_perm [1, 0, 3, 2]): no LDS round trip
__device__ __forceinline__ float dpp_xor1(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
}

// lane x + 1 / lane x - 1 of the same 16-lane row, zero at the row's ends (DPP row_shl:1 / row_shr:1, bound_ctrl)
__device__ __forceinline__ float dpp_row_shl1(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x101, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_row_shr1(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xF, 0xF, true));
}

template <class G>
__global__ __launch_bounds__(512, G::WGS) void conv_quad_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ pooled,
                                                               unsigned* __restrict__ mask, int N) {
  constexpr int C = G::C, H = G::H, W = G::W, O = G::O, CQ = G::CQ, CP = G::CP, PW = G::PW, NP = G::NP, OT = G::OT,
                WS = G::WS, IPT = G::IPT, IMGS = G::IMGS, NV = G::NV, UNITS = G::UNITS;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* wl = lds;                       // [OT * 32][WS]
  float* bs = wl + OT * 32 * WS;         // [OT * 32]
  float* im = bs + OT * 32;              // [2][IPT][IMGS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;

  // weights in contraction order: row oc, pair p, half h, j -> w[oc][4 quad(u) + j][tap(u)], u = p + h NP
  for (int e = tid; e < OT * 32 * NP * 8; e += 512) {
    const int oc = e / (NP * 8), r = e - oc * (NP * 8), p = r >> 3, h = (r >> 2) & 1, j = r & 3;
    const int u = p + h * NP;
    float v = 0.f;
    if (u < UNITS && oc < O) {
      const int tap = u / CQ, c = (u - tap * CQ) * 4 + j;
      if (c < C) v = w[(oc * C + c) * 9 + tap];
    }
    wl[oc * WS + r] = v;
  }
  for (int e = tid; e < OT * 32; e += 512) bs[e] = (bias && e < O) ? bias[e] : 0.f;
  for (int e = tid; e < 2 * IPT * IMGS; e += 512) im[e] = 0.f;        // halo and padded channels stay zero
  // staging plan of this thread (the same for every tile group): piece f = (image, y, c, x quad), x quad fastest
  int srest[NV], simg[NV], doff[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int f = tid + i * 512;
    const int xq = f % (W / 4), t1 = f / (W / 4), c = t1 % C, t2 = t1 / C, y = t2 % H, ii = t2 / H;
    simg[i] = ii;
    srest[i] = (c * H + y) * W + xq * 4;
    doff[i] = f < G::F4 ? ii * IMGS + ((y + 1) * PW + xq * 4 + 1) * CP + c : -1;
  }
  float4 pv[NV];
  const int groups = (N + IPT - 1) / IPT;
  auto issue = [&](int g) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int n = g * IPT + (doff[i] >= 0 ? simg[i] : 0);
      n = n < N ? n : N - 1;                                        // (an odd tail re-reads the last image; never stored)
      pv[i] = *reinterpret_cast<const float4*>(x + (int64_t)n * (C * H * W) + (doff[i] >= 0 ? srest[i] : 0));
    }
  };
  auto commit = [&](float* dst) {
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (doff[i] >= 0) {
        float* d = dst + doff[i];
        d[0] = pv[i].x; d[CP] = pv[i].y; d[2 * CP] = pv[i].z; d[3 * CP] = pv[i].w;
      }
  };
  __syncthreads();
  if ((int)blockIdx.x < groups) { issue(blockIdx.x); commit(im); }
  __syncthreads();

  const int iml = W == 16 ? (l31 >> 4) : 0, xl = W == 16 ? (l31 & 15) : l31;
  const float* wrow = wl + l31 * WS + half * 4;
  int buf = 0;
  for (int g = blockIdx.x; g < groups; g += gridDim.x) {
    const int gn = g + gridDim.x;
    if (gn < groups) issue(gn);
    const float* frame = im + buf * (IPT * IMGS);
#pragma unroll 1
    for (int pass = 0; pass < G::PASSES; ++pass) {
      const int rp = pass * 8 + wave;
      if (rp >= G::RP) break;
      const float* lb = frame + iml * IMGS + ((2 * rp) * PW + xl) * CP;          // padded (y0 + kh, x + kw) from here
      f32x16 acc[OT][2];                  // start from the bias: the epilogue adds nothing
#pragma unroll
      for (int o = 0; o < OT; ++o)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[o][0][r] = acc[o][1][r] = bs[o * 32 + qacc_row(r, half)];
      f32x4 a[2][OT], b[2][2];
      auto load = [&](int p, f32x4 (&aa)[OT], f32x4 (&bb)[2]) {
        const int o0 = G::unit_off(p), o1 = p + NP < UNITS ? G::unit_off(p + NP) : o0;
        const float* bp = lb + (half ? o1 : o0);
#pragma unroll
        for (int o = 0; o < OT; ++o) aa[o] = *reinterpret_cast<const f32x4*>(wrow + o * 32 * WS + p * 8);
        bb[0] = *reinterpret_cast<const f32x4*>(bp);
        bb[1] = *reinterpret_cast<const f32x4*>(bp + PW * CP);
      };
      auto mul = [&](f32x4 (&aa)[OT], f32x4 (&bb)[2]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int o = 0; o < OT; ++o)
#pragma unroll
            for (int c = 0; c < 2; ++c)
              acc[o][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[o][j], bb[c][j], acc[o][c], 0, 0, 0);
      };
      load(0, a[0], b[0]);
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        if (p + 1 < NP) load(p + 1, a[(p + 1) & 1], b[(p + 1) & 1]);
        mul(a[p & 1], b[p & 1]);
      }
      // epilogue: bias + relu + 2 x 2 max-pool + hit map (semantics of conv_direct.hip EP = 1: a position is hit when
      // its relu equals the window maximum AND y >= 0 -- ties all pass, relu'(0) = 1, tensor.py:808-815).  lane =
      // position of row y0 (acc[.][0]) and y0 + 1 (acc[.][1]), register = channel: the window is in-lane + lane ^ 1.
      const int n = g * IPT + iml;
      const bool live = n < N;
      constexpr int PM = (H / 2) * (W / 2), MW = H * W / 32;
#pragma unroll
      for (int o = 0; o < OT; ++o) {
        unsigned w0 = 0, w1 = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          // hit <=> v == m: m = max(0, window) >= 0, so equality already implies relu'(v) = [v >= 0] (and -0 == +0)
          const int oc = o * 32 + qacc_row(r, half);
          const float v0 = acc[o][0][r], v1 = acc[o][1][r];
          const float mv = fmaxf(fmaxf(v0, v1), 0.f);
          const float m = fmaxf(mv, dpp_xor1(mv));
          const unsigned long long h0 = __ballot(v0 == m), h1 = __ballot(v1 == m);
          if ((l31 & 15) == r) {
            if (W == 16) {            // a 32-position word = rows y0, y0 + 1 of ONE image: 16 bits of each ballot
              const int sh = 32 * half + 16 * iml;
              w0 = ((unsigned)(h0 >> sh) & 0xFFFFu) | ((unsigned)(h1 >> sh) << 16);
            } else {                  // a word per row
              w0 = (unsigned)(h0 >> (32 * half));
              w1 = (unsigned)(h1 >> (32 * half));
            }
          }
          if (live && !(l31 & 1) && oc < O) pooled[((int64_t)n * O + oc) * PM + rp * (W / 2) + (xl >> 1)] = m;
        }
        const int ocm = o * 32 + qacc_row(l31 & 15, half);
        if (W == 16) {
          if (live && ocm < O) mask[((int64_t)n * O + ocm) * MW + rp] = w0;
        } else {
          if (live && l31 < 16 && ocm < O)
            *reinterpret_cast<uint2*>(mask + ((int64_t)n * O + ocm) * MW + 2 * rp) = make_uint2(w0, w1);
        }
      }
    }
    if (gn < groups) commit(im + (buf ^ 1) * (IPT * IMGS));
    __syncthreads();
    buf ^= 1;
  }
}


// ---- data gradient, col2im style ---------------------------------------------------------------------------------
// dx = conv_transpose(dy, w) with dy = the pooled gradient expanded through the hit map.  As a convolution over dy the
// product has N = C = 20 output columns in a 32-wide tile (62 % useful) and K = 9 O = 450; as the reference writes it
// (functional.py:224-232: dcol = dy @ W, then np.add.at col2im) it is D[(c, tap)][position] = sum_o W[o][c][tap] dy[o][position]
// with 180 rows (six 32-row tiles, 94 % useful) and K = O = 50, followed by a scatter-add of every accumulator into
// dx[c][y + kh - 1][x + kw - 1].  Here:
//   * rows are ordered so that accumulator register r of a lane holds (channel pair, tap) = item 16 T + r and the lanes
//     32-63 the odd channel of the pair: the scatter address is lane base + compile-time constant (+ one plane for the upper
//     half-wave) and every accumulator leaves with ONE ds_add_f32 into a zero-haloed frame of the wave's own half image;
//   * B operands (dy) never touch LDS: lane = position, MFMA s needs dy[o(s, half)][position] = one dword of the pooled
//     gradient + one hit word, loaded straight into registers a tile ahead (12.8 KB + 1.6 KB per image, read once);
//   * A operands (W^T) are [row][o] in LDS, four k-steps per ds_read_b128 (rows 13 slots apart: conflict-free);
//   * two waves share an image (rows 0-7 / 8-15), each with its own 10-row frame; the two frame rows they both
//     contribute to are summed at write-out in a fixed order: bit-reproducible.
template <int C_, int H_, int W_, int O_>
struct QuadDgradGeom {
  static constexpr int C = C_, H = H_, W = W_, O = O_;
  static constexpr int PW = W, FROWS = H / 2 + 2, P = FROWS * PW;       // frame of a half image (row halo only): plane stride
  static constexpr int ITEMS = (C / 2) * 9, RT = (ITEMS + 15) / 16;    // (channel pair, tap) items; 32-row tiles
  static constexpr int KG = O / 8, KS = 4 * KG + (O % 8 ? 1 : 0);       // full groups of 8 dy channels; k-steps
  static constexpr int WSD = (O + 3) / 4 * 4;                          // weight row stride (13 slots for O = 50)
  static constexpr int TILES = H / 4;                                  // 32-position tiles (two rows) per half image
  static constexpr int FRAME = C * P;                                  // floats per wave
  static constexpr int STG = O * (W / 2) + 64;                          // per wave: O pooled rows of a tile + its hit words
  static constexpr int LDS_FLOATS = RT * 32 * WSD + 8 * FRAME + 8 * STG;
  static_assert(W == 16 && (C & 1) == 0 && (H % 4) == 0, "two image rows per tile; channel pairs");
  static_assert(O <= 64 && (STG & 3) == 0, "one lane per dy channel stages its pooled row");
  static_assert(O % 8 == 0 || O % 8 == 2, "k-steps: groups of 8 dy channels + one pair");
  static_assert(((WSD / 4) & 1) == 1, "odd slot count per weight row");
};

template <class G>
__global__ __launch_bounds__(512, 1) void conv_quad_dgrad_kernel(const float* __restrict__ dp, const unsigned* __restrict__ hit,
                                                                 const float* __restrict__ w, float* __restrict__ dx, int N) {
  constexpr int C = G::C, H = G::H, W = G::W, O = G::O, PW = G::PW, P = G::P, RT = G::RT, KG = G::KG, KS = G::KS,
                WSD = G::WSD, TILES = G::TILES, ITEMS = G::ITEMS, FROWS = G::FROWS;
  constexpr int PM = (H / 2) * (W / 2), MW = H * W / 32;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* wl = lds;                          // [RT * 32][WSD]
  float* frames = wl + RT * 32 * WSD;       // [8 waves][C][FROWS][PW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  for (int e = tid; e < RT * 32 * WSD; e += 512) {
    const int row = e / WSD, o = e - row * WSD, T = row >> 5, i = row & 31;
    const int item = 16 * T + 4 * (i >> 3) + (i & 3), odd = (i >> 2) & 1;
    float v = 0.f;
    if (item < ITEMS && o < O) v = w[(o * C + 2 * (item / 9) + odd) * 9 + item % 9];
    wl[e] = v;
  }
  for (int e = tid; e < 8 * G::FRAME; e += 512) frames[e] = 0.f;
  __syncthreads();

  float* frame = frames + wave * G::FRAME;
  const int hw = wave & 1, x = l31 & 15, rowbit = l31 >> 4;
  const float* arow = wl + l31 * WSD + 4 * half;
  float* fl = frame + half * P + rowbit * PW + x;          // + 2 t PW per tile, + (channel pair, kh) constant
  const int groups = (N + 3) / 4;
  // dy operands: lane o (< O) loads the pooled row (W / 2 floats) and the hit word of channel o for one tile -- three
  // load instructions per tile and wave; a load per operand and lane (2 KS of them) kept the texture addresser busier than
  // the matrix pipe once the operands came from HBM instead of the cache (287 us inside the step against 224 us alone).
  // They pass through a private staging area of the wave: written one tile ahead, read back per k-step (lane = position).
  float* stg = frames + 8 * G::FRAME + wave * G::STG;       // [O][W / 2] pooled rows, then [64] hit words
  unsigned* stgm = reinterpret_cast<unsigned*>(stg + O * (W / 2));
  const int ol = lane < O ? lane : O - 1;
  float4 g0, g1;
  unsigned gm;
  auto where = [&](int k, int& img, int& t) {               // tile k of this wave's sequence -> (image, tile); false: none
    const int grp = blockIdx.x + (k / TILES) * (int)gridDim.x;
    img = grp * 4 + (wave >> 1);
    t = k % TILES;
    return grp < groups && img < N;
  };
  auto issue = [&](int k) {                                 // (always issued: an invalid k re-reads tile 0 of a valid image)
    int img, t;
    if (!where(k, img, t)) { img = blockIdx.x * 4 + (wave >> 1); img = img < N ? img : N - 1; t = 0; }
    const int tg = TILES * hw + t;
    const float* src = dp + ((int64_t)img * O + ol) * PM + tg * (W / 2);
    g0 = *reinterpret_cast<const float4*>(src);
    g1 = *reinterpret_cast<const float4*>(src + 4);
    gm = hit[((int64_t)img * O + ol) * MW + tg];
  };
  int kseq = 0;
  issue(0);
  if (lane < O) { *reinterpret_cast<float4*>(stg + lane * (W / 2)) = g0; *reinterpret_cast<float4*>(stg + lane * (W / 2) + 4) = g1; }
  stgm[lane] = gm;
  issue(1);
  const float* bvl = stg + (4 * half) * (W / 2) + (x >> 1);  // + (8 g + j) (W / 2) per k-step
  const unsigned* bml = stgm + 4 * half;
  for (int grp = blockIdx.x; grp < groups; grp += gridDim.x) {
    const int img = grp * 4 + (wave >> 1);
    if (img < N) {
#pragma unroll 1
      for (int t = 0; t < TILES; ++t, ++kseq) {
        float bv[KS];
        {
          float rv[KS];
          unsigned rm[KS];
#pragma unroll
          for (int s = 0; s < 4 * KG; ++s) {
            const int o = 8 * (s >> 2) + (s & 3);
            rv[s] = bvl[o * (W / 2)];
            rm[s] = bml[o];
          }
          if (KS > 4 * KG) {                                  // the last pair of dy channels: one k-step
            rv[KS - 1] = stg[(8 * KG + half) * (W / 2) + (x >> 1)];
            rm[KS - 1] = stgm[8 * KG + half];
          }
#pragma unroll
          for (int s = 0; s < KS; ++s) bv[s] = (rm[s] >> l31) & 1u ? rv[s] : 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);
        // the next tile's rows (loaded a tile ago) take the staging area -- the LDS serves this wave's reads above first --
        // and the tile after that is requested: EXACTLY one prefetch per tile (see the comment in the weight gradient)
        if (lane < O) { *reinterpret_cast<float4*>(stg + lane * (W / 2)) = g0; *reinterpret_cast<float4*>(stg + lane * (W / 2) + 4) = g1; }
        stgm[lane] = gm;
        issue(kseq + 2);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc[RT];
#pragma unroll
        for (int T = 0; T < RT; ++T)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[T][r] = 0.f;
        // A fragments one step ahead of the MFMAs that use them: step i = (row tile T, group g); g = KG is the last
        // pair of dy channels (one float, one MFMA)
        constexpr int SPT = KG + (KS > 4 * KG ? 1 : 0), NS = RT * SPT;
        f32x4 a[2];
        auto load_a = [&](int i, f32x4& aa) {
          const int T = i / SPT, g = i - T * SPT;
          if (g < KG) aa = *reinterpret_cast<const f32x4*>(arow + T * 32 * WSD + 8 * g);
          else aa[0] = arow[T * 32 * WSD + 8 * KG - 3 * half];          // column 8 KG + half
        };
        load_a(0, a[0]);
#pragma unroll
        for (int i = 0; i < NS; ++i) {
          if (i + 1 < NS) load_a(i + 1, a[(i + 1) & 1]);
          const int T = i / SPT, g = i - T * SPT;
          if (g < KG) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i & 1][j], bv[4 * g + j], acc[T], 0, 0, 0);
          } else {
            acc[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i & 1][0], bv[KS - 1], acc[T], 0, 0, 0);
          }
        }
        // col2im: accumulator (T, r) = item 16 T + r -> frame[2 cpair + half][2 t + rowbit + kh][x + kw]
        // col2im.  The three kw taps of a (channel, kh) meet in registers: lane = x, so tap kw belongs to lane
        // x + 1 - kw -- two DPP row shifts (zero shifted in at the image edge: the frame needs no column halo) -- and
        // each (channel, kh) leaves with ONE read-add-write of frame row 2 t + rowbit + kh, in rounds of one kh over the
        // channel pairs (different planes).  ds_add_f32 serialises its lanes (measured: 90 per tile cost 4x the tile's
        // MFMAs); the rounds overlap ACROSS lanes (kh + 1 of the upper row is kh of the lower one), which a per-thread
        // alias analysis cannot see: the accesses are volatile = issued in program order, and the LDS serves one
        // wave's instructions in order.
        volatile float* ft = fl + 2 * t * PW;
        {
#pragma unroll
          for (int kh = 0; kh < 3; ++kh) {
            float sum[C / 2], old[C / 2];
#pragma unroll
            for (int cp = 0; cp < C / 2; ++cp) {
              const int i0 = cp * 9 + 3 * kh, i1 = i0 + 1, i2 = i0 + 2;
              sum[cp] = dpp_row_shl1(acc[i0 / 16][i0 % 16]) + acc[i1 / 16][i1 % 16] + dpp_row_shr1(acc[i2 / 16][i2 % 16]);
            }
#pragma unroll
            for (int cp = 0; cp < C / 2; ++cp) old[cp] = ft[2 * cp * P + kh * PW];
#pragma unroll
            for (int cp = 0; cp < C / 2; ++cp) ft[2 * cp * P + kh * PW] = old[cp] + sum[cp];
          }
        }
      }
    }
    __syncthreads();
    // write-out: frame row f + 1 = image row 8 hw + f; the shared rows (frame rows 8, 9 of the upper wave = rows 0, 1 of
    // the lower one) are summed upper + lower
    if (img < N) {
      const float* other = frames + (wave ^ 1) * G::FRAME;
      float* dxn = dx + (int64_t)img * C * H * W + (H / 2) * hw * W;
      for (int e = lane; e < C * (H / 2) * (W / 4); e += 64) {
        const int xq = e % (W / 4), t1 = e / (W / 4), f = t1 % (H / 2), c = t1 / (H / 2);
        float4 v = *reinterpret_cast<const float4*>(frame + c * P + (f + 1) * PW + 4 * xq);
        if (hw == 0 && f == H / 2 - 1) {            // image row 7 = my frame row 8 + the lower wave's frame row 0
          const float4 u = *reinterpret_cast<const float4*>(other + c * P + 4 * xq);
          v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        } else if (hw == 1 && f == 0) {             // image row 8 = the upper wave's frame row 9 + my frame row 1
          const float4 u = *reinterpret_cast<const float4*>(other + c * P + (FROWS - 1) * PW + 4 * xq);
          v = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
        }
        *reinterpret_cast<float4*>(dxn + (c * H + f) * W + 4 * xq) = v;
      }
    }
    __syncthreads();
    if (img < N)
      for (int e = lane * 4; e < G::FRAME; e += 256) *reinterpret_cast<float4*>(frame + e) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

template <class G>
int launch_dgrad(const float* dp, const unsigned* hit, const float* w, float* dx, int N, hipStream_t st) {
  auto kern = conv_quad_dgrad_kernel<G>;
  constexpr int lds = G::LDS_FLOATS * 4;
  static_assert(lds <= 160 * 1024, "LDS budget");
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) { pdn_set_error("conv_quad: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
  const int groups = (N + 3) / 4;
  hipLaunchKernelGGL(kern, dim3(groups < 256 ? groups : 256), dim3(512), lds, st, dp, hit, w, dx, N);
  PDN_LAUNCH_CHECK();
  pdn_count(PDN_CNT_CONV_QUAD_DGRAD);
  return PDN_OK;
}


// ---- weight gradient -----------------------------------------------------------------------------------------------
// dW[o][(c, tap)] = sum over images and positions of dy[o][pos] x_pad[c][pos + tap]  (+ db[o] = the column of ones): the
// contraction runs over POSITIONS, so "four k-steps per read" means four consecutive x of one image row.
//   * B (x): three copies of the zero-haloed image, SHIFTED by kw = 0, 1, 2 columns, so that the four positions a lane
//     needs for its column (c, kh, kw) are one ALIGNED ds_read_b128 at copy kw, plane c, row y + kh; lane base + compile-
//     time constant, one read per 32-column tile and group of 8 positions (lanes 0-31: x quad 2 g, lanes 32-63: 2 g + 1).
//     The column order inside the tiles is chosen on the host so that the 16 lanes of every ds_read_b128 service group
//     fall into 16 different 16-byte slots (planes padded by one slot: every slot residue holds <= 12 of the 180
//     columns = one per group); the reduce kernel undoes the permutation.  Bias column and padding lanes read a plane of ones.
//   * A (dy) never touches LDS: lane = output channel, its four positions are two floats of the pooled gradient +
//     four bits of the hit word, loaded one image ahead straight into registers and expanded with a few VALU
//     instructions per 8 positions.
//   * a wave owns ONE 32-row tile of output channels, ALL column tiles and a quarter (OT = 2) / an eighth of the image
//     rows; accumulators stay in registers over all images of the workgroup, are summed over the waves through LDS at
//     the end and leave as one slab per workgroup (fixed-order reduction: bit-reproducible).
struct QuadWgradPerm {
  unsigned short col[6 * 32];    // tile lane -> weight column c * 9 + tap, K = bias, 0xFFFF = padding
  unsigned short slot[6 * 32];   // ... and the 16-byte slot of its operand in the shifted-copy frame
};

template <int C_, int H_, int W_, int O_>
struct QuadWgradGeom {
  static constexpr int C = C_, H = H_, W = W_, O = O_;
  static constexpr int K = C * 9, CT = (K + 1 + 31) / 32, OT = (O + 31) / 32, OPAD = OT * 32, KCOLS = CT * 32;
  static constexpr int PQ = 8 / OT, RW = H / PQ;                       // position splits; image rows per wave
  static constexpr int SC = (H + 2) * W + 4, SK = C * SC;              // plane (+ one slot), copy
  static constexpr int BUF = 3 * SK;                                   // one image: three shifted copies
  static constexpr int ONES = 2 * BUF;                                 // plane of ones behind the two buffers
  static constexpr int ONES_FLOATS = H * W + 16 + 64;                  // + 16 slots: a padding lane picks its group's free slot residue
  static constexpr int LDS_FLOATS = 2 * BUF + ONES_FLOATS;
  static constexpr int F4 = C * H * W / 4, NV = (F4 + 511) / 512;
  static constexpr int GPR = W / 8;                                    // groups of 8 positions per image row
  static constexpr int NPR = RW / 2, NW = (W == 16 ? NPR : RW);        // pooled rows / hit words per wave and image
  static_assert(OT == 1 || OT == 2, "one or two 32-row tiles of output channels");
  static_assert(CT <= 6 && (W == 16 || W == 32) && RW >= 2 && (RW & 1) == 0, "shape outside the kernel");
  static_assert(8 * 16 * 64 <= 2 * BUF, "reduction scratch fits the image buffers");
};

template <class G>
__global__ __launch_bounds__(512, 1) void conv_quad_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dp,
                                                                 const unsigned* __restrict__ hit, float* __restrict__ partial,
                                                                 QuadWgradPerm perm, int N) {
  constexpr int C = G::C, H = G::H, W = G::W, O = G::O, CT = G::CT, OT = G::OT, PQ = G::PQ, RW = G::RW, SC = G::SC,
                SK = G::SK, BUF = G::BUF, NV = G::NV, GPR = G::GPR, NPR = G::NPR, NW = G::NW;
  constexpr int PM = (H / 2) * (W / 2), MW = H * W / 32;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const int ot = wave % OT, q = wave / OT, row0 = q * RW;
  for (int e = tid; e < 2 * BUF; e += 512) lds[e] = 0.f;
  for (int e = tid; e < G::ONES_FLOATS; e += 512) lds[G::ONES + e] = 1.f;
  // x staging plan: piece f = (c, y, x quad) in memory order; copy kw holds x_pad[..][x' + kw], x_pad[..][x''] = x[x'' - 1]
  int dst1[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int f = tid + i * 512, xq = f % (W / 4), t1 = f / (W / 4), y = t1 % H, c = t1 / H;
    dst1[i] = f < G::F4 ? SK + c * SC + (y + 1) * W + 4 * xq : -1;            // copy 1: aligned
  }
  float4 pv[NV];
  auto issue_x = [&](int n) {
#pragma unroll
    for (int i = 0; i < NV; ++i)
      pv[i] = *reinterpret_cast<const float4*>(x + (int64_t)n * (C * H * W) + 4 * (dst1[i] >= 0 ? tid + i * 512 : 0));
  };
  auto commit_x = [&](float* b) {
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (dst1[i] >= 0) {
        float* d1 = b + dst1[i];
        *reinterpret_cast<float4*>(d1) = pv[i];
        const int xq = (tid + i * 512) % (W / 4);
        float* d0 = d1 - SK + 1;                       // copy 0: one column to the right (x' = xx + 1)
        d0[0] = pv[i].x; d0[1] = pv[i].y; d0[2] = pv[i].z;
        if (xq != W / 4 - 1) d0[3] = pv[i].w;
        float* d2 = d1 + SK - 1;                       // copy 2: one column to the left (x' = xx - 1)
        if (xq != 0) d2[0] = pv[i].x;
        d2[1] = pv[i].y; d2[2] = pv[i].z; d2[3] = pv[i].w;
      }
  };
  // dy operands of one image: pooled pairs and hit words of this lane's channel, rows row0 .. row0 + RW - 1
  const int oc = ot * 32 + l31, ocl = oc < O ? oc : O - 1;
  float2 dv[NPR][GPR];
  unsigned dm[NW];
  auto issue_dy = [&](int n) {
    const float* dpb = dp + ((int64_t)n * O + ocl) * PM + (row0 >> 1) * (W / 2) + 2 * half;
    const unsigned* hb = hit + ((int64_t)n * O + ocl) * MW + (row0 * W >> 5);
#pragma unroll
    for (int pr = 0; pr < NPR; ++pr)
#pragma unroll
      for (int g = 0; g < GPR; ++g) dv[pr][g] = *reinterpret_cast<const float2*>(dpb + pr * (W / 2) + 4 * g);
#pragma unroll
    for (int i = 0; i < NW; ++i) dm[i] = oc < O ? hb[i] : 0u;
  };
  // B operand bases of this lane's columns (floats, buffer 0)
  int bb[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t) bb[t] = 4 * perm.slot[t * 32 + l31] + row0 * W + 4 * half;

  f32x16 acc[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  __syncthreads();
  int n = blockIdx.x;
  if (n < N) { issue_x(n); issue_dy(n); commit_x(lds); }
  __syncthreads();
  int buf = 0;
  for (; n < N; n += gridDim.x) {
    // operands of THIS image leave their prefetch registers first, then the next image's loads are issued
    float2 cv[NPR][GPR];
    unsigned cm[NW];
#pragma unroll
    for (int pr = 0; pr < NPR; ++pr)
#pragma unroll
      for (int g = 0; g < GPR; ++g) cv[pr][g] = dv[pr][g];
#pragma unroll
    for (int i = 0; i < NW; ++i) cm[i] = dm[i];
    __builtin_amdgcn_sched_barrier(0);
    const int nn = n + gridDim.x < N ? n + gridDim.x : n;          // (always issued: see the data-gradient kernel)
    issue_x(nn);
    issue_dy(nn);
    __builtin_amdgcn_sched_barrier(0);
    const float* fb = lds + buf * BUF;
    // step i = (row yy, position group g, column tile t); the B fragment of step i + 1 is in flight while step i multiplies
    constexpr int NS = RW * GPR * CT;
    f32x4 b[2];
    auto load_b = [&](int i, f32x4& bf) {
      const int t = i % CT, kg = i / CT, g = kg % GPR, yy = kg / GPR;
      bf = *reinterpret_cast<const f32x4*>(fb + bb[t] + yy * W + 8 * g);
    };
    load_b(0, b[0]);
    float a[4];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      // the next fragment's read is issued HERE, in front of this step's MFMAs (hipcc otherwise sinks it behind them
      // into the registers they just freed and waits for it at once: a full LDS round trip per group of four)
      if (i + 1 < NS) load_b(i + 1, b[(i + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      const int t = i % CT, kg = i / CT, g = kg % GPR, yy = kg / GPR;
      if (t == 0) {
        // the lane's four positions (row0 + yy, 4 (2 g + half) + j): bits of the hit word, pooled pair cv[yy / 2][g]
        const unsigned wd = cm[W == 16 ? yy / 2 : yy] >> ((W == 16 ? (yy & 1) * 16 : 0) + 8 * g);
        const unsigned bits = half ? wd >> 4 : wd;
        const float2 p = cv[yy / 2][g];
        a[0] = bits & 1u ? p.x : 0.f; a[1] = bits & 2u ? p.x : 0.f; a[2] = bits & 4u ? p.y : 0.f; a[3] = bits & 8u ? p.y : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[i & 1][j], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (n + (int)gridDim.x < N) commit_x(lds + (buf ^ 1) * BUF);
    __syncthreads();
    buf ^= 1;
  }
  // sum over the PQ waves of an output-channel tile, one column tile per round, through LDS; one slab per workgroup
  float* red = lds;                                    // [8 waves][16][64]
  float* slab = partial + (int64_t)blockIdx.x * G::OPAD * G::KCOLS;
#pragma unroll                                       // (unrolled: a runtime index would put the accumulators in scratch)
  for (int t = 0; t < CT; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[t][r];
    __syncthreads();
    for (int e = tid; e < OT * 16 * 64; e += 512) {
      const int ln = e & 63, r = (e >> 6) & 15, o2 = e >> 10;
      float sum = 0.f;
#pragma unroll
      for (int qq = 0; qq < PQ; ++qq) sum += red[((qq * OT + o2) * 16 + r) * 64 + ln];
      slab[(int64_t)(o2 * 32 + qacc_row(r, ln >> 5)) * G::KCOLS + t * 32 + (ln & 31)] = sum;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void conv_quad_wgrad_reduce_kernel(const float* __restrict__ partial, int slabs, int OPAD,
                                                                     int KCOLS, int O, int K, QuadWgradPerm perm,
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
    for (int qq = 0; qq < 8; ++qq) t += red[qq][threadIdx.x];
    if (j < K) {
      if (dw) dw[(int64_t)o * K + j] = accumulate ? dw[(int64_t)o * K + j] + t : t;
    } else if (db) {
      db[o] = accumulate ? db[o] + t : t;
    }
  }
}

// Column order: weight column (c, kh, kw) reads slot (kw SK + c SC + kh W) / 4 (+ a position term common to all lanes);
// every ds_read_b128 service group (MI355X_MICROARCH.md, LDS: lanes {0-3, 12-15, 20-27} and {4-11, 16-19, 28-31} of each
// half-wave) gets columns of 16 different slot residues where the class sizes allow it.
template <class G>
void build_wgrad_perm(QuadWgradPerm& pm) {
  static const int grp_lanes[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                       {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
  constexpr int NG = 2 * G::CT;
  int fill[NG] = {0};
  unsigned used[NG] = {0};
  for (int i = 0; i < 6 * 32; ++i) { pm.col[i] = 0xFFFF; pm.slot[i] = (unsigned short)(G::ONES / 4); }
  auto slot_of = [](int j) { const int c = j / 9, tap = j % 9; return ((tap % 3) * G::SK + c * G::SC + (tap / 3) * G::W) / 4; };
  auto place = [&](int g, int j, int slot) {
    const int ln = grp_lanes[g & 1][fill[g]++];
    pm.col[(g >> 1) * 32 + ln] = (unsigned short)j;
    pm.slot[(g >> 1) * 32 + ln] = (unsigned short)slot;
    used[g] |= 1u << (slot & 15);
  };
  bool placed[G::K + 1] = {false};
  for (int pass = 0; pass < 2; ++pass)              // conflict-free placements first, the rest where there is room
    for (int j = 0; j < G::K; ++j) {
      if (placed[j]) continue;
      const int slot = slot_of(j);
      int best = -1;
      for (int g = 0; g < NG; ++g)
        if (fill[g] < 16 && (pass == 1 || !(used[g] >> (slot & 15) & 1u)) && (best < 0 || fill[g] < fill[best])) best = g;
      if (best >= 0) { place(best, j, slot); placed[j] = true; }
    }
  // the bias column (a lane of ones) and the padding lanes: they all read the plane of ones, each group's at the 16-byte
  // slot residue its real columns leave free (one shared address per group = a broadcast: no conflict with anybody)
  bool bias_placed = false;
  for (int g = NG - 1; g >= 0; --g) {
    if (fill[g] == 16) continue;
    int free_res = 0;
    while (free_res < 16 && (used[g] >> free_res & 1u)) ++free_res;
    const int base = G::ONES / 4, slot = base + ((free_res - base) % 16 + 16) % 16;
    while (fill[g] < 16) {
      const int ln = grp_lanes[g & 1][fill[g]++];
      pm.col[(g >> 1) * 32 + ln] = (unsigned short)(bias_placed ? 0xFFFF : G::K);
      pm.slot[(g >> 1) * 32 + ln] = (unsigned short)slot;
      bias_placed = true;
    }
  }
}

template <class G>
int launch_wgrad_quad(const float* x, const float* dp, const unsigned* hit, float* dw, float* db, int accumulate, int N,
                      void* workspace, int64_t workspace_bytes, hipStream_t st) {
  auto kern = conv_quad_wgrad_kernel<G>;
  constexpr int lds = G::LDS_FLOATS * 4;
  static_assert(lds <= 160 * 1024, "LDS budget");
  const int grid = N < 256 ? N : 256;
  if (!workspace || workspace_bytes < 4ll * grid * G::OPAD * G::KCOLS) {
    pdn_set_error("conv_quad weight gradient: workspace too small");
    return PDN_EWORKSPACE;
  }
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) { pdn_set_error("conv_quad: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
  static QuadWgradPerm pm;
  static bool built = false;
  if (!built) { build_wgrad_perm<G>(pm); built = true; }
  float* partial = (float*)workspace;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, x, dp, hit, partial, pm, N);
  PDN_LAUNCH_CHECK();
  pdn_count(PDN_CNT_CONV_QUAD_WGRAD);
  const int total = G::O * G::KCOLS;
  hipLaunchKernelGGL(conv_quad_wgrad_reduce_kernel, dim3((total + 31) / 32), dim3(256), 0, st, partial, grid, G::OPAD,
                     G::KCOLS, G::O, G::K, pm, dw, db, accumulate);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

template <class G>
int launch_fwd(const float* x, const float* w, const float* bias, float* pooled, unsigned* mask, int N, hipStream_t st) {
  auto kern = conv_quad_fwd_kernel<G>;
  constexpr int lds = G::LDS_FLOATS * 4;
  static_assert(lds <= 160 * 1024, "LDS budget");
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) { pdn_set_error("conv_quad: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
  const int groups = (N + G::IPT - 1) / G::IPT;
  const int grid = groups < 256 * G::WGS ? groups : 256 * G::WGS;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, x, w, bias, pooled, mask, N);
  PDN_LAUNCH_CHECK();
  pdn_count(PDN_CNT_CONV_QUAD_FWD);
  return PDN_OK;
}

bool quad_enabled() {
  static const int on = getenv("PDN_CONV_QUAD") ? atoi(getenv("PDN_CONV_QUAD")) : 1;
  return on != 0;
}

}  // namespace

bool conv_quad_fwd_supported(int C, int H, int W, int O, int k, int stride, int pad) {
  if (!quad_enabled() || k != 3 || stride != 1 || pad != 1) return false;
  return (C == 20 && H == 16 && W == 16 && O == 50) || (C == 3 && H == 32 && W == 32 && O == 20);
}

bool conv_quad_dgrad_supported(int C, int H, int W, int O, int k, int stride, int pad) {
  if (!quad_enabled() || k != 3 || stride != 1 || pad != 1) return false;
  return C == 20 && H == 16 && W == 16 && O == 50;
}

int conv_quad_relu_pool_bwd_data(const float* dpooled, const unsigned* mask, const float* w, float* dx, int N, int C, int H,
                                 int W, int O, void* stream) {
  if (C == 20 && H == 16 && W == 16 && O == 50)
    return launch_dgrad<QuadDgradGeom<20, 16, 16, 50>>(dpooled, mask, w, dx, N, (hipStream_t)stream);
  pdn_set_error("conv_quad_relu_pool_bwd_data: no instantiation for this shape");
  return PDN_EUNSUPPORTED;
}

bool conv_quad_wgrad_supported(int C, int H, int W, int O, int k, int stride, int pad) {
  if (!quad_enabled() || k != 3 || stride != 1 || pad != 1) return false;
  return (C == 20 && H == 16 && W == 16 && O == 50) || (C == 3 && H == 32 && W == 32 && O == 20);
}

int conv_quad_relu_pool_bwd_weight(const float* x, const float* dpooled, const unsigned* mask, float* dw, float* db,
                                   int accumulate, int N, int C, int H, int W, int O, void* workspace,
                                   int64_t workspace_bytes, void* stream) {
  if (C == 20 && H == 16 && W == 16 && O == 50)
    return launch_wgrad_quad<QuadWgradGeom<20, 16, 16, 50>>(x, dpooled, mask, dw, db, accumulate, N, workspace,
                                                            workspace_bytes, (hipStream_t)stream);
  if (C == 3 && H == 32 && W == 32 && O == 20)
    return launch_wgrad_quad<QuadWgradGeom<3, 32, 32, 20>>(x, dpooled, mask, dw, db, accumulate, N, workspace,
                                                           workspace_bytes, (hipStream_t)stream);
  pdn_set_error("conv_quad_relu_pool_bwd_weight: no instantiation for this shape");
  return PDN_EUNSUPPORTED;
}

int conv_quad_relu_pool_fwd(const float* x, const float* w, const float* bias, float* pooled, unsigned* mask, int N, int C,
                            int H, int W, int O, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (C == 20 && H == 16 && W == 16 && O == 50) return launch_fwd<QuadGeom<20, 16, 16, 50>>(x, w, bias, pooled, mask, N, st);
  if (C == 3 && H == 32 && W == 32 && O == 20) return launch_fwd<QuadGeom<3, 32, 32, 20>>(x, w, bias, pooled, mask, N, st);
  pdn_set_error("conv_quad_relu_pool_fwd: no instantiation for this shape");
  return PDN_EUNSUPPORTED;
}
