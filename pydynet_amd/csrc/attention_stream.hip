// General fused attention (forward + backward) for gfx950, fp32 MFMA: any sequence lengths, head
// dims 16 / 24 / 32 / 48 / 64 / 96 / 128, causal (with a start position) and / or additive masks,
// separate query and key/value strides -- key tiles STREAM through LDS with an online softmax, so
// nothing is bounded by what fits on chip and nothing of size Lq x Lk touches HBM.
//
// Serves every attention site of the reference that the resident-K/V kernels of attention.hip
// (head_dim 48, L <= 256: the benchmark shape) do not:
//   llm/llama/model.py:95-121        prompts longer than 256, other head dims, KV-cache prefill
//                                     (keys = cache[:, :start_pos + L], causal with start_pos)
//   llm/clip/model.py:35-63          biased MHA, head_dim 64, non-causal (image) / causal mask (text)
//   examples/pydynet/transformer.py:53-130   head_dim 128, additive padding mask (B, 1, 1, L)
// Math, as the reference composes it: S = q k^T / sqrt(hd) (+ mask), P = softmax(S, -1), O = P v.
//
// Structure (shared by the three kernels): a workgroup = 4 wave64 = 4 tiles of 32 rows of ONE
// (batch, head); the other operand arrives in 32-row tiles through a double-buffered LDS ring filled by
// register-staged loads issued one tile ahead.  Score tiles are computed TRANSPOSED (S^T = K Q^T): a
// lane owns one query, its 16 accumulator registers are 16 keys, so the softmax reductions are
// in-lane plus one cross-half shuffle, the online rescale of O^T is one per-lane scalar, and the P^T
// accumulators feed the next MFMA (O^T += V^T P^T) as they are, with no LDS round trip
// (v_mfma_f32_32x32x2_f32: the two half-waves contract over keys krow(r) and krow(r) + 4).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define AS_LD(hd) ((hd) + 4)

struct AttnArgs {
  int H, Lq, Lk, causal, start_pos;
  int64_t q_rs, q_bs, kv_rs, kv_bs;            // row / batch strides (floats) of q,o,dq and of k,v,dk,dv
  float inv_sqrt;
  const float* mask;                            // additive, element (b, h, q, k) at b*m_b + h*m_h + q*m_q + k*m_k
  int64_t m_b, m_h, m_q, m_k;
  const float* rc;                              // RoPE tables (positions x hd/2), rotate q and k rows on load
  const float* rs;
};

__device__ __forceinline__ int as_krow(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

__device__ __forceinline__ float4 as_rot(float4 v, const float* __restrict__ cs, const float* __restrict__ sn,
                                         int pos, int pair0, int half, float sign) {
  const float2 c = *reinterpret_cast<const float2*>(cs + (int64_t)pos * half + pair0);
  float2 s = *reinterpret_cast<const float2*>(sn + (int64_t)pos * half + pair0);
  s.x *= sign; s.y *= sign;
  float4 o;
  o.x = v.x * c.x - v.y * s.x; o.y = v.x * s.x + v.y * c.x;
  o.z = v.z * c.y - v.w * s.y; o.w = v.z * s.y + v.w * c.y;
  return o;
}

// Register-staged loader of one 32-row tile of TWO [rows][HD] matrices (row stride rs floats) into two
// padded LDS images [32][HD+4].  Rows >= n_rows read as zero.  rot0: rotate matrix 0 rows (RoPE).
template <int HD>
struct TileLoader {
  static constexpr int F4 = HD / 4, PIECES = 32 * F4, NP = (PIECES + 255) / 256;
  float4 r0[NP], r1[NP];
  __device__ __forceinline__ void issue(const float* __restrict__ g0, const float* __restrict__ g1, int64_t rs,
                                        int row0, int n_rows, const float* __restrict__ rc,
                                        const float* __restrict__ rsn, bool rot0) {
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int u = threadIdx.x + 256 * j;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
      if (u < PIECES) {
        const int row = u / F4, c4 = u - row * F4;
        if (row0 + row < n_rows) {
          a = *reinterpret_cast<const float4*>(g0 + (int64_t)(row0 + row) * rs + 4 * c4);
          c = *reinterpret_cast<const float4*>(g1 + (int64_t)(row0 + row) * rs + 4 * c4);
          if (rc && rot0) a = as_rot(a, rc, rsn, row0 + row, 2 * c4, HD / 2, 1.f);
        }
      }
      r0[j].x = a.x; r0[j].y = a.y; r0[j].z = a.z; r0[j].w = a.w;
      r1[j].x = c.x; r1[j].y = c.y; r1[j].z = c.z; r1[j].w = c.w;
    }
  }
  __device__ __forceinline__ void commit(float* __restrict__ s0, float* __restrict__ s1) {
    constexpr int LD = AS_LD(HD);
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int u = threadIdx.x + 256 * j;
      if (u < PIECES) {
        const int row = u / F4, c4 = u - row * F4;
        *reinterpret_cast<float4*>(s0 + row * LD + 4 * c4) = r0[j];
        *reinterpret_cast<float4*>(s1 + row * LD + 4 * c4) = r1[j];
      }
    }
  }
};

// additive terms of one score register: -inf for keys past the end / after the query (causal), plus the mask
__device__ __forceinline__ float as_bias(const AttnArgs& a, const float* __restrict__ mrow, int q, int key) {
  if (key >= a.Lk || (a.causal && key > q + a.start_pos)) return -INFINITY;
  return mrow ? mrow[(int64_t)key * a.m_k] : 0.f;
}

// number of 32-key tiles a block of queries [q0, q1) can see
__device__ __forceinline__ int as_key_tiles(const AttnArgs& a, int q1) {
  int last = a.Lk - 1;
  if (a.causal) last = min(last, q1 - 1 + a.start_pos);
  return last < 0 ? 0 : last / 32 + 1;
}

// ---- forward ------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256) void attn_fwd_stream_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                              const float* __restrict__ V, float* __restrict__ O,
                                                              float* __restrict__ LSE, AttnArgs a) {
  constexpr int LD = AS_LD(HD), NT8 = HD / 8, F4 = HD / 4, DT = (HD + 31) / 32;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  auto Kbuf = [&](int i) { return lds + (i & 1) * (2 * 32 * LD); };
  auto Vbuf = [&](int i) { return lds + (i & 1) * (2 * 32 * LD) + 32 * LD; };
  const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
  const int q0 = blockIdx.x * 128, qt0 = q0 + wave * 32;
  const float* Qb = Q + (int64_t)b * a.q_bs + (int64_t)h * HD;
  const float* Kg = K + (int64_t)b * a.kv_bs + (int64_t)h * HD;
  const float* Vg = V + (int64_t)b * a.kv_bs + (int64_t)h * HD;
  float* Ob = O + (int64_t)b * a.q_bs + (int64_t)h * HD;
  const int qpos = qt0 + li, qc = min(qpos, a.Lq - 1);
  const float* mrow = a.mask ? a.mask + (int64_t)b * a.m_b + (int64_t)h * a.m_h + (int64_t)qc * a.m_q : nullptr;

  const int nkt = as_key_tiles(a, min(q0 + 128, a.Lq));
  const int my_nkt = qt0 < a.Lq ? as_key_tiles(a, min(qt0 + 32, a.Lq)) : 0;     // wave-uniform
  TileLoader<HD> ld;
  if (nkt > 0) {
    ld.issue(Kg, Vg, a.kv_rs, 0, a.Lk, a.rc, a.rs, true);
    ld.commit(Kbuf(0), Vbuf(0));
  }
  // Q fragments: lane (li, lh) holds Q[q][8t + 4lh .. +3]
  float4 qf[NT8];
  {
    const float* qrow = Qb + (int64_t)qc * a.q_rs + 4 * lh;
#pragma unroll
    for (int t = 0; t < NT8; ++t) {
      qf[t] = *reinterpret_cast<const float4*>(qrow + 8 * t);
      if (a.rc) qf[t] = as_rot(qf[t], a.rc, a.rs, qc + a.start_pos, 4 * t + 2 * lh, HD / 2, 1.f);
    }
  }
  f32x16 o[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m = -INFINITY, l = 0.f;
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) ld.issue(Kg, Vg, a.kv_rs, (kt + 1) * 32, a.Lk, a.rc, a.rs, true);
    if (kt < my_nkt) {
      const float* Ks = Kbuf(kt);
      const float* Vs = Vbuf(kt);
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
      const float* krow = Ks + li * LD + 4 * lh;
#pragma unroll
      for (int t = 0; t < NT8; ++t) {
        const float4 kf = *reinterpret_cast<const float4*>(krow + 8 * t);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[t].x, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[t].y, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[t].z, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[t].w, s, 0, 0, 0);
      }
      float tm = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = s[r] * a.inv_sqrt + as_bias(a, mrow, qpos, kt * 32 + as_krow(r, lh));
        s[r] = v;
        tm = fmaxf(tm, v);
      }
      tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
      const float mn = fmaxf(m, tm);
      // rows that have seen only -inf so far keep p = 0 and an unscaled (zero) accumulator
      const float alpha = mn == -INFINITY ? 1.f : __expf(m - mn);
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = mn == -INFINITY ? 0.f : __expf(s[r] - mn);
        s[r] = p;
        ps += p;
      }
      l = l * alpha + ps;
      m = mn;
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* vrow = Vs + as_krow(r, lh) * LD + li;
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          const float av = (d * 32 + 32 <= HD || d * 32 + li < HD) ? vrow[d * 32] : 0.f;
          o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, s[r], o[d], 0, 0, 0);
        }
      }
    }
    if (kt + 1 < nkt) ld.commit(Kbuf(kt + 1), Vbuf(kt + 1));
    __syncthreads();
  }
  // ---- normalise, stage [q][d] through this wave's quarter of the ring, store rows coalesced -------
  l += __shfl_xor(l, 32, 64);
  const float inv_l = 1.f / l;                      // a fully masked row gives 0 * inf = NaN, as the reference does
  float* Ow = lds + wave * 32 * LD;
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dd = d * 32 + as_krow(r, lh);
      if (dd < HD) Ow[li * LD + dd] = o[d][r] * inv_l;
    }
  if (lh == 0 && qpos < a.Lq) LSE[(int64_t)bh * a.Lq + qpos] = m + logf(l);
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  for (int u = lane; u < 32 * F4; u += 64) {
    const int row = u / F4, c4 = u - row * F4;
    if (qt0 + row < a.Lq)
      *reinterpret_cast<float4*>(Ob + (int64_t)(qt0 + row) * a.q_rs + 4 * c4) =
          *reinterpret_cast<const float4*>(Ow + row * LD + 4 * c4);
  }
}

// ---- backward: dQ (and delta) ---------------------------------------------------------------------
// X^T[d][row] accumulators (lane = row, registers = d) -> [row][d] through LDS -> coalesced row stores
template <int HD, int DT>
__device__ __forceinline__ void as_store_T(float* __restrict__ slot, const f32x16 (&t)[DT], float* __restrict__ dst,
                                           int64_t rs, int row0, int n_rows, int li, int lh, int lane,
                                           const float* __restrict__ rc, const float* __restrict__ rsn, int pos0) {
  constexpr int LD = AS_LD(HD), F4 = HD / 4;
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dd = d * 32 + as_krow(r, lh);
      if (dd < HD) slot[li * LD + dd] = t[d][r];
    }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  for (int u = lane; u < 32 * F4; u += 64) {
    const int row = u / F4, c4 = u - row * F4;
    if (row0 + row < n_rows) {
      float4 v = *reinterpret_cast<const float4*>(slot + row * LD + 4 * c4);
      if (rc) v = as_rot(v, rc, rsn, pos0 + row0 + row, 2 * c4, HD / 2, -1.f);     // gradient of RoPE
      *reinterpret_cast<float4*>(dst + (int64_t)(row0 + row) * rs + 4 * c4) = v;
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
}

template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_dq_stream_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
    const float* __restrict__ O, const float* __restrict__ dO, const float* __restrict__ LSE,
    float* __restrict__ dQ, float* __restrict__ Delta, AttnArgs a) {
  constexpr int LD = AS_LD(HD), NT8 = HD / 8, DT = (HD + 31) / 32;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  auto Kbuf = [&](int i) { return lds + (i & 1) * (2 * 32 * LD); };
  auto Vbuf = [&](int i) { return lds + (i & 1) * (2 * 32 * LD) + 32 * LD; };
  const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
  const int q0 = blockIdx.x * 128, qt0 = q0 + wave * 32;
  const int64_t qbase = (int64_t)b * a.q_bs + (int64_t)h * HD;
  const float* Kg = K + (int64_t)b * a.kv_bs + (int64_t)h * HD;
  const float* Vg = V + (int64_t)b * a.kv_bs + (int64_t)h * HD;
  const int qpos = qt0 + li, qc = min(qpos, a.Lq - 1);
  const float* mrow = a.mask ? a.mask + (int64_t)b * a.m_b + (int64_t)h * a.m_h + (int64_t)qc * a.m_q : nullptr;
  const int nkt = as_key_tiles(a, min(q0 + 128, a.Lq));
  const int my_nkt = qt0 < a.Lq ? as_key_tiles(a, min(qt0 + 32, a.Lq)) : 0;
  TileLoader<HD> ld;
  if (nkt > 0) {
    ld.issue(Kg, Vg, a.kv_rs, 0, a.Lk, a.rc, a.rs, true);
    ld.commit(Kbuf(0), Vbuf(0));
  }
  float4 qf[NT8], gf[NT8];
  float dpart = 0.f;
  {
    const float* qrow = Q + qbase + (int64_t)qc * a.q_rs + 4 * lh;
    const float* grow = dO + qbase + (int64_t)qc * a.q_rs + 4 * lh;
    const float* orow = O + qbase + (int64_t)qc * a.q_rs + 4 * lh;
#pragma unroll
    for (int t = 0; t < NT8; ++t) {
      qf[t] = *reinterpret_cast<const float4*>(qrow + 8 * t);
      if (a.rc) qf[t] = as_rot(qf[t], a.rc, a.rs, qc + a.start_pos, 4 * t + 2 * lh, HD / 2, 1.f);
      gf[t] = *reinterpret_cast<const float4*>(grow + 8 * t);
      const float4 ov = *reinterpret_cast<const float4*>(orow + 8 * t);
      dpart += (ov.x * gf[t].x + ov.y * gf[t].y) + (ov.z * gf[t].z + ov.w * gf[t].w);
    }
  }
  const float delta_q = dpart + __shfl_xor(dpart, 32, 64);
  const float lse_q = LSE[(int64_t)bh * a.Lq + qc];
  if (lh == 0 && qpos < a.Lq) Delta[(int64_t)bh * a.Lq + qpos] = delta_q;
  f32x16 dq[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[d][r] = 0.f;
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) ld.issue(Kg, Vg, a.kv_rs, (kt + 1) * 32, a.Lk, a.rc, a.rs, true);
    if (kt < my_nkt) {
      const float* Ks = Kbuf(kt);
      const float* Vs = Vbuf(kt);
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
      const float* krow = Ks + li * LD + 4 * lh;
      const float* vrow = Vs + li * LD + 4 * lh;
#pragma unroll
      for (int t = 0; t < NT8; ++t) {
        const float4 kf = *reinterpret_cast<const float4*>(krow + 8 * t);
        const float4 vf = *reinterpret_cast<const float4*>(vrow + 8 * t);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[t].x, s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.x, gf[t].x, dp, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[t].y, s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.y, gf[t].y, dp, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[t].z, s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.z, gf[t].z, dp, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[t].w, s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.w, gf[t].w, dp, 0, 0, 0);
      }
      // dS^T[key][q] = P^T o (dP^T - delta_q) / sqrt(hd)   (lane = q)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float bias = as_bias(a, mrow, qpos, kt * 32 + as_krow(r, lh));
        const float p = bias == -INFINITY ? 0.f : __expf(s[r] * a.inv_sqrt + bias - lse_q);
        s[r] = p * (dp[r] - delta_q) * a.inv_sqrt;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* kr = Ks + as_krow(r, lh) * LD + li;
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          const float av = (d * 32 + 32 <= HD || d * 32 + li < HD) ? kr[d * 32] : 0.f;
          dq[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, s[r], dq[d], 0, 0, 0);
        }
      }
    }
    if (kt + 1 < nkt) ld.commit(Kbuf(kt + 1), Vbuf(kt + 1));
    __syncthreads();
  }
  as_store_T<HD, DT>(lds + wave * 32 * LD, dq, dQ + qbase, a.q_rs, qt0, a.Lq, li, lh, lane, a.rc, a.rs, a.start_pos);
}

// ---- backward: dK, dV ------------------------------------------------------------------------------
// A workgroup owns 4 key tiles (K, V fragments in registers); query tiles (Q, dO rows + their lse and
// delta) stream through the ring.  S[q][key] = Q K^T: lane = key, registers = queries.
template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_dkv_stream_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
    const float* __restrict__ dO, const float* __restrict__ LSE, const float* __restrict__ Delta,
    float* __restrict__ dK, float* __restrict__ dV, AttnArgs a) {
  constexpr int LD = AS_LD(HD), NT8 = HD / 8, DT = (HD + 31) / 32;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  auto Qbuf = [&](int i) { return lds + (i & 1) * (2 * 32 * LD); };
  auto Gbuf = [&](int i) { return lds + (i & 1) * (2 * 32 * LD) + 32 * LD; };
  float* stat = lds + 4 * 32 * LD;                  // [2 buffers][lse 32 | delta 32]
  const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
  const int k0 = blockIdx.x * 128, kt0 = k0 + wave * 32;
  const int64_t qbase = (int64_t)b * a.q_bs + (int64_t)h * HD, kbase = (int64_t)b * a.kv_bs + (int64_t)h * HD;
  const float* Qg = Q + qbase;
  const float* Gg = dO + qbase;
  const int kpos = kt0 + li, kc = min(kpos, a.Lk - 1);
  const float* mcol = a.mask ? a.mask + (int64_t)b * a.m_b + (int64_t)h * a.m_h + (int64_t)kc * a.m_k : nullptr;
  // query tiles that can see this block's keys: causal -> q + start_pos >= k0
  const int nqt = (a.Lq + 31) / 32;
  int qt_first = 0;
  if (a.causal) qt_first = max(0, k0 - a.start_pos) / 32;
  const int my_first = a.causal ? max(0, kt0 - a.start_pos) / 32 : 0;          // wave-uniform
  const bool active = kt0 < a.Lk;
  TileLoader<HD> ld;
  auto load_stats = [&](int qt, int buf) {
    if (threadIdx.x < 64) {
      const int q = qt * 32 + (threadIdx.x & 31);
      const float* src = threadIdx.x < 32 ? LSE : Delta;
      stat[buf * 64 + threadIdx.x] = q < a.Lq ? src[(int64_t)bh * a.Lq + q] : 0.f;
    }
  };
  if (qt_first < nqt) {
    ld.issue(Qg, Gg, a.q_rs, qt_first * 32, a.Lq, a.rc, a.rs, true);
    ld.commit(Qbuf(0), Gbuf(0));
    load_stats(qt_first, 0);
  }
  float4 kf[NT8], vf[NT8];
  {
    const float* krow = K + kbase + (int64_t)kc * a.kv_rs + 4 * lh;
    const float* vrow = V + kbase + (int64_t)kc * a.kv_rs + 4 * lh;
#pragma unroll
    for (int t = 0; t < NT8; ++t) {
      kf[t] = *reinterpret_cast<const float4*>(krow + 8 * t);
      if (a.rc) kf[t] = as_rot(kf[t], a.rc, a.rs, kc, 4 * t + 2 * lh, HD / 2, 1.f);
      vf[t] = *reinterpret_cast<const float4*>(vrow + 8 * t);
    }
  }
  f32x16 dk[DT], dv[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[d][r] = 0.f; dv[d][r] = 0.f; }
  __syncthreads();
  for (int qt = qt_first, it = 0; qt < nqt; ++qt, ++it) {
    if (qt + 1 < nqt) ld.issue(Qg, Gg, a.q_rs, (qt + 1) * 32, a.Lq, a.rc, a.rs, true);
    if (active && qt >= my_first) {
      const float* Qs = Qbuf(it);
      const float* Gs = Gbuf(it);
      const float* st = stat + (it & 1) * 64;
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
      const float* qrow = Qs + li * LD + 4 * lh;
      const float* grow = Gs + li * LD + 4 * lh;
#pragma unroll
      for (int t = 0; t < NT8; ++t) {
        const float4 q4 = *reinterpret_cast<const float4*>(qrow + 8 * t);
        const float4 g4 = *reinterpret_cast<const float4*>(grow + 8 * t);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.x, kf[t].x, s, 0, 0, 0);      // S[q][key]
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(g4.x, vf[t].x, dp, 0, 0, 0);    // dP[q][key]
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.y, kf[t].y, s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(g4.y, vf[t].y, dp, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.z, kf[t].z, s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(g4.z, vf[t].z, dp, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.w, kf[t].w, s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(g4.w, vf[t].w, dp, 0, 0, 0);
      }
      // lane = key, registers = queries
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qr = as_krow(r, lh), q = qt * 32 + qr;
        float bias = (q >= a.Lq || kpos >= a.Lk || (a.causal && kpos > q + a.start_pos)) ? -INFINITY : 0.f;
        if (mcol && bias == 0.f) bias = mcol[(int64_t)q * a.m_q];
        const float p = bias == -INFINITY ? 0.f : __expf(s[r] * a.inv_sqrt + bias - st[qr]);
        s[r] = p;                                               // P[q][key]
        dp[r] = p * (dp[r] - st[32 + qr]) * a.inv_sqrt;         // dS[q][key]
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qr = as_krow(r, lh);
        const float* gr = Gs + qr * LD + li;
        const float* qq = Qs + qr * LD + li;
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          const bool ok = d * 32 + 32 <= HD || d * 32 + li < HD;
          const float g0 = ok ? gr[d * 32] : 0.f, q0v = ok ? qq[d * 32] : 0.f;
          dv[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(g0, s[r], dv[d], 0, 0, 0);      // dV^T += dO^T P
          dk[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(q0v, dp[r], dk[d], 0, 0, 0);    // dK^T += Q^T dS
        }
      }
    }
    if (qt + 1 < nqt) {
      ld.commit(Qbuf(it + 1), Gbuf(it + 1));
      load_stats(qt + 1, (it + 1) & 1);
    }
    __syncthreads();
  }
  float* slot = lds + wave * 32 * LD;
  as_store_T<HD, DT>(slot, dk, dK + kbase, a.kv_rs, kt0, a.Lk, li, lh, lane, a.rc, a.rs, 0);
  as_store_T<HD, DT>(slot, dv, dV + kbase, a.kv_rs, kt0, a.Lk, li, lh, lane, nullptr, nullptr, 0);
}

// ---- host ---------------------------------------------------------------------------------------------
namespace {
int64_t stream_lds(int hd) { return 4ll * (4 * 32 * AS_LD(hd) + 128); }

bool hd_ok(int hd) { return hd == 16 || hd == 24 || hd == 32 || hd == 48 || hd == 64 || hd == 96 || hd == 128; }

#define AS_DISPATCH(HDV, BODY)                       \
  switch (HDV) {                                     \
    case 16: { constexpr int HD = 16; BODY; } break; \
    case 24: { constexpr int HD = 24; BODY; } break; \
    case 32: { constexpr int HD = 32; BODY; } break; \
    case 48: { constexpr int HD = 48; BODY; } break; \
    case 64: { constexpr int HD = 64; BODY; } break; \
    case 96: { constexpr int HD = 96; BODY; } break; \
    default: { constexpr int HD = 128; BODY; } break; \
  }

int check_common(const char* who, int head_dim, int64_t q_rs, int64_t q_bs, int64_t kv_rs, int64_t kv_bs,
                 const float* rc, const float* rs, int start_pos, uintptr_t ptr_or) {
  if (!hd_ok(head_dim)) {
    pdn_set_error("%s: head_dim %d not in {16, 24, 32, 48, 64, 96, 128}", who, head_dim);
    return PDN_EUNSUPPORTED;
  }
  if ((q_rs % 4) || (q_bs % 4) || (kv_rs % 4) || (kv_bs % 4) || (ptr_or & 15)) {
    pdn_set_error("%s: 16-byte alignment of operands and strides required", who);
    return PDN_EINVAL;
  }
  if ((rc == nullptr) != (rs == nullptr) || ((((uintptr_t)rc | (uintptr_t)rs) & 7) != 0)) {
    pdn_set_error("%s: rope tables must come as an 8-byte aligned pair", who);
    return PDN_EINVAL;
  }
  if (rc && start_pos != 0) {
    pdn_set_error("%s: RoPE-in-load needs start_pos == 0 (cached keys are already rotated)", who);
    return PDN_EUNSUPPORTED;
  }
  return PDN_OK;
}
}  // namespace

extern "C" {

int64_t pdn_attention_stream_bwd_workspace_bytes(int B, int H, int Lq) { return 4ll * B * H * Lq; }

int pdn_attention_stream_supported(int head_dim) { return hd_ok(head_dim) ? 1 : 0; }

/* q, o: (B, Lq, H, hd) through (q_row_stride, q_batch_stride); k, v: (B, Lk, H, hd) through the kv
 * strides; lse: (B, H, Lq).  causal: key > query + start_pos masked.  mask: additive, element
 * (b, h, q, k) at b*mask_sb + h*mask_sh + q*mask_sq + k*mask_sk (0 strides broadcast), NULL = none.
 * rope_cos / rope_sin: optional (positions, hd/2) tables applied to q (position q + start_pos) and k. */
int pdn_attention_stream_fwd_f32(const float* q, const float* k, const float* v, float* o, float* lse, int B,
                                 int H, int Lq, int Lk, int head_dim, int64_t q_row_stride,
                                 int64_t q_batch_stride, int64_t kv_row_stride, int64_t kv_batch_stride,
                                 int causal, int start_pos, const float* mask, int64_t mask_sb,
                                 int64_t mask_sh, int64_t mask_sq, int64_t mask_sk, const float* rope_cos,
                                 const float* rope_sin, void* stream) {
  if (B == 0 || H == 0 || Lq == 0) return PDN_OK;
  PDN_CHECK_ARG(q && k && v && o && lse && Lk > 0, "pdn_attention_stream_fwd_f32: bad operand");
  int rc = check_common("pdn_attention_stream_fwd_f32", head_dim, q_row_stride, q_batch_stride, kv_row_stride,
                        kv_batch_stride, rope_cos, rope_sin, start_pos,
                        (uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o);
  if (rc) return rc;
  AttnArgs a{H, Lq, Lk, causal, start_pos, q_row_stride, q_batch_stride, kv_row_stride, kv_batch_stride,
             1.f / sqrtf((float)head_dim), mask, mask_sb, mask_sh, mask_sq, mask_sk, rope_cos, rope_sin};
  const dim3 grid((Lq + 127) / 128, B * H);
  AS_DISPATCH(head_dim, {
    auto kern = attn_fwd_stream_kernel<HD>;
    PDN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)stream_lds(HD)));
    pdn_count(PDN_CNT_ATT_STREAM);
    hipLaunchKernelGGL(kern, grid, dim3(256), stream_lds(HD), (hipStream_t)stream, q, k, v, o, lse, a);
  });
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

int pdn_attention_stream_bwd_f32(const float* q, const float* k, const float* v, const float* o, const float* d_o,
                                 const float* lse, float* dq, float* dk, float* dv, int B, int H, int Lq, int Lk,
                                 int head_dim, int64_t q_row_stride, int64_t q_batch_stride,
                                 int64_t kv_row_stride, int64_t kv_batch_stride, int causal, int start_pos,
                                 const float* mask, int64_t mask_sb, int64_t mask_sh, int64_t mask_sq,
                                 int64_t mask_sk, const float* rope_cos, const float* rope_sin, void* workspace,
                                 int64_t workspace_bytes, void* stream) {
  if (B == 0 || H == 0 || Lq == 0) return PDN_OK;
  PDN_CHECK_ARG(q && k && v && o && d_o && lse && dq && dk && dv && Lk > 0, "pdn_attention_stream_bwd_f32: bad operand");
  int rc = check_common("pdn_attention_stream_bwd_f32", head_dim, q_row_stride, q_batch_stride, kv_row_stride,
                        kv_batch_stride, rope_cos, rope_sin, start_pos,
                        (uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)d_o |
                            (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv);
  if (rc) return rc;
  if (!workspace || workspace_bytes < pdn_attention_stream_bwd_workspace_bytes(B, H, Lq)) {
    pdn_set_error("pdn_attention_stream_bwd_f32: workspace too small");
    return PDN_EWORKSPACE;
  }
  float* delta = (float*)workspace;
  AttnArgs a{H, Lq, Lk, causal, start_pos, q_row_stride, q_batch_stride, kv_row_stride, kv_batch_stride,
             1.f / sqrtf((float)head_dim), mask, mask_sb, mask_sh, mask_sq, mask_sk, rope_cos, rope_sin};
  hipStream_t st = (hipStream_t)stream;
  AS_DISPATCH(head_dim, {
    auto kq = attn_bwd_dq_stream_kernel<HD>;
    auto kkv = attn_bwd_dkv_stream_kernel<HD>;
    PDN_HIP(hipFuncSetAttribute((const void*)kq, hipFuncAttributeMaxDynamicSharedMemorySize, (int)stream_lds(HD)));
    PDN_HIP(hipFuncSetAttribute((const void*)kkv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)stream_lds(HD)));
    pdn_count(PDN_CNT_ATT_STREAM);
    hipLaunchKernelGGL(kq, dim3((Lq + 127) / 128, B * H), dim3(256), stream_lds(HD), st, q, k, v, o, d_o, lse, dq,
                       delta, a);
    hipLaunchKernelGGL(kkv, dim3((Lk + 127) / 128, B * H), dim3(256), stream_lds(HD), st, q, k, v, d_o, lse, delta,
                       dk, dv, a);
  });
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

}  // extern "C"
