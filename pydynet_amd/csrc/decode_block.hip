// Attention half of a decode layer in ONE launch (llm/llama/model.py:105-121 Attention.forward on one new token with
// the KV cache, called from TransformerBlock.forward, model.py:118-121):
//     x   = previous block's h + its feed-forward records        <- decode_stage.h hand-off (plain records)
//     n   = RMSNorm(x);  q | k | v = n @ [Wq | Wk | Wv];  RoPE(q, k) at position *pos;  cache[*pos] = k, v
//     att = softmax(q . K^T / sqrt(hd)) V over positions [0, *pos];  records of att @ Wo per (key range, head)
// Two launches per layer (this one + pdn_decode_mlp_f32) instead of three: the q | k | v projection is done where its
// result is used.  A token is launch latency, not bytes, and what bounds a kernel is what ONE CU can pull through its
// address path (16 clocks per load instruction and wave), so the work of a head is cut into roles that balance loads:
//   * NS "range" workgroups: q only (hd columns of Wq), then the CACHED keys [s * chunk, min(*pos, (s + 1) * chunk)) of
//     the head, chunk = ceil(*pos / NS): K rows (thread = key), V rows (thread = (row group, column quad));
//   * one "new token" workgroup: q, k AND v (3 hd columns), RoPE of k, the cache append, and the new key's own
//     contribution as one more softmax partial (m = q . k / sqrt(hd), l = 1, sum = v);
//   * each of them C times, every copy owning D / C columns of the head's rows of Wo (model.py:116).
// Records (B, NS + 1, H, 4 + D) = [m, l, -, - | unnormalised contribution to the projected row]: merged, summed over
// heads and added to x by the staging of pdn_decode_mlp_f32 (n_splits = NS + 1).  Every load is issued up front in the
// order the results are needed (a wave's loads return in issue order); the staging of x overlaps the weights' flight.
#include "common.h"
#include "decode_stage.h"

#define DEC_MAX_B 8

// F4 = head_dim / 4 (12 or 16), VPRE = V rows per thread held in registers (a 256-key range), QP = k-steps of the
// q | k | v products held in registers (14 at hd 48: D <= 294, i.e. 288; later steps load in place; QP >= F4, VPRE: the
// same registers hold the K / V rows of a range role), PW = rows of Wo per thread.  EXACT: the shape needs exactly QP
// steps and PW rows -- every weight load is unconditional (a zero-initialised register overwritten by a conditional
// load costs a copy that WAITS for the load: the compiler's phi).
template <int F4, int VPRE, int QP, int PW, bool EXACT>
__global__ __launch_bounds__(256) void decode_block_kernel(
    const int* __restrict__ pos_ptr, const float* __restrict__ base, const float* __restrict__ parts, int D, int R,
    const float* __restrict__ norm_w, const float* __restrict__ Wqkv, int64_t w_bs,
    // ^ 14 dwords: in SGPRs at dispatch (kernarg preload)
    int w_rs, int H, int NS, int C, float eps, float* x_out, const float* __restrict__ cs, const float* __restrict__ sn,
    float* __restrict__ kc, float* __restrict__ vc, int64_t cbs, const float* __restrict__ Wo, int wo_rs,
    float* __restrict__ rec_out, int base_rs, int parts_rs, int x_out_rs, float inv_sqrt, int sc_floats) {
  constexpr int HD = 4 * F4, HALF = 2 * F4, SL = 256 / F4;         // k-slices of the q | k | v products (21 / 16)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ __attribute__((aligned(16))) float qs[64], ks[64], vs[64];
  __shared__ float red[16];
  __shared__ float ssq[4];
  const int tid = threadIdx.x;
  DEC_T_BEGIN(5);
  // (the position: requested FIRST and as a vector load -- a scalar load would be sunk to its first use by the
  //  compiler and cost a whole round trip in the middle of the issue phase; everything about the keys waits for it)
  const int pos_v = __hip_atomic_load(pos_ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int ci = blockIdx.x % C, role = (blockIdx.x / C) % (NS + 1), bh = blockIdx.x / (C * (NS + 1)), b = bh / H, h = bh % H;
  const bool isnew = role == NS;           // (uniform)
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const int nq = D >> 2, G = dec_div(256, nq);
  float* xs = lds;                         // [D] the staged row
  float* scratch = xs + D;                 // [G * D] staging scratch, later [3][SL][HD] partial products
  float* sc = scratch + max(G * D, 3 * SL * HD);   // [chunk] scores, then partial sums

  // ---- loads, most urgent first ----
  const DecSum sum{base + (unsigned)(b * base_rs), parts + (unsigned)(b * parts_rs), x_out + (unsigned)(b * x_out_rs),
                   0, 0, 0, R, 0, 0, 0};
  DecStage stg;
  dec_stage_issue(sum, D, 0, norm_w, stg);
  const int slice = tid / F4, quad = tid - slice * F4;
  const bool live = slice < SL;
  const int nsteps = (D + SL - 1) / SL;
  const float* wq = Wqkv + (unsigned)(h * HD + 4 * quad);
  // (row base per thread + a uniform step offset: two instructions per load; only the last step can run past D)
  const float* wq_t = wq + (unsigned)((live ? slice : 0) * w_rs);
  float4 wa[QP], wb[QP], wc[QP];
#pragma unroll
  for (int i = 0; i < QP; ++i) {
    const int k = slice + i * SL;
    if (!EXACT) wa[i] = z4;
    if (EXACT || i < nsteps) wa[i] = *reinterpret_cast<const float4*>(wq_t + ((i < nsteps - 1 || k < D) ? (unsigned)(i * SL * w_rs) : 0u));
  }
  const int pos = __builtin_amdgcn_readfirstlane(pos_v);           // (first use: by now the load is back)
  const int chunk = (pos + NS - 1) / NS, t0 = isnew ? 0 : role * chunk, t1 = isnew ? 0 : min(pos, t0 + chunk);
  float* kb = kc + (int64_t)b * cbs + (unsigned)(h * HD);
  float* vb = vc + (int64_t)b * cbs + (unsigned)(h * HD);
  const int kt = t0 + tid;
  constexpr int groups = 256 / F4;         // (= SL)
  const int tg = slice, vc4 = quad;
  if (isnew) {
    const float* wk = wq_t + w_bs;
    const float* wv = wk + w_bs;
#pragma unroll
    for (int i = 0; i < QP; ++i) {
      const int k = slice + i * SL;
      if (!EXACT) { wb[i] = z4; wc[i] = z4; }
      if (EXACT || i < nsteps) {
        const unsigned off = (i < nsteps - 1 || k < D) ? (unsigned)(i * SL * w_rs) : 0u;
        wb[i] = *reinterpret_cast<const float4*>(wk + off);
        wc[i] = *reinterpret_cast<const float4*>(wv + off);
      }
    }
  } else {
    if (t0 + (tid & ~63) < t1) {           // (this wave has a key)
      const float* kp = kb + (unsigned)((kt < t1 ? kt : t0) * D);
#pragma unroll
      for (int c = 0; c < F4; ++c) wb[c] = *reinterpret_cast<const float4*>(kp + 4 * c);
    }
    const int nv_rows = __builtin_amdgcn_readfirstlane((t1 - t0 + groups - 1) / groups);
#pragma unroll
    for (int i = 0; i < VPRE; ++i) {
      const int t = t0 + tg + i * groups;
      if (i < nv_rows) wc[i] = *reinterpret_cast<const float4*>(vb + (unsigned)(((live && t < t1) ? t : t0) * D + 4 * vc4));
    }
  }
  const int hq = min(tid, HALF - 1);
  const float rc = cs[(unsigned)(pos * HALF + hq)], rs = sn[(unsigned)(pos * HALF + hq)];
  const int Dc = D / C, nqd = Dc >> 2, Go = dec_div(256, nqd), osl = dec_div(tid, nqd), oq = tid - osl * nqd;
  float4 wo[PW];
  {
    const float* wop = Wo + (unsigned)(h * HD * wo_rs + ci * Dc + 4 * oq);
    const int nw_rows = __builtin_amdgcn_readfirstlane((HD + Go - 1) / Go);
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      const int d = osl + i * Go;
      if (!EXACT) wo[i] = z4;
      if (EXACT || i < nw_rows) wo[i] = *reinterpret_cast<const float4*>(wop + (unsigned)((d < HD ? d : 0) * wo_rs));
    }
  }

  DEC_T(1);
  // ---- x = base + records (left in x_out once per row), n = RMSNorm(x) ----
  dec_stage_row(sum, D, 0, stg, xs, scratch, ssq, h == 0 && role == 0 && ci == 0, true);

  DEC_T(2);
  float* out = rec_out + (unsigned)((((b * (NS + 1) + role) * H + h) * (4 + D)));
  if (!isnew && t0 >= t1) {                // no cached keys in this range (uniform over the workgroup)
    if (tid == 0 && ci == 0) { out[0] = -INFINITY; out[1] = 0.f; }
    for (int i = tid; i < nqd; i += 256) reinterpret_cast<float4*>(out + 4 + ci * Dc)[i] = z4;
    return;
  }

  // ---- q (| k | v) columns of this head: thread = (k-slice, column quad), slices summed through LDS in fixed order ----
  {
    float4 aq = z4, ak = z4, av = z4;
#pragma unroll
    for (int i = 0; i < QP; ++i) {
      const int k = slice + i * SL;
      float a = xs[(live && k < D) ? k : 0];
      a = (live && k < D) ? a : 0.f;       // (weights: zeros where the step was skipped)
      aq.x = fmaf(a, wa[i].x, aq.x); aq.y = fmaf(a, wa[i].y, aq.y); aq.z = fmaf(a, wa[i].z, aq.z); aq.w = fmaf(a, wa[i].w, aq.w);
      if (isnew) {
        ak.x = fmaf(a, wb[i].x, ak.x); ak.y = fmaf(a, wb[i].y, ak.y); ak.z = fmaf(a, wb[i].z, ak.z); ak.w = fmaf(a, wb[i].w, ak.w);
        av.x = fmaf(a, wc[i].x, av.x); av.y = fmaf(a, wc[i].y, av.y); av.z = fmaf(a, wc[i].z, av.z); av.w = fmaf(a, wc[i].w, av.w);
      }
    }
    if (live)
      for (int k = slice + QP * SL; k < D; k += SL) {              // (D beyond the registers' reach: loads in place)
        const float a = xs[k];
        const float4 w = *reinterpret_cast<const float4*>(wq + (unsigned)(k * w_rs));
        aq.x = fmaf(a, w.x, aq.x); aq.y = fmaf(a, w.y, aq.y); aq.z = fmaf(a, w.z, aq.z); aq.w = fmaf(a, w.w, aq.w);
        if (isnew) {
          const float4 wk4 = *reinterpret_cast<const float4*>(wq + w_bs + (unsigned)(k * w_rs));
          const float4 wv4 = *reinterpret_cast<const float4*>(wq + 2 * w_bs + (unsigned)(k * w_rs));
          ak.x = fmaf(a, wk4.x, ak.x); ak.y = fmaf(a, wk4.y, ak.y); ak.z = fmaf(a, wk4.z, ak.z); ak.w = fmaf(a, wk4.w, ak.w);
          av.x = fmaf(a, wv4.x, av.x); av.y = fmaf(a, wv4.y, av.y); av.z = fmaf(a, wv4.z, av.z); av.w = fmaf(a, wv4.w, av.w);
        }
      }
    lds_barrier();                         // (the staging is done with `scratch`)
    if (live) {
      float4* pr = reinterpret_cast<float4*>(scratch);
      pr[slice * F4 + quad] = aq;
      if (isnew) { pr[(SL + slice) * F4 + quad] = ak; pr[(2 * SL + slice) * F4 + quad] = av; }
    }
    lds_barrier();
    const int nvec = isnew ? 3 : 1;
    if (tid < nvec * HD) {
      const int vec = tid / HD, d = tid - vec * HD;
      const float* pp = scratch + vec * SL * HD + d;
      float r = 0.f;
#pragma unroll
      for (int s = 0; s < SL; ++s) r += pp[s * HD];
      (vec == 0 ? qs : (vec == 1 ? ks : vs))[d] = r * dec_norm_scale(ssq, 0, D, eps);   // (RMSNorm's scalar)
    }
    lds_barrier();
  }
  DEC_T(3);
  // ---- RoPE at position *pos (interleaved pairs, model.py:23-44); the new token's k / v go to cache row *pos ----
  if (tid < HALF) {
    const float2 q = *reinterpret_cast<const float2*>(qs + 2 * tid);
    *reinterpret_cast<float2*>(qs + 2 * tid) = make_float2(q.x * rc - q.y * rs, q.x * rs + q.y * rc);
    if (isnew) {
      const float2 k = *reinterpret_cast<const float2*>(ks + 2 * tid);
      const float2 kr = make_float2(k.x * rc - k.y * rs, k.x * rs + k.y * rc);
      *reinterpret_cast<float2*>(ks + 2 * tid) = kr;
      if (ci == 0) {
        *reinterpret_cast<float2*>(kb + (unsigned)(pos * D + 2 * tid)) = kr;
        *reinterpret_cast<float2*>(vb + (unsigned)(pos * D + 2 * tid)) = *reinterpret_cast<const float2*>(vs + 2 * tid);
      }
    }
  }
  lds_barrier();
  DEC_T(4);
  const float4* q4 = reinterpret_cast<const float4*>(qs);
  float m, l;
  if (isnew) {
    // the new key alone: one softmax partial with m = its score, l = 1, sum = v
    float s = 0.f;
    if (tid < F4) {
      const float4 a = q4[tid], k = reinterpret_cast<const float4*>(ks)[tid];
      s = (a.x * k.x + a.y * k.y) + (a.z * k.z + a.w * k.w);
    }
    s = block_sum_lds(s, red);
    m = s * inv_sqrt; l = 1.f;
    lds_barrier();
    if (tid < HD) qs[tid] = vs[tid];       // (q is dead: the hd sums go there for the projection)
  } else {
    // scores; softmax statistics per WAVE first (its own maximum and sum, no barrier), merged by every thread after
    // ONE barrier: m = max_w m_w, l = sum_w exp(m_w - m) l_w, and a key's probability is rescaled by its wave's factor
    // where it is used (keys kt, kt + 256, ... of a thread belong to the same wave)
    float mw = -INFINITY;
    for (int t = kt; t < t1; t += 256) {
      float s = 0.f;
      if (t == kt) {
#pragma unroll
        for (int c = 0; c < F4; ++c) {
          const float4 a = q4[c], k = wb[c];
          s += (a.x * k.x + a.y * k.y) + (a.z * k.z + a.w * k.w);
        }
      } else {
        const float4* k4 = reinterpret_cast<const float4*>(kb + (unsigned)(t * D));
        for (int c = 0; c < F4; ++c) {
          const float4 a = q4[c], k = k4[c];
          s += (a.x * k.x + a.y * k.y) + (a.z * k.z + a.w * k.w);
        }
      }
      s *= inv_sqrt;
      sc[t - t0] = s;
      mw = fmaxf(mw, s);
    }
    mw = wave_max(mw);
    float lw = 0.f;
    for (int t = kt; t < t1; t += 256) {
      const float pr = expf(sc[t - t0] - mw);  // (this thread's own entries: no barrier in between)
      sc[t - t0] = pr;
      lw += pr;
    }
    lw = wave_sum(lw);
    if ((tid & 63) == 0) { red[tid >> 6] = mw; red[4 + (tid >> 6)] = lw; }
    lds_barrier();
    DEC_T(5);
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float fw[4];
    l = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      fw[w] = red[w] > -INFINITY ? expf(red[w] - m) : 0.f;         // (a wave without keys: m_w = -inf, l_w = 0)
      l += fw[w] * red[4 + w];
    }
    float4 acc = z4;
    if (live) {
#pragma unroll
      for (int i = 0; i < VPRE; ++i) {
        const int t = t0 + tg + i * groups;
        if (t < t1) {
          const int w = ((t - t0) >> 6) & 3;
          const float pr = sc[t - t0] * (w == 0 ? fw[0] : (w == 1 ? fw[1] : (w == 2 ? fw[2] : fw[3])));
          acc.x += pr * wc[i].x; acc.y += pr * wc[i].y; acc.z += pr * wc[i].z; acc.w += pr * wc[i].w;
        }
      }
      for (int t = t0 + tg + VPRE * groups; t < t1; t += groups) {
        const int w = ((t - t0) >> 6) & 3;
        const float pr = sc[t - t0] * (w == 0 ? fw[0] : (w == 1 ? fw[1] : (w == 2 ? fw[2] : fw[3])));
        const float4 v = *reinterpret_cast<const float4*>(vb + (unsigned)(t * D + 4 * vc4));
        acc.x += pr * v.x; acc.y += pr * v.y; acc.z += pr * v.z; acc.w += pr * v.w;
      }
    }
    lds_barrier();                         // scores are dead: reuse the buffer for the partial sums
    float4* part = reinterpret_cast<float4*>(sc);
    if (live) part[tg * F4 + vc4] = acc;
    lds_barrier();
    // combine in a fixed order: 8 threads per column quad add every 8th group, then one thread adds those 8
    if (tid < 8 * F4) {
      const int g0 = tid / F4;
      float4 r = z4;
      for (int g = g0; g < groups; g += 8) { const float4 t = part[g * F4 + vc4]; r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w; }
      part[groups * F4 + tid] = r;
    }
    lds_barrier();
    if (tid < F4) {
      float4 r = part[groups * F4 + tid];
      for (int g = 1; g < 8; ++g) { const float4 t = part[groups * F4 + g * F4 + tid]; r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w; }
      reinterpret_cast<float4*>(qs)[tid] = r;                    // (q is dead: the head's hd sums go there)
    }
  }
  if (tid == 0 && ci == 0) { out[0] = m; out[1] = l; }
  lds_barrier();
  DEC_T(6);
  // ---- this workgroup's columns of the head's rows of Wo: thread = (row slice, column quad) ----
  float4 oacc = z4;
  float4* part = reinterpret_cast<float4*>(sc);
  if (osl < Go) {
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      const int d = osl + i * Go;
      float a = qs[d < HD ? d : 0];
      a = d < HD ? a : 0.f;                // (wo: zeros where the step was skipped)
      oacc.x = fmaf(a, wo[i].x, oacc.x); oacc.y = fmaf(a, wo[i].y, oacc.y);
      oacc.z = fmaf(a, wo[i].z, oacc.z); oacc.w = fmaf(a, wo[i].w, oacc.w);
    }
    for (int d = osl + PW * Go; d < HD; d += Go) {
      const float4 w = *reinterpret_cast<const float4*>(Wo + (unsigned)((h * HD + d) * wo_rs + ci * Dc + 4 * oq));
      const float a = qs[d];
      oacc.x = fmaf(a, w.x, oacc.x); oacc.y = fmaf(a, w.y, oacc.y); oacc.z = fmaf(a, w.z, oacc.z); oacc.w = fmaf(a, w.w, oacc.w);
    }
    part[osl * nqd + oq] = oacc;
  }
  lds_barrier();
  if (tid < nqd) {
    float4 r = part[tid];
    for (int g = 1; g < Go; ++g) { const float4 t = part[g * nqd + tid]; r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w; }
    reinterpret_cast<float4*>(out + 4 + ci * Dc)[tid] = r;
  }
  DEC_T(7);
  DEC_T_END();
}
DEC_TRACE_DUMP(pdn_dec_trace_dump_block)

// 1 when pdn_decode_block_f32 takes the shape (head_dim 48 or 64, D = H * head_dim <= 1024, (n_ranges + 1) * H <= 256,
// n_ranges <= 7 -- the merge of pdn_decode_mlp_f32 holds eight partials per head in registers).
extern "C" int pdn_decode_block_supported(int D, int H, int head_dim, int n_ranges) {
  return (head_dim == 48 || head_dim == 64) && H > 0 && D == H * head_dim && D <= 1024 && n_ranges >= 1 && n_ranges <= 7 &&
         (n_ranges + 1) * H <= 256;
}

// LDS bytes of one decode_block workgroup: the residual row, the staging region and the scores of ONE key range
// (ceil(max_len / n_ranges) positions) -- the range count must be high enough for the cache length, whatever the position.
static int64_t dec_block_lds(int D, int head_dim, int NS, int max_len, int* scf_out) {
  const int C = D % 16 == 0 ? 4 : (D % 12 == 0 ? 3 : (D % 8 == 0 ? 2 : 1));
  const int f4 = head_dim / 4, SL = 256 / f4, groups = SL, G = 256 / (D / 4), nqd = D / C / 4, Go = 256 / nqd;
  const int chunk = (max_len + NS - 1) / NS;
  int scf = chunk;                                                  // scores | partial sums | projection partials
  if ((groups + 8) * head_dim > scf) scf = (groups + 8) * head_dim;
  if (Go * nqd * 4 > scf) scf = Go * nqd * 4;
  const int scr = G * D > 3 * SL * head_dim ? G * D : 3 * SL * head_dim;
  if (scf_out) *scf_out = scf;
  return (int64_t)sizeof(float) * ((int64_t)D + scr + scf);
}
// 0 when the shape is not taken at all; otherwise the LDS bytes a launch with `n_ranges` key ranges over a cache of
// `max_len` positions needs (it runs when that is <= 64 KiB)
extern "C" int64_t pdn_decode_block_lds_bytes(int D, int H, int head_dim, int n_ranges, int max_len) {
  if (n_ranges < 1 || max_len < 1 || !pdn_decode_block_supported(D, H, head_dim, n_ranges)) return 0;
  return dec_block_lds(D, head_dim, n_ranges, max_len, nullptr);
}

// base (B, D) rows + parts (B, n_parts, D) plain records of the previous feed-forward (n_parts = 0: none) = x, written
// to x_out; Wqkv: three (D, D) matrices (in, out) w_block_stride floats apart, rows w_row_stride apart; cos / sin
// tables (max_len, head_dim / 2); caches (B, max_len, H, head_dim) with cache_batch_stride between sequences; Wo (D, D).
// records: (B, n_ranges + 1, H, 4 + D), the last one per head being the new token's own partial.
extern "C" int pdn_decode_block_f32(const float* base, int64_t base_row_stride, const float* parts, int n_parts,
                                    int64_t parts_row_stride, float* x_out, int64_t x_out_row_stride, const float* norm_w,
                                    float eps, const float* Wqkv, int64_t w_row_stride, int64_t w_block_stride,
                                    const float* cos_table, const float* sin_table, float* k_cache, float* v_cache,
                                    int64_t cache_batch_stride, const int* pos, int max_len, const float* Wo,
                                    int64_t wo_row_stride, float* records, int B, int H, int head_dim, int n_ranges,
                                    void* stream) {
  if (B == 0) return PDN_OK;
  const int D = H * head_dim, NS = n_ranges;
  PDN_CHECK_ARG(base && x_out && norm_w && Wqkv && cos_table && sin_table && k_cache && v_cache && pos && Wo && records &&
                    max_len > 0 && n_parts >= 0 && (n_parts == 0 || parts),
                "pdn_decode_block_f32: bad arguments");
  if (!pdn_decode_block_supported(D, H, head_dim, NS)) {
    pdn_set_error("pdn_decode_block_f32: head_dim %d, D %d, %d ranges are not taken (pdn_decode_block_supported)", head_dim, D, NS);
    return PDN_EUNSUPPORTED;
  }
  PDN_CHECK_ARG(B <= DEC_MAX_B, "pdn_decode_block_f32: B = %d > %d", B, DEC_MAX_B);
  const int64_t lim = (int64_t)1 << 31;
  PDN_CHECK_ARG(base_row_stride >= 0 && parts_row_stride >= 0 && x_out_row_stride >= 0 && w_row_stride >= 0 &&
                    wo_row_stride >= 0 && base_row_stride % 4 == 0 && parts_row_stride % 4 == 0 &&
                    x_out_row_stride % 4 == 0 && w_row_stride % 4 == 0 && w_block_stride % 4 == 0 &&
                    wo_row_stride % 4 == 0 && cache_batch_stride % 4 == 0 && (int64_t)B * base_row_stride < lim &&
                    (int64_t)B * parts_row_stride + (int64_t)n_parts * D < lim && (int64_t)B * x_out_row_stride < lim &&
                    (int64_t)D * w_row_stride < lim && (int64_t)D * wo_row_stride < lim && (int64_t)max_len * D < lim &&
                    (int64_t)B * (NS + 1) * H * (4 + D) < lim,
                "pdn_decode_block_f32: strides must be multiples of 4 within the 32-bit range of the kernel");
  PDN_CHECK_ARG(((((uintptr_t)base | (uintptr_t)parts | (uintptr_t)x_out | (uintptr_t)norm_w | (uintptr_t)Wqkv |
                   (uintptr_t)k_cache | (uintptr_t)v_cache | (uintptr_t)Wo | (uintptr_t)records) & 15) == 0),
                "pdn_decode_block_f32: 16-byte aligned operands");
  // copies per (role, head), each with D / C columns of Wo (measured flat between 1 and 4 at D = 288: kept at the
  // count that keeps a workgroup's share of Wo small)
  const int C = D % 16 == 0 ? 4 : (D % 12 == 0 ? 3 : (D % 8 == 0 ? 2 : 1));
  const int f4 = head_dim / 4, SL = 256 / f4, G = 256 / (D / 4), nqd = D / C / 4, Go = 256 / nqd;
  int scf = 0;
  const size_t shm = (size_t)dec_block_lds(D, head_dim, NS, max_len, &scf);
  if (shm > 64 * 1024) {        // (a valid request this build does not take: callers fall back, pdn_decode_block_lds_bytes says so beforehand)
    pdn_set_error("pdn_decode_block_f32: max_len = %d too long for %d ranges", max_len, NS);
    return PDN_EUNSUPPORTED;
  }
  const dim3 grid(B * H * (NS + 1) * C);
  hipStream_t st = (hipStream_t)stream;
  const float inv_sqrt = 1.f / sqrtf((float)head_dim);
#define BLK_GO(F4, VP, QP, PW, EX)                                                                                      \
  hipLaunchKernelGGL((decode_block_kernel<F4, VP, QP, PW, EX>), grid, dim3(256), shm, st, pos, base, parts, D, n_parts,  \
                     norm_w, Wqkv, w_block_stride, (int)w_row_stride, H, NS, C, eps, x_out, cos_table, sin_table,       \
                     k_cache, v_cache, cache_batch_stride, Wo, (int)wo_row_stride, records, (int)base_row_stride,       \
                     (int)parts_row_stride, (int)x_out_row_stride, inv_sqrt, scf)
  const int pw_need = (head_dim + Go - 1) / Go, nsteps = (D + SL - 1) / SL;
  if (head_dim == 48) {
    if (pw_need == 4 && nsteps == 14) BLK_GO(12, 13, 14, 4, true);
    else if (pw_need <= 4) BLK_GO(12, 13, 14, 4, false);
    else BLK_GO(12, 13, 14, 16, false);
  } else {
    if (pw_need == 4 && nsteps == 16) BLK_GO(16, 16, 16, 4, true);
    else if (pw_need <= 4) BLK_GO(16, 16, 16, 4, false);
    else BLK_GO(16, 16, 16, 16, false);
  }
#undef BLK_GO
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}
