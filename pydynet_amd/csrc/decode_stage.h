// Hand-off of a residual-stream row between the kernels of the fused decode layer (decode.hip, decode_layer.hip).
//
// The reference adds a block's output to its input right away (llm/llama/model.py:118-121: x + attention, then
// h + feed_forward).  At batch 1 that addition would cost a launch of its own, because the output projection of a
// block is a sum over ALL heads / hidden units: here the producing kernel stops one step earlier -- every workgroup
// writes the contribution of ITS head (or hidden-unit slice) to the projected row as a RECORD -- and every workgroup
// of the consuming kernel adds the records to the base row in one fixed order while it stages its input:
//      row = base + sum_r w_r * rec_r
// Plain records (hdr = 0) have w = 1.  Attention records (hdr = 4) start with [m, l, -, -] of a (key range, head)
// softmax partial; their weight is the flash-decoding merge  w = exp(m - M_h) / sum_s exp(m_s - M_h) l_s  over the
// `ns` key ranges of the head.  The order is the same in every workgroup, so they all see bit-identical rows; the
// first one also leaves the row in `x_out` -- the base of the next hand-off.
#pragma once
#include "common.h"

// -DDEC_TRACE (tools/decode_trace.sh, never the shipped build): workgroup 0 of every decode kernel keeps 100 MHz
// timestamps of its phases and leaves them in a per-translation-unit buffer the probe reads back.
#ifdef DEC_TRACE
#define DEC_TRACE_SLOTS 8192
static __device__ unsigned long long dec_trace_buf[DEC_TRACE_SLOTS][10];
static __device__ unsigned dec_trace_cnt;
#define DEC_T_BEGIN(kind) unsigned long long tr_[9] = {}; const bool tr_on = threadIdx.x == 0 && blockIdx.x == 0; \
  const int tr_kind = (kind); if (tr_on) tr_[0] = __builtin_amdgcn_s_memrealtime()
#define DEC_T(i) do { if (tr_on) tr_[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define DEC_T_END() do { if (tr_on) { const unsigned sl = atomicAdd(&dec_trace_cnt, 1u); if (sl < DEC_TRACE_SLOTS) { \
  for (int i_ = 0; i_ < 9; ++i_) dec_trace_buf[sl][i_] = tr_[i_]; dec_trace_buf[sl][9] = tr_kind; } } } while (0)
#define DEC_TRACE_DUMP(name) extern "C" int name(unsigned long long* out, unsigned* n) { \
  hipMemcpyFromSymbol(out, HIP_SYMBOL(dec_trace_buf), sizeof(unsigned long long) * DEC_TRACE_SLOTS * 10); \
  hipMemcpyFromSymbol(n, HIP_SYMBOL(dec_trace_cnt), sizeof(unsigned)); unsigned z = 0; \
  hipMemcpyToSymbol(HIP_SYMBOL(dec_trace_cnt), &z, sizeof(unsigned)); return 0; }
#else
#define DEC_T_BEGIN(kind)
#define DEC_T(i)
#define DEC_T_END()
#define DEC_TRACE_DUMP(name)
#endif

// a / d for 0 <= a <= 256, 1 <= d <= 256, exact: (a + 0.5) / d is at least 0.5 / d away from an integer, far more than
// the error of the reciprocal (4 instructions instead of the ~20 of an integer division; these kernels are short
// chains at one wave per SIMD, where every instruction in front of the first load is latency)
__device__ __forceinline__ int dec_div(int a, int d) { return (int)(((float)a + 0.5f) * __frcp_rn((float)d)); }

struct DecSum {
  const float* base; const float* recs; float* x_out;   // (B, K) rows; per row R records of hdr + K floats (R = 0: none)
  int base_rs, recs_rs, x_out_rs;                       // row strides in floats (32-bit: the hot loop is address math)
  int R, hdr, ns, H;                                    // hdr = 4: R = ns * H records ordered (range, head)
};

// floats of LDS scratch dec_stage_sum needs behind the staged rows
__host__ __device__ inline int dec_sum_scratch(int K, int R, int hdr) {
  const int nq = K / 4, G = 256 / nq > 0 ? 256 / nq : 1;
  return G * K + (hdr ? 3 * R : 0);
}

// A decode kernel is a chain of memory round trips, and a wave's loads come back in the order they were issued: the
// few KB a workgroup needs FIRST (base row, records, norm weight) are requested before its bulk weight prefetch, and
// everything that depends on them is computed while the weights are still on their way.  Hence the two halves:
//   dec_stage_issue  -- only loads (row b's base quad, the thread's first PF records, the (m, l) headers)
//   dec_stage_row    -- weights of the merge, the sum in fixed order, optional RMSNorm; leaves the row in xs
#define DEC_PF 8
struct DecStage {
  float4 bs, nw;                           // base quad / norm-weight quad of thread tid < K / 4
  float4 p[DEC_PF];
  float2 ml;
};

// K % 4 == 0, K <= 1024, 256 threads.  thread = (group g, quad q): g walks the records g, g + G, ...
// Branch-free: every thread loads from a valid (clamped) address and the masks are applied where the values are used
// -- a conditional load costs ~20 instructions of control flow, and the waitcnt pass can only count loads it is sure of.
__device__ __forceinline__ void dec_stage_issue(const DecSum& s, int K, int b, const float* __restrict__ norm_w, DecStage& r) {
  const int tid = threadIdx.x, nq = K >> 2, G = dec_div(256, nq), g = dec_div(tid, nq), q = tid - g * nq;
  const int rec = s.hdr + K;
  r.bs = *reinterpret_cast<const float4*>(s.base + (unsigned)(b * s.base_rs + 4 * q));
  if (b == 0) r.nw = norm_w ? *reinterpret_cast<const float4*>(norm_w + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
  r.ml = make_float2(0.f, 0.f);
  if (s.R > 0) {                           // (uniform)
    const float* rb = s.recs + (unsigned)(b * s.recs_rs);
    if (s.hdr) r.ml = *reinterpret_cast<const float2*>(rb + (unsigned)(min(tid, s.R - 1) * rec));
#pragma unroll
    for (int i = 0; i < DEC_PF; ++i)
      r.p[i] = *reinterpret_cast<const float4*>(rb + (unsigned)(min(g + i * G, s.R - 1) * rec + s.hdr + 4 * q));
  }
}

// `scratch`: dec_sum_scratch(K, R, hdr) floats of LDS.  With norm (r.nw loaded) the row left in xs is x * w of
// RMSNorm(x) = x / sqrt(mean(x^2) + eps) * w (norm.py:245-248) WITHOUT the scalar 1 / sqrt(...): a product of the row
// with a matrix is linear in it, so the consumer multiplies its few outputs by dec_norm_scale() instead -- the sum of
// squares then needs no barriers of its own (each wave leaves its part in ssq[4 b + wave], published by the barrier
// that publishes xs).  x_out always receives the un-normalised sum.  ssq: 4 floats per row.
__device__ __forceinline__ float dec_norm_scale(const float* ssq, int b, int K, float eps) {
  const float ss = (ssq[4 * b] + ssq[4 * b + 1]) + (ssq[4 * b + 2] + ssq[4 * b + 3]);
  return 1.f / sqrtf(ss / (float)K + eps);
}
__device__ __forceinline__ void dec_stage_row(const DecSum& s, int K, int b, DecStage& r, float* xs, float* scratch,
                                              float* ssq, bool writer, bool norm) {
  const int tid = threadIdx.x, nq = K >> 2, G = dec_div(256, nq), g = dec_div(tid, nq), q = tid - g * nq;
  const bool on = g < G;
  const int rec = s.hdr + K;
  float* wts = scratch + G * K;            // [R] weights, then [R] m, [R] l
  if (s.R > 0) {
    if (s.hdr) {
      if (tid < s.R) { wts[s.R + tid] = r.ml.x; wts[2 * s.R + tid] = r.ml.y; }
      lds_barrier();
      if (tid < s.H) {
        // the head's (m, l) pairs go to registers with ONE round of independent LDS reads: done through LDS value by
        // value, this merge is ~60 dependent LDS round trips (> 1 us on the chain of every feed-forward kernel)
        constexpr int NSM = 8;
        if (s.ns <= NSM) {
          float mm[NSM], ll[NSM];
#pragma unroll
          for (int sp = 0; sp < NSM; ++sp) {
            const int rr = min(sp, s.ns - 1) * s.H + tid;
            mm[sp] = wts[s.R + rr]; ll[sp] = wts[2 * s.R + rr];
          }
          float M = -INFINITY;
#pragma unroll
          for (int sp = 0; sp < NSM; ++sp)
            if (sp < s.ns && ll[sp] > 0.f) M = fmaxf(M, mm[sp]);
          float den = 0.f;
#pragma unroll
          for (int sp = 0; sp < NSM; ++sp) {
            mm[sp] = (sp < s.ns && ll[sp] > 0.f) ? expf(mm[sp] - M) : 0.f;
            den += mm[sp] * ll[sp];
          }
          const float inv = 1.f / den;
#pragma unroll
          for (int sp = 0; sp < NSM; ++sp)
            if (sp < s.ns) wts[sp * s.H + tid] = mm[sp] * inv;
        } else {
          float M = -INFINITY;
          for (int sp = 0; sp < s.ns; ++sp) {
            const int rr = sp * s.H + tid;
            if (wts[2 * s.R + rr] > 0.f) M = fmaxf(M, wts[s.R + rr]);
          }
          float den = 0.f;
          for (int sp = 0; sp < s.ns; ++sp) {
            const int rr = sp * s.H + tid;
            const float w = wts[2 * s.R + rr] > 0.f ? expf(wts[s.R + rr] - M) : 0.f;
            wts[rr] = w;
            den += w * wts[2 * s.R + rr];
          }
          const float inv = 1.f / den;
          for (int sp = 0; sp < s.ns; ++sp) wts[sp * s.H + tid] *= inv;
        }
      }
      lds_barrier();
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (on) {
      const float* rb = s.recs + (unsigned)(b * s.recs_rs + s.hdr + 4 * q);
#pragma unroll
      for (int i = 0; i < DEC_PF; ++i) {
        const int rr = g + i * G;
        float w = s.hdr ? wts[min(rr, s.R - 1)] : 1.f;
        w = rr < s.R ? w : 0.f;            // (w = 0 also marks a key range without keys: its record is zeros)
        acc.x = fmaf(w, r.p[i].x, acc.x); acc.y = fmaf(w, r.p[i].y, acc.y);
        acc.z = fmaf(w, r.p[i].z, acc.z); acc.w = fmaf(w, r.p[i].w, acc.w);
      }
      for (int r0 = g + DEC_PF * G; r0 < s.R; r0 += DEC_PF * G) {
#pragma unroll
        for (int i = 0; i < DEC_PF; ++i)
          r.p[i] = *reinterpret_cast<const float4*>(rb + (unsigned)(min(r0 + i * G, s.R - 1) * rec));
#pragma unroll
        for (int i = 0; i < DEC_PF; ++i) {
          const int rr = r0 + i * G;
          float w = s.hdr ? wts[min(rr, s.R - 1)] : 1.f;
          w = rr < s.R ? w : 0.f;
          acc.x = fmaf(w, r.p[i].x, acc.x); acc.y = fmaf(w, r.p[i].y, acc.y);
          acc.z = fmaf(w, r.p[i].z, acc.z); acc.w = fmaf(w, r.p[i].w, acc.w);
        }
      }
      *reinterpret_cast<float4*>(scratch + g * K + 4 * q) = acc;
    }
    lds_barrier();
  }
  float4 v = r.bs;
  float ss = 0.f;
  if (tid < nq) {
    if (s.R > 0)
      for (int gg = 0; gg < G; ++gg) {
        const float4 t = *reinterpret_cast<const float4*>(scratch + gg * K + 4 * q);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
    if (writer && s.x_out) *reinterpret_cast<float4*>(s.x_out + (unsigned)(b * s.x_out_rs + 4 * q)) = v;
    ss = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  if (norm) {
    ss = wave_sum(ss);
    if ((tid & 63) == 0) ssq[4 * b + (tid >> 6)] = ss;
    v.x *= r.nw.x; v.y *= r.nw.y; v.z *= r.nw.z; v.w *= r.nw.w;
  }
  if (tid < nq) *reinterpret_cast<float4*>(xs + b * K + 4 * q) = v;
  lds_barrier();                         // (xs is read next; scratch / weights / red may be rewritten)
}
