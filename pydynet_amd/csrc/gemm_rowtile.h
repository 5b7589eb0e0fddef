// Launch interface of the tile-piece row-resident GEMM (csrc/gemm_rowtile.hip), shared with csrc/gemm_rowres.hip,
// whose entry points (pdn_gemm_rowres_f32, pdn_gateup_swiglu_fwd_f32, pdn_swiglu_bwd_gemm_f32, pdn_qkv_rope_fwd_f32,
// pdn_linear_rowmax_fwd_f32) route the shapes it takes here.
#pragma once
#include <stdint.h>

struct RowTileArgs {
  const float* A;                 // (M x 288), rows contiguous, leading dimension lda
  const float* B;                 // NN: (288 x N) row-major; NT: (N x 288) row-major (its transpose is meant)
  float* C;                       // (M x N) (EPI 1 / 2: M x 2F, the packed [gate | up] layout)
  const float* bias;              // EPI 0 / 5, may be null
  const float* residual;          // EPI 0, may be null: (M x N, leading dimension ldc) added to the product
  int M, N;
  int64_t lda, ldb, ldc;
  int b_trans;
  int nblocks;                    // B as `nblocks` equally spaced matrices side by side (Wq | Wk | Wv), N / nblocks columns each
  int64_t b_block_stride;
  int epi;                        // 0 none, 1 SwiGLU forward, 2 SwiGLU backward, 3 RoPE, 5 row maxima
  float* H; int64_t ldh;          // EPI 1
  const float* GU;                // EPI 2
  int F;                          // EPI 1 / 2
  const float* rope; int L, hd, rope_cols;    // EPI 3
  unsigned g_off, u_off;          // EPI 1
  float* lse;                     // EPI 5 (null: plan only -> *parts)
  int* parts;                     // EPI 5: number of column ranges = vectors of maxima
  // RMSNorm in front of the projection (EPI 1 / 3): A = the rows before the norm, norm_w (288), y -> xn (M x 288, ldxn), rms (M)
  const float* norm_w; float* xn; float* rms; int64_t ldxn; float norm_eps;
};

// 1 when the tile-piece kernel takes this shape (K = 288, N a multiple of 32, enough rows to give every CU one 8-wave
// workgroup -- with the columns split over grid.y if need be)
int pdn_rowtile_takes(const RowTileArgs& a);
int pdn_rowtile_launch(const RowTileArgs& a, void* stream);
