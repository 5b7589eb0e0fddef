/* pdn_hip.h -- C ABI of libpdnhip.so, the MI355X (gfx950) compute backend for PyDyNet's
 * Tensor-op hot path.
 *
 * The reference (WeltXing/PyDyNet) has exactly one device seam: `Device.xp` returns the
 * module `numpy` or `cupy` (pydynet/cuda.py:89-91) and every operator calls `self.xp.<fn>`
 * or an ndarray operator on `Tensor.data`.  This header is what a HIP backend binds instead
 * of CuPy: each entry point cites the reference call site(s) it replaces.  The host shim
 * (pydynet_amd/_lib.py, ctypes) is the only caller.
 *
 * Conventions
 *   - plain C: pointers are DEVICE pointers unless named *_host; sizes/strides are in
 *     ELEMENTS (not bytes); `stream` is a hipStream_t passed as void* (NULL = default).
 *   - every function returns 0 on success, a negative PDN_E* code for a rejected argument,
 *     or a positive hipError_t; pdn_last_error() returns a thread-local message.
 *   - kernels never allocate and never retain pointers after the launch is enqueued;
 *     scratch is passed in as `workspace`, sized by the matching *_workspace_bytes query.
 *   - launches are asynchronous on `stream`; nothing synchronises except
 *     pdn_stream_synchronize and pdn_gemm_prof_collect.
 *   - dtype codes: 0 = float32, 1 = float64, 2 = int64, 3 = bool (uint8 0/1), 4 = int32,
 *     5 = float16 (storage type of the elementwise / cast / fill entry points; math in float32).
 */
#ifndef PDN_HIP_H
#define PDN_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PDN_OK 0
#define PDN_EINVAL (-1)
#define PDN_EUNSUPPORTED (-2)
#define PDN_EWORKSPACE (-3)

enum pdn_dtype { PDN_F32 = 0, PDN_F64 = 1, PDN_I64 = 2, PDN_BOOL = 3, PDN_I32 = 4, PDN_F16 = 5 };
enum pdn_binary_op { PDN_ADD = 0, PDN_SUB, PDN_MUL, PDN_DIV, PDN_POW, PDN_MAXIMUM, PDN_MINIMUM,
                     PDN_EQ = 16, PDN_NE, PDN_LT, PDN_LE, PDN_GT, PDN_GE };
enum pdn_unary_op { PDN_COPY = 0, PDN_NEG, PDN_EXP, PDN_LOG, PDN_ABS, PDN_SIGN, PDN_SQRT,
                    PDN_SQUARE, PDN_RECIP, PDN_SIGMOID, PDN_TANH };
enum pdn_reduce_op { PDN_SUM = 0, PDN_MEAN, PDN_MAX, PDN_MIN, PDN_ARGMAX, PDN_ARGMIN };

/* ---- library ------------------------------------------------------------------------- */
const char* pdn_last_error(void);
int pdn_abi_version(void);
/* replaces cp.cuda.runtime.getDeviceCount (pydynet/cuda.py:20-24) */
int pdn_device_count(void);
int pdn_device_info(int device, char* name, int cap, int* compute_units, int64_t* total_mem);
/* replaces the implicit sync of cupy `.get()` / `.item()` (pydynet/core/tensor.py:385-393) */
int pdn_stream_synchronize(void* stream);

/* ---- device runtime: what CuPy does behind pydynet/cuda.py:16-32,89-99 besides arithmetic ----
 * cp.cuda.Device(id).use() inside `with device:` (cuda.py:93-99) */
int pdn_set_device(int device);
int pdn_get_device(int* device);
int pdn_device_synchronize(void);
/* cupy's memory pool behind every xp.zeros / xp.array / op output (core/tensor.py:80,90): a caching
 * allocator (size-class free lists per device; hipMalloc only on a miss; reuse is ordered on the
 * device's compute stream).  pdn_free returns the block to the cache, never to the driver. */
int pdn_malloc(void** ptr, int64_t bytes);
int pdn_free(void* ptr);
int pdn_empty_cache(void);
int pdn_mem_stats(int device, int64_t* in_use, int64_t* reserved, int64_t* peak_in_use,
                  int64_t* device_allocs, int64_t* requests, int64_t* cache_hits);
/* `xp.asarray(host)` / `.get()` (tensor.py:385-403): *_host pointers are pageable host memory.
 * h2d returns when the source may be reused, d2h when the data is on the host; d2d and memset are
 * asynchronous on `stream`. */
int pdn_memcpy_h2d(void* dst, const void* src_host, int64_t bytes, void* stream);
int pdn_memcpy_d2h(void* dst_host, const void* src, int64_t bytes, void* stream);
int pdn_memcpy_d2d(void* dst, const void* src, int64_t bytes, void* stream);
int pdn_memset(void* dst, int byte_value, int64_t bytes, void* stream);
/* streams (non-blocking: no implicit sync with the null stream) and events; the compute stream of
 * the current device is created on first use and is what the front end passes to every kernel */
int pdn_compute_stream(void** stream);
/* pinned host memory and a device -> host copy that returns at once (order it with events; the destination must come
 * from pdn_host_alloc): lets a small read-back ride on its own stream while the compute stream goes on */
int pdn_host_alloc(void** out, int64_t bytes);
/* coherent pinned host memory mapped into the device: a kernel's system-scope store is seen by a polling host thread
 * without any copy command or event (the token of a decode step); free with pdn_host_free(*host_ptr) */
int pdn_host_alloc_mapped(void** host_ptr, void** device_ptr, int64_t bytes);
int pdn_host_free(void* ptr);
int pdn_memcpy_d2h_async(void* dst_pinned_host, const void* src, int64_t bytes, void* stream);
int pdn_stream_create(void** stream, int high_priority);
int pdn_stream_destroy(void* stream);
int pdn_stream_wait_event(void* stream, void* event);
int pdn_event_create(void** event, int timing);
int pdn_event_record(void* event, void* stream);
int pdn_event_synchronize(void* event);
int pdn_event_elapsed_ms(void* start, void* stop, float* ms);
int pdn_event_destroy(void* event);
/* hipGraph capture / replay of a whole step (the reference pays one Python object + >= 1 kernel launch per
 * scalar-level op, SURVEY 8a-3: at small batch the step is launch-bound).  Buffers a graph refers to by
 * address come from a PRIVATE pool: while a pool is active every pdn_malloc / pdn_free is served by its
 * own free lists, and its blocks rejoin the general cache only at pdn_pool_destroy. */
int pdn_pool_create(int* pool);
int pdn_pool_activate(int pool);                 /* 0 = back to the general cache */
int pdn_pool_destroy(int pool);
int pdn_pool_stats(int pool, int64_t* in_use, int64_t* reserved, int64_t* device_allocs);
int pdn_graph_begin_capture(void* stream);
int pdn_graph_end_capture(void* stream, void** graph_exec, int* n_nodes);
int pdn_graph_launch(void* graph_exec, void* stream);
int pdn_graph_destroy(void* graph_exec);
/* ---- collectives over xGMI (RCCL, bound with dlopen at first use).  No counterpart in the
 * reference: this is the one exchange step of data-parallel training.
 * One communicator rank per process; id128 = 128 bytes from rank 0's pdn_comm_unique_id, handed to
 * the other ranks by the host (pydynet_amd/rendezvous.py).  All calls are asynchronous on `stream`. */
int pdn_comm_unique_id(char* id128);
int pdn_comm_init(void** comm, int rank, int world, const char* id128);
int pdn_comm_destroy(void* comm);
int pdn_comm_allreduce_f32(void* comm, float* buf, int64_t n, int op, void* stream); /* op 0 sum, 1 max */
int pdn_comm_broadcast(void* comm, void* buf, int64_t bytes, int root, void* stream);
int pdn_comm_allgather(void* comm, const void* send, void* recv, int64_t bytes_per_rank, void* stream);

/* ---- matmul: `x.data @ y.data`, and `grad @ B^T`, `A^T @ grad` (tensor.py:659,670-675) ----
 * C[b1,b2] = alpha * A[b1,b2](MxK) * B[b1,b2](KxN) + bias[N] + beta * C[b1,b2]
 * A(m,k)=A[m*a_rs+k*a_cs], B(k,n)=B[k*b_rs+n*b_cs], C(m,n)=C[m*ldc+n]; two batch dims with
 * independent (possibly 0) strides.  fp32 MFMA, split-K through `workspace` when given.
 * Optional epilogue fusions: `residual` (indexed like C) is added to the result -- the
 * `x + sublayer(x)` of llm/llama/model.py:146,150; `b_colsum[N]` receives the column sums of B
 * for the A^T-form product x^T @ g -- the bias gradient the engine obtains by summing the
 * broadcast axes (tensor.py:360-370) -- in the same pass that forms dW. */
int pdn_gemm_f32(int M, int N, int K, float alpha, const float* A, int64_t a_rs, int64_t a_cs,
                 const float* B, int64_t b_rs, int64_t b_cs, float beta, float* C, int64_t ldc,
                 const float* bias, int nb1, int nb2, int64_t a_bs1, int64_t a_bs2,
                 int64_t b_bs1, int64_t b_bs2, int64_t c_bs1, int64_t c_bs2, const float* residual,
                 float* b_colsum, int colsum_accumulate, void* workspace, int64_t workspace_bytes,
                 void* stream);
int64_t pdn_gemm_f32_workspace_bytes(int M, int N, int K, int nbatch);
/* `relu(linear(x))` as one product, and its backward without an elementwise pass (examples/pydynet/mnist.py:70-78:
 * Linear -> ReLU -> Linear -> ReLU -> Linear; nn/functional.py:31-32 relu = maximum(0., x); tensor.py:808-814: its
 * gradient passes where out == x, i.e. where the pre-activation is >= 0).
 *  - pdn_linear_relu_fwd_f32: h (M x N, ldh) = max(0, x W + b), and mask: ONE BIT per element, set where x W + b >= 0;
 *    bit c of word [row * (N / 32) + col / 32] is column 32 * (col / 32) + c.  x rows x_rs apart (unit stride inside),
 *    W (K x N) with strides (w_rs, w_cs).  The pre-activation is never stored.
 *  - pdn_linear_dx_masked_f32: dx (M x fin, ld) = mask o (g (M x fout) W^T + existing), W (fin x fout): the consumer of h
 *    hands this layer the gradient of the PRE-activation straight from its input-gradient product.  `colsum_partials`
 *    (may be null; ceil(M / 32) x fin floats, 16-byte aligned): row b receives the column sums of dx over rows 32 b ..
 *    32 b + 31 -- summed over b they are the bias gradient of the layer below (tensor.py:360-370), without a pass over dx.
 *  - pdn_relu_mask_bwd_f32: dz = mask o g over a contiguous (rows x cols) array (any other consumer of h).
 * N / fin / cols must be multiples of 32 (pdn_relu_mask_supported). */
int pdn_relu_mask_supported(int64_t rows, int cols);
int pdn_linear_relu_fwd_f32(const float* x, int64_t x_rs, const float* W, int64_t w_rs, int64_t w_cs, const float* bias,
                            float* h, int64_t ldh, uint32_t* mask, int M, int N, int K, void* stream);
int pdn_linear_dx_masked_f32(const float* g, int64_t g_rs, const float* W, int64_t w_rs, int64_t w_cs, float* dx, int64_t ld,
                             const float* existing, const uint32_t* mask, float* colsum_partials, int M, int fin, int fout,
                             void* stream);
int pdn_relu_mask_bwd_f32(const float* g, const uint32_t* mask, float* dz, int64_t rows, int cols, void* stream);
/* Row-resident product for the layer projections (tall A with contiguous rows, contraction of a few
 * hundred): C (M x N) = A (M x K) * B + bias[N] + residual[M x N]; B is (K x N) row-major, or with
 * `b_trans` the (N x K) row-major matrix whose transpose is meant (`grad @ W^T`, tensor.py:670).
 * A's rows stay in registers, B streams through LDS (csrc/gemm_rowres.hip).  pdn_gemm_f32 routes the
 * shapes `pdn_gemm_rowres_supported` accepts here by itself; the entry point is exported for tests. */
int pdn_gemm_rowres_supported(int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int b_trans);
int pdn_gemm_rowres_f32(const float* A, const float* B, float* C, const float* bias, const float* residual,
                        int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int b_trans, void* stream);
/* Round 5: the two projections above with the RMSNorm IN FRONT of them folded into the A load (nn/modules/norm.py:221-248
 * feeding llm/llama/model.py:56-58, 93-104): x = the rows before the norm; the wave that holds 32 rows of x in registers forms
 * their rms, normalises in place, multiplies the normalised rows and leaves xn (M x K) = x / sqrt(mean(x^2) + eps) * norm_w and
 * rms (M) for the backward -- the norm's own pass over the activation is gone.  Tile-piece kernel only: `*_norm_supported`. */
int pdn_gateup_swiglu_norm_supported(int M, int F, int K);
int pdn_gateup_swiglu_norm_fwd_f32(const float* x, const float* norm_w, float eps, float* xn, float* rms, const float* w_gate,
                                   int64_t w_stride, float* gu, float* h, int M, int F, int K, int64_t ldx, void* stream);
int pdn_qkv_rope_norm_supported(int M, int D, int K, int L, int hd);
int pdn_qkv_rope_norm_fwd_f32(const float* x, const float* norm_w, float eps, float* xn, float* rms, const float* wq,
                              int64_t w_stride, float* qkv, const float* rope, int M, int D, int K, int L, int hd,
                              int64_t ldx, void* stream);
/* Launch counters per kernel: which kernel the entry points really launched since the last reset -- bench.py's parity
 * gates and the tests assert on them (a dispatch that silently falls back to a slower kernel must not stay green).
 * Copies min(n, 24) counters to `out` (may be null), clears all of them when `reset` != 0.  Slots:
 *   0 gemm_rowres_kernel (chunk kernel, any)      1 gemm_rowtile_kernel plain        2 ... + SwiGLU forward (gate | up)
 *   3 ... + SwiGLU backward (dh)                  4 ... + RoPE (q | k | v)           5 ... + row maxima (lm_head forward)
 *   6 gemm_rowres_kernel with a fused epilogue    7 attention_p forward (persistent) 8 attention_p backward (dQ + dK/dV)
 *   9 resident attention forward                 10 resident attention backward     11 streaming attention (either direction)
 *  12 lm_head input gradient + sum of exponentials (gemm_outres_kernel CE 2)        13 lm_head weight gradient, CE gradient inside
 *  14 gemm_outres_kernel plain                   15 gemm_outres_tn_kernel plain
 *  16 pdn_linear_relu_fwd_f32                    17 pdn_linear_dx_masked_f32        18 cross entropy over <= 32 classes
 *  19 pdn_gateup_swiglu_tiled_fwd_f32            20 pdn_swiglu_bwd_tiled_f32
 *  21 conv_quad_fwd_kernel (csrc/conv_quad.hip)  22 conv_quad_dgrad_kernel          23 conv_quad_wgrad_kernel */
int pdn_kernel_counters(int64_t* out, int n, int reset);
/* Round 5: which kernel the row-resident entry points below and above launch.  The tile-piece kernel
 * (csrc/gemm_rowtile.hip: one 32-column tile of B over the whole contraction per piece, rotating accumulator sets, stores
 * and epilogue reads drained under the next tile's MFMAs) takes a shape when every CU gets an 8-wave workgroup; the chunk
 * kernel of csrc/gemm_rowres.hip the rest.  mode 0 = chunk kernel only, 1 = as described (default), 2 = tile-piece kernel
 * for every valid shape (tests), anything else = query.  Returns the previous mode.  Results are bit-identical. */
int pdn_gemm_rowtile_mode(int mode);
/* Projections with the bandwidth pass next to them folded into the store of the accumulators (round 4;
 * csrc/gemm_rowres.hip, contraction 288 only -- `*_supported` says whether a shape is taken; PDN_EUNSUPPORTED otherwise):
 *  - gate | up projection + SwiGLU (llm/llama/model.py:56-58, nn/functional.py:39-40): gu (M x 2F) = x [Wg | Wu] and
 *    h (M x F) = silu(gate) * up in ONE launch; Wg, Wu (K x F) row-major, `w_stride` floats apart;
 *  - its backward half: dgu (M x 2F) = d[gate | up] from dh = dy (M x K) W_down^T, W_down (F x K) row-major, and the
 *    saved gu -- dh itself is never written;
 *  - q | k | v projection + RoPE on the q and k blocks (model.py:23-44, 93-104): qkv (M x 3D) = x [Wq | Wk | Wv], row m is
 *    position m % L; `rope` is the (L x hd x 2) table pdn_rope_table_f32 expands from the reference's (L x hd/2)
 *    cos / sin tables: (cos, sin with the sign of the column's place in its pair). */
/* dX (M x 288) = [d_1 | ... | d_nb] (M x nb * kb) [W_1 | ... | W_nb]^T (+ residual): the input gradient of projections
 * that share their input, the weights W_i (288 x kb, row-major, `b_block_stride` floats apart) read where they live
 * (csrc/gemm_outres.hip; `grad @ W^T` of tensor.py:670, one contraction over all projections) */
int pdn_gemm_outres_blocks_supported(int M, int kb, int nb);
int pdn_gemm_outres_blocks_nt_f32(const float* A, const float* W, int64_t b_block_stride, int kb, int nb, float* C,
                                  const float* residual, int M, int64_t lda, int64_t ldc, void* stream);
/* vocabulary projection WITH the row statistics of cross entropy (llm/llama/model.py:179 + nn/functional.py:364-381):
 * logits (M x V) = x (M x 288) W (288 x V) + bias and lse[m] = log sum_v exp(logits[m][v]) in one launch (transposed
 * accumulators: a lane owns a token); pdn_cross_entropy_from_lse_f32 then forms the loss with one gather per row --
 * loss_row[r] = lse[r] - logits[r][target[r]], loss_out = their sum (mean != 0: mean) -- instead of the pass over the
 * logits pdn_cross_entropy_fwd_f32 makes.  err_flag is set to 1 on an out-of-range target. */
int pdn_linear_lse_supported(int64_t M, int V, int K);
int pdn_linear_lse_fwd_f32(const float* x, const float* w, const float* bias, float* logits, float* lse, int M, int V,
                           int K, int64_t ldx, int64_t ldw, int64_t ldl, void* stream);
int pdn_cross_entropy_from_lse_f32(const float* logits, int64_t ldl, const float* lse, const int64_t* targets,
                                   int64_t rows, int V, int mean, float* loss_row, float* loss_out, int* err_flag,
                                   void* stream);
/* the same cross entropy with its statistics split over the two products that touch every logit anyway
 * (nn/functional.py:364-381 after llm/llama/model.py:179): the projection leaves rowmax[m] = max_v logits[m][v]
 * (pdn_linear_rowmax_fwd_f32: `pdn_linear_rowmax_parts` vectors of M, the row's maximum is the maximum over them -- with
 * few rows the chunks of the vocabulary are split over the grid), the input-gradient product forms exp(logit - max),
 * sums it per row while it multiplies, and normalises its rows at the end:
 *   dx[t] = gscale * (sum_v exp(l[t][v] - max[t]) W[:, v] / Z[t] - W[:, target[t]]),  lse[t] = max[t] + log Z[t]
 * (W (in x V) row-major, in = 288; independent of the upstream gradient, a scalar the caller applies).  Run in the
 * forward pass of a training step, the loss follows from lse by pdn_cross_entropy_from_lse_f32 and backward only has
 * the weight gradient left (pdn_linear_ce_backward_f32 with dx = NULL). */
int pdn_linear_rowmax_supported(int64_t M, int V, int K);
int pdn_linear_rowmax_parts(int64_t M, int V, int K);      /* vectors of M maxima `rowmax` must hold (few rows: the chunk ranges) */
int pdn_linear_rowmax_fwd_f32(const float* x, const float* w, const float* bias, float* logits, float* rowmax, int M,
                              int V, int K, int64_t ldx, int64_t ldw, int64_t ldl, void* stream);
int pdn_linear_ce_dx_deferred_supported(int64_t rows, int V, int in_features);
int64_t pdn_linear_ce_dx_deferred_workspace_bytes(int64_t rows, int V, int in_features);   /* a W^T copy (V x 288: the rows W[:, target]
                                                                                             * are read from it) + the range slabs of few-row launches */
int pdn_linear_ce_dx_deferred_f32(const float* logits, const float* rowmax, int max_parts, const int64_t* targets,
                                  float gscale, const float* W, float* dx, float* lse, int64_t rows, int V,
                                  int in_features, void* workspace, int64_t workspace_bytes, void* stream);
int pdn_gateup_swiglu_supported(int M, int F, int K);
int pdn_gateup_swiglu_fwd_f32(const float* x, const float* w_gate, int64_t w_stride, float* gu, float* h, int M,
                              int F, int K, int64_t ldx, void* stream);
int pdn_swiglu_bwd_gemm_f32(const float* dy, const float* w_down, const float* gu, float* dgu, int M, int F, int K,
                            int64_t ldy, void* stream);
/* The same two epilogues for the model widths the row-resident kernels do not take (contraction other than 288), in the
 * stores of the tiled kernel (csrc/gemm.hip): [Wg | Wu] is packed once per call into `workspace`
 * (pdn_gateup_swiglu_tiled_workspace_bytes) with alternating groups of 32 gate / 32 up columns, so a wave's accumulators
 * hold a gate group beside its up group.  M a multiple of 128, F a multiple of 32 (backward: 4), 16-byte aligned operands. */
int pdn_gateup_swiglu_tiled_supported(int M, int F, int K);
int64_t pdn_gateup_swiglu_tiled_workspace_bytes(int F, int K);
int pdn_gateup_swiglu_tiled_fwd_f32(const float* x, int64_t ldx, const float* w_gate, int64_t w_stride, float* gu, float* h,
                                    int M, int F, int K, void* workspace, int64_t workspace_bytes, void* stream);
int pdn_swiglu_bwd_tiled_supported(int M, int F, int K);
int pdn_swiglu_bwd_tiled_f32(const float* dy, int64_t ldy, const float* w_down, const float* gu, float* dgu, int M, int F,
                             int K, void* stream);
int pdn_qkv_rope_supported(int M, int D, int K, int L, int hd);
int pdn_qkv_rope_fwd_f32(const float* x, const float* wq, int64_t w_stride, float* qkv, const float* rope, int M,
                         int D, int K, int L, int hd, int64_t ldx, void* stream);
int pdn_rope_table_f32(const float* cos_t, const float* sin_t, float* out, int L, int hd, void* stream);
/* float64 matmul (the reference's default dtype: a script that never says float32 still gets the right numbers
 * on the HIP device).  v_mfma_f64_16x16x4_f64, same stride / batch conventions, no epilogue fusions. */
int pdn_gemm_f64(int M, int N, int K, double alpha, const double* A, int64_t a_rs, int64_t a_cs,
                 const double* B, int64_t b_rs, int64_t b_cs, double beta, double* C, int64_t ldc,
                 int nb1, int nb2, int64_t a_bs1, int64_t a_bs2, int64_t b_bs1, int64_t b_bs2,
                 int64_t c_bs1, int64_t c_bs2, void* stream);
/* Output-resident product for the GEMMs that end in the model width: C (M x 288) = A (M x K) * B + bias +
 * residual, K a multiple of 32, A rows contiguous; B (K x 288) row-major or, with `b_trans`, the (288 x K)
 * row-major matrix whose transpose is meant.  A wave keeps 32 x 288 outputs in accumulators, A is read once
 * straight into MFMA operand registers, B streams through LDS (csrc/gemm_outres.hip).  pdn_gemm_f32 routes the
 * shapes it pays for here by itself; the entry point is exported for tests. */
int pdn_gemm_outres_supported(int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int b_trans);
int pdn_gemm_outres_f32(const float* A, const float* B, float* C, const float* bias, const float* residual,
                        int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int b_trans, void* stream);
/* K split over the grid for short M (fewer row workgroups than fill the chip): pdn_gemm_outres_plan -> number of
 * K ranges (1 = none; *nw waves per workgroup, *kps 32-row pieces per range); with a workspace of
 * pdn_gemm_outres_workspace_bytes(M, K) bytes pdn_gemm_outres_ws_f32 writes one (M x 288) slab per range and adds
 * them (+ bias + residual) in a fixed order; without it the product runs unsplit. */
int pdn_gemm_outres_plan(int M, int K, int* nw, int* kps);
int64_t pdn_gemm_outres_workspace_bytes(int M, int K);
int pdn_gemm_outres_ws_f32(const float* A, const float* B, float* C, const float* bias, const float* residual,
                           int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int b_trans, void* workspace,
                           int64_t workspace_bytes, void* stream);
/* per-launch HIP-event timing of the GEMM kernel for bench.py's roofline block */
int pdn_gemm_prof_enable(int on);
int pdn_gemm_prof_collect(double* total_ms, double* total_flops, int64_t* launches);
/* the same split by kernel family: [0] gemm_f32_mfma_kernel (tiled, incl. its split-K reduce),
 * [1] gemm_tn_stream_*_kernel (weight gradients), [2] gemm_rowres_kernel (projections, contraction 288),
 * [3] gemm_outres_kernel (products ending in the model width, incl. the fused lm_head input gradient),
 * [4] gemm_outres_tn_kernel (weight gradients out of the model width, incl. the fused lm_head one);
 * each argument points to FIVE values */
int pdn_gemm_prof_collect_families(double* ms5, double* flops5, int64_t* launches5);
/* the row-resident launches with a fused epilogue (SwiGLU forward / backward, RoPE), timed apart: total milliseconds,
 * FLOPs, algorithmic HBM bytes (operands read + results written once) and launch count; removes their records -- if it
 * is not called first, pdn_gemm_prof_collect_families counts them with the row-resident family */
int pdn_gemm_prof_collect_fused(double* ms, double* flops, double* bytes, int64_t* launches);

/* ---- broadcasting elementwise: + - * / ** maximum minimum, comparisons
 * (tensor.py:548,564,591,612,634,811,820; 289-316).  mode 0: a op b, 1: a op scalar,
 * 2: scalar op a.  Broadcast = stride 0.  Comparison ops write uint8. */
int pdn_ew_binary(int dtype, int op, int mode, int ndim, const int64_t* shape, const void* a,
                  const int64_t* sa, const void* b, const int64_t* sb, double scalar, void* out,
                  const int64_t* so, void* stream);
/* xp.exp/log/abs/sign, unary minus, x**0.5, sigmoid/tanh piecewise forms
 * (tensor.py:689,786,802,829,1000-1002,1013-1015) */
int pdn_ew_unary(int dtype, int op, int ndim, const int64_t* shape, const void* a,
                 const int64_t* sa, void* out, const int64_t* so, void* stream);
/* ndarray.astype / .copy() / `view[...] = array` (tensor.py:168-177, 279) */
int pdn_cast(int src_dtype, int dst_dtype, int ndim, const int64_t* shape, const void* a,
             const int64_t* sa, void* out, const int64_t* so, void* stream);
/* xp.zeros / xp.ones / `grad[...] = 0.` (tensor.py:90,355,380-383) */
int pdn_fill(int dtype, double value, int ndim, const int64_t* shape, void* out,
             const int64_t* so, void* stream);
/* `data[bool_mask] = value` (Tensor.__setitem__, tensor.py:279) */
int pdn_masked_fill(int dtype, double value, int ndim, const int64_t* shape, const void* mask,
                    const int64_t* smask, void* out, const int64_t* so, void* stream);

/* ---- reductions: getattr(xp,'sum'|'mean'|'max'|'min'|'argmax'|'argmin')(x, axis, keepdims)
 * (tensor.py:701,705) and the engine's un-broadcast sums (tensor.py:360-370).
 * reduce_axis[k] != 0 marks a reduced dim; out is contiguous over the kept dims. */
int pdn_reduce(int dtype, int op, int ndim, const int64_t* shape, const int64_t* strides,
               const uint8_t* reduce_axis, const void* x, void* out, void* workspace,
               int64_t workspace_bytes, void* stream);

/* ---- fused softmax (nn/functional.py:43-49) with the attention prologue
 * `scores / sqrt(hd) + causal_mask` (llm/llama/model.py:113-117, 199-203) folded in.
 * causal_L = 0 disables the mask; divisor = 1 for plain softmax. */
int pdn_softmax_fwd_f32(const float* x, float* y, int64_t rows, int cols, float divisor,
                        int causal_L, int start_pos, void* stream);
int pdn_softmax_bwd_f32(const float* y, const float* dy, float* dx, int64_t rows, int cols,
                        float divisor, void* stream);

/* ---- RMSNorm (nn/modules/norm.py:245-248): y = x / sqrt(mean(x^2)+eps) * w; `rms` (rows,)
 * is saved for backward.  bwd: dw (+)= sum_rows dy*x/rms when dw != NULL. */
int pdn_rmsnorm_fwd_f32(const float* x, const float* w, float* y, float* rms, int64_t rows,
                        int cols, float eps, void* stream);
int pdn_rmsnorm_bwd_f32(const float* x, const float* w, const float* rms, const float* dy,
                        const float* dx_residual, float* dx, float* dw, int accumulate_dw,
                        int64_t rows, int cols,
                        void* workspace, int64_t workspace_bytes, void* stream);
int64_t pdn_rmsnorm_bwd_workspace_bytes(int64_t rows, int cols);

/* ---- last-axis LayerNorm (llm/clip/model.py:66-80, CLIPLayerNorm: mean / var over the last axis,
 * `(x - mean) / sqrt(var + eps) * scale + shift`) and CLIP's sigmoid-gated GELU
 * `x * sigmoid(1.702 x)` (llm/clip/model.py:92-95).  (The reference's own nn.LayerNorm reduces over the
 * LEADING axes: that is pdn_colnorm_*.)  bwd: dw (+)= sum_rows dy * xhat, db (+)= sum_rows dy. */
int pdn_layernorm_fwd_f32(const float* x, const float* w, const float* b, float* y, float* mean, float* rstd,
                          int64_t rows, int cols, float eps, void* stream);
int pdn_layernorm_bwd_f32(const float* x, const float* w, const float* mean, const float* rstd, const float* dy,
                          const float* dx_residual, float* dx, float* dw, float* db, int accumulate, int64_t rows,
                          int cols, void* workspace, int64_t workspace_bytes, void* stream);
int64_t pdn_layernorm_bwd_workspace_bytes(int64_t rows, int cols);
int pdn_gated_sigmoid_fwd_f32(const float* x, float* y, float alpha, int64_t n, void* stream);
int pdn_gated_sigmoid_bwd_f32(const float* x, const float* dy, float* dx, float alpha, int64_t n, void* stream);

/* ---- SiLU / SwiGLU (nn/functional.py:39-40; llm/llama/model.py:56-58):
 * y = g/(1+exp(-g)) [* u];  u == NULL selects plain SiLU. */
int pdn_swiglu_fwd_f32(const float* g, const float* u, float* y, int64_t n, void* stream);
int pdn_swiglu_bwd_f32(const float* g, const float* u, const float* dy, float* dg, float* du,
                       int64_t n, void* stream);
/* the same on a PACKED projection: gu (rows, 2F) = [gate | up] written by one batched GEMM, y (rows, F),
 * dgu (rows, 2F) = [dgate | dup] */
int pdn_swiglu_rows_fwd_f32(const float* gu, float* y, int64_t rows, int F, void* stream);
int pdn_swiglu_rows_bwd_f32(const float* gu, const float* dy, float* dgu, int64_t rows, int F, void* stream);
/* grad of relu = maximum(0., x): (out == x) * dy (tensor.py:814-815, functional.py:31-32) */
int pdn_relu_bwd_f32(const float* x, const float* dy, float* dx, int64_t n, void* stream);

/* ---- RoPE on interleaved pairs (llm/llama/model.py:23-44); x,y: (rows, heads, head_dim),
 * cos/sin tables (L, head_dim/2) already offset by start_pos; position = row % L.
 * backward != 0 rotates by -theta (the gradient).  In-place (y == x) is allowed. */
int pdn_rope_f32(const float* x, const float* cos_t, const float* sin_t, float* y, int64_t rows,
                 int L, int heads, int head_dim, int backward, void* stream);
/* the same on rows `x_row_stride` / `y_row_stride` floats apart (in place allowed): the q | k column blocks of a packed
 * q | k | v projection, rotated before the persistent attention kernels read them */
int pdn_rope_rows_f32(const float* x, const float* cos_t, const float* sin_t, float* y, int64_t rows, int L, int heads,
                      int head_dim, int64_t x_row_stride, int64_t y_row_stride, int backward, void* stream);

/* ---- fused causal self-attention of the training path (llm/llama/model.py:112-121):
 * softmax(q k^T / sqrt(hd) + causal_mask) v per (batch, head), scores kept in registers.
 * q, k, v, o (and their gradients) are (B, L, H, head_dim) as the projections produce them;
 * lse (B, H, L) = row log-sum-exp saved for the backward, which recomputes the probabilities.
 * Supported: head_dim 48, L a multiple of 32 up to 256 (else PDN_EUNSUPPORTED: the caller uses
 * the GEMM + softmax path).
 * rope_cos / rope_sin (nullable pair, (L, head_dim/2)): RoPE (model.py:23-44) fused into the
 * kernels -- q and k are rotated by their position as they are loaded, and the backward rotates
 * dq and dk back as it stores them, so the caller keeps (and differentiates) UN-rotated q, k.
 * q, k, v, dq, dk, dv share (row_stride, batch_stride) -- e.g. the three column blocks of ONE packed
 * (B*L, 3*H*hd) projection buffer; o and d_o have (o_row_stride, o_batch_stride). */
/* 1 when rotation-free operands of this shape run on the persistent, DMA-staged kernels (csrc/attention_p.hip: head dim 48,
 * L <= 256; csrc/attention_blocks.hip: 512 / 768 / 1024 positions as 256-row block pairs on them) */
int pdn_attention_persistent_supported(int L, int head_dim);
int pdn_attention_fwd_f32(const float* q, const float* k, const float* v, float* o, float* lse, int B,
                          int H, int L, int head_dim, int64_t row_stride, int64_t batch_stride,
                          int64_t o_row_stride, int64_t o_batch_stride,
                          int causal, const float* rope_cos, const float* rope_sin, void* stream);
int pdn_attention_bwd_f32(const float* q, const float* k, const float* v, const float* o,
                          const float* d_o, const float* lse, float* dq, float* dk, float* dv, int B,
                          int H, int L, int head_dim, int64_t row_stride, int64_t batch_stride,
                          int64_t o_row_stride, int64_t o_batch_stride,
                          int causal, const float* rope_cos, const float* rope_sin, void* workspace,
                          int64_t workspace_bytes, void* stream);
/* the same with q and k given ALREADY ROTATED (pdn_qkv_rope_fwd_f32 applied RoPE in the projection's epilogue):
 * nothing is rotated on the way in, dq / dk are rotated back on the way out (gradients of the un-rotated projections) */
int pdn_attention_bwd_rotated_f32(const float* q, const float* k, const float* v, const float* o,
                          const float* d_o, const float* lse, float* dq, float* dk, float* dv, int B,
                          int H, int L, int head_dim, int64_t row_stride, int64_t batch_stride,
                          int64_t o_row_stride, int64_t o_batch_stride,
                          int causal, const float* rope_cos, const float* rope_sin, void* workspace,
                          int64_t workspace_bytes, void* stream);
/* the fused kernels with an additive KEY bias (B x L; `kb_batch_stride` floats between batches, 0 = one vector for all
 * batches): the (B, 1, 1, L) padding mask of examples/pydynet/transformer.py:92-96 (-inf = masked key), and the way keys
 * beyond a length that is not a multiple of 32 are switched off after zero-padding q / k / v.  One extra rank-1 MFMA step
 * per score tile ([k | b sqrt(hd)] . [q | 1]); no RoPE inside; non-causal or causal. */
int pdn_attention_fwd_bias_f32(const float* q, const float* k, const float* v, float* o, float* lse, int B, int H, int L,
                               int head_dim, int64_t row_stride, int64_t batch_stride, int64_t o_row_stride,
                               int64_t o_batch_stride, int causal, const float* key_bias, int64_t kb_batch_stride,
                               void* stream);
int pdn_attention_bwd_bias_f32(const float* q, const float* k, const float* v, const float* o, const float* d_o,
                               const float* lse, float* dq, float* dk, float* dv, int B, int H, int L, int head_dim,
                               int64_t row_stride, int64_t batch_stride, int64_t o_row_stride, int64_t o_batch_stride,
                               int causal, const float* key_bias, int64_t kb_batch_stride, void* workspace,
                               int64_t workspace_bytes, void* stream);
int64_t pdn_attention_bwd_workspace_bytes(int B, int H, int L);   /* delta = rowsum(dO * O) */
/* Decode step (model.py:105-121 in eval mode, L = 1): q, o (B, H, head_dim); the new token attends to
 * positions [0, T) of the KV cache (max_batch, max_len, H, head_dim); no mask. */
int pdn_attention_decode_f32(const float* q, const float* k_cache, const float* v_cache, float* o, int B,
                             int H, int T, int head_dim, int64_t cache_batch_stride, void* stream);
/* Greedy decode step as a replayable graph (csrc/decode.hip): the loop of llm/llama/model.py:254-269 /
 * infer.py:46-63.  Nothing in these kernels depends on a host value that changes from token to token: the position
 * lives in device memory (`pos`, int32) and is advanced by pdn_decode_argmax_tick_f32, the last kernel of a step.
 *   pdn_decode_gemv_f32      y (B, N) = f(x) (B, K) @ W + bias + residual, B <= 8.  f = RMSNorm(norm_w, eps)
 *                            (norm.py:245-248) when norm_w != NULL; act = 1: x rows are packed [gate | up] of width
 *                            2 K and f = silu(gate) * up (functional.py:39-40).  W = N / blk_cols equally spaced
 *                            (K, blk_cols) row-major blocks (fused q | k | v, gate | up; one matrix: blk_cols = N).
 *                            y must not alias x; residual may alias y.
 *                            blk_max / blk_arg (optional, B x pdn_decode_gemv_blocks(N)): per row and workgroup the
 *                            first maximum of the workgroup's columns + its column: first half of a greedy pick.
 *                            act = 2: x rows are the partials of pdn_decode_attention_f32 (act_ns key ranges,
 *                            heads of act_hd) and f is their softmax-weighted merge.
 *   pdn_decode_attention_f32 qkv (B, 3 D) rows [q | k | v] of the new token: q, k rotated by the angle of position
 *                            *pos (model.py:23-44), k / v appended to cache row *pos (model.py:105-110), then the
 *                            query attends to cache positions [0, *pos] (model.py:112-121, L = 1: no mask), the keys
 *                            cut into n_splits ranges: partials (B, n_splits, H, 4 + hd)
 *                            = [max, sum of exp, -, - | sum of exp(s - max) v] per range, merged by the output
 *                            projection's loads (act = 2).
 *   pdn_decode_pick_tick_f32 next_ids[b] = column of the first maximum over the candidates (model.py:262-268:
 *                            argmax(-1)), also stored (system scope) at (*history)[*pos * B + b] when a history (a
 *                            device-resident pointer to a (max_len, B) int64 buffer, possibly mapped host memory:
 *                            pdn_host_alloc_mapped) is given -- a per-position slot the host reads while later steps run; with an embedding table (V, D) the picked token's row is copied to
 *                            x_next (B, D) at once, so the next step starts at its first projection; then *pos += 1.  pdn_decode_argmax_tick_f32: the same pick from
 *                            full logit rows. */
int pdn_decode_gemv_blocks(int N);
int pdn_decode_gemv_f32(const float* x, int64_t x_row_stride, const float* norm_w, float eps, const float* W,
                        int64_t w_row_stride, int blk_cols, int64_t w_block_stride, const float* bias,
                        const float* residual, int64_t res_row_stride, float* y, int64_t y_row_stride,
                        int B, int K, int N, int act, int act_ns, int act_hd, float* blk_max, int* blk_arg,
                        void* stream);
int pdn_decode_attention_f32(const float* qkv, int64_t qkv_row_stride, const float* cos_table, const float* sin_table,
                             float* k_cache, float* v_cache, float* partials, int B, int H, int head_dim, int n_splits,
                             int64_t cache_batch_stride, const int* pos, int max_len, void* stream);
int pdn_decode_pick_tick_f32(const float* blk_max, const int* blk_arg, int B, int n_blocks, int64_t* next_ids, int* pos,
                             int64_t* const* history, const float* emb, int64_t emb_row_stride, int D, float* x_next,
                             void* stream);
int pdn_decode_argmax_tick_f32(const float* logits, int64_t row_stride, int B, int V, int64_t* next_ids, int* pos,
                               void* stream);
/* Fused decode layer (csrc/decode_layer.hip, decode_stage.h): three launches per TransformerBlock (model.py:118-121)
 * instead of five.  An output projection is a sum over heads / hidden units, so its producer stops one step early:
 * every workgroup leaves the contribution of ITS head / 32 hidden units to the projected row as a record, and every
 * workgroup of the next kernel adds the records to the residual row, in one fixed order, while staging its input.
 *   pdn_decode_attention_oproj_f32  pdn_decode_attention_f32 + the head's rows of Wo (D, D; model.py:116):
 *                            records (B, n_splits, H, 4 + D) = [max, sum of exp, -, - | unnormalised contribution].
 *   pdn_decode_mlp_f32       h = base + merge(records) (written to x_out), n = RMSNorm(h), then per workgroup 32
 *                            hidden units of silu(n @ Wg) * (n @ Wu) (model.py:56-58) times their rows of Wd:
 *                            parts (B, pdn_decode_mlp_slices(F), D) plain partial rows of the feed-forward output.
 *   pdn_decode_gemv_sum_f32  pdn_decode_gemv_f32 whose input rows are base + sum of n_parts partial rows (written to
 *                            x_out), then RMSNorm: the next layer's q | k | v projection, or the vocabulary projection. */
int pdn_decode_attention_oproj_f32(const float* qkv, int64_t qkv_row_stride, const float* cos_table,
                                   const float* sin_table, float* k_cache, float* v_cache, const float* Wo,
                                   int64_t wo_row_stride, float* records, int B, int H, int head_dim, int n_splits,
                                   int64_t cache_batch_stride, const int* pos, int max_len, void* stream);
int pdn_decode_mlp_slices(int F);
int pdn_decode_mlp_f32(const float* base, int64_t base_row_stride, const float* records, int64_t records_row_stride,
                       int n_splits, int H, float* x_out, int64_t x_out_row_stride, const float* norm_w, float eps,
                       const float* Wg, const float* Wu, int64_t w_row_stride, const float* Wd, int64_t wd_row_stride,
                       float* parts, int64_t parts_row_stride, int B, int D, int F, void* stream);
/* Two launches per TransformerBlock (csrc/decode_block.hip): the q | k | v projection is done where it is used.
 *   pdn_decode_block_f32     x = base + sum of n_parts plain records (written to x_out), n = RMSNorm(x),
 *                            q | k | v = n @ Wqkv (three (D, D) blocks), RoPE at *pos, cache[*pos] = k, v, attention
 *                            over [0, *pos]: per head n_ranges workgroups for the cached keys (ceil(*pos / n_ranges)
 *                            each) + one for the new key, each times its columns of the head's rows of Wo:
 *                            records (B, n_ranges + 1, H, 4 + D) for pdn_decode_mlp_f32 (n_splits = n_ranges + 1).
 *                            head_dim 48 / 64 (pdn_decode_block_supported), else PDN_EUNSUPPORTED. */
int pdn_decode_block_supported(int D, int H, int head_dim, int n_ranges);
/* LDS bytes a launch with `n_ranges` key ranges over a cache of `max_len` positions needs (it runs when that is
 * <= 64 KiB: a workgroup holds the scores of ceil(max_len / n_ranges) positions); 0 = shape not taken at all */
int64_t pdn_decode_block_lds_bytes(int D, int H, int head_dim, int n_ranges, int max_len);
int pdn_decode_block_f32(const float* base, int64_t base_row_stride, const float* parts, int n_parts,
                         int64_t parts_row_stride, float* x_out, int64_t x_out_row_stride, const float* norm_w, float eps,
                         const float* Wqkv, int64_t w_row_stride, int64_t w_block_stride, const float* cos_table,
                         const float* sin_table, float* k_cache, float* v_cache, int64_t cache_batch_stride, const int* pos,
                         int max_len, const float* Wo, int64_t wo_row_stride, float* records, int B, int H, int head_dim,
                         int n_ranges, void* stream);
int pdn_decode_gemv_sum_f32(const float* base, int64_t base_row_stride, const float* parts, int n_parts,
                            int64_t parts_row_stride, float* x_out, int64_t x_out_row_stride, const float* norm_w,
                            float eps, const float* W, int64_t w_row_stride, int blk_cols, int64_t w_block_stride,
                            const float* bias, float* y, int64_t y_row_stride, int B, int K, int N, float* blk_max,
                            int* blk_arg, void* stream);
/* shapes the resident (K / V of a head chunk-wise in LDS) kernels above take: head_dim 48 or 64, L a multiple of 32 up
 * to 1024 -- sequences beyond 256 pass through LDS in 256-row chunks (forward: one online rescale per chunk) -- and
 * (round 6, csrc/attention_hd128.hip) head_dim 128 at ANY length 1 .. 1024, no RoPE inside: the shape of
 * examples/pydynet/transformer.py:53-130 (dim 512, 4 heads, (B, 1, 1, L) padding mask as the key bias) */
int pdn_attention_supported(int L, int head_dim);
int64_t pdn_attention_lds_bytes(int L, int head_dim);
int64_t pdn_attention_bwd_lds_bytes(int L, int head_dim);

/* General streaming attention (csrc/attention_stream.hip): any Lq / Lk, head_dim in {16, 24, 32, 48,
 * 64, 96, 128}, causal with a start position (KV-cache prefill, llm/llama/model.py:105-117) and / or an
 * additive mask (the padding mask of examples/pydynet/transformer.py:120-128, the causal mask tensor of
 * llm/clip/model.py:8-13,54-55), q and k/v with their own strides (views into a packed QKV projection
 * or into a cache).  Key tiles stream through LDS with an online softmax; nothing of size Lq x Lk
 * touches HBM.  mask element (b, h, q, k) = mask[b*sb + h*sh + q*sq + k*sk] (stride 0 broadcasts). */
int pdn_attention_stream_supported(int head_dim);
int pdn_attention_stream_fwd_f32(const float* q, const float* k, const float* v, float* o, float* lse, int B,
                                 int H, int Lq, int Lk, int head_dim, int64_t q_row_stride,
                                 int64_t q_batch_stride, int64_t kv_row_stride, int64_t kv_batch_stride,
                                 int causal, int start_pos, const float* mask, int64_t mask_sb,
                                 int64_t mask_sh, int64_t mask_sq, int64_t mask_sk, const float* rope_cos,
                                 const float* rope_sin, void* stream);
int pdn_attention_stream_bwd_f32(const float* q, const float* k, const float* v, const float* o, const float* d_o,
                                 const float* lse, float* dq, float* dk, float* dv, int B, int H, int Lq, int Lk,
                                 int head_dim, int64_t q_row_stride, int64_t q_batch_stride,
                                 int64_t kv_row_stride, int64_t kv_batch_stride, int causal, int start_pos,
                                 const float* mask, int64_t mask_sb, int64_t mask_sh, int64_t mask_sq,
                                 int64_t mask_sk, const float* rope_cos, const float* rope_sin, void* workspace,
                                 int64_t workspace_bytes, void* stream);
int64_t pdn_attention_stream_bwd_workspace_bytes(int B, int H, int Lq);

/* ---- RNN / LSTM cells (nn/modules/rnn.py:35-47, 244-262): the pointwise half after the two GEMMs as one
 * kernel per direction.  act: 0 tanh, 1 relu.  LSTM: lin (B, 4H) = [f | i | o | g] pre-activations; gates
 * (B, 4H) = [sigmoid f | sigmoid i | sigmoid o | tanh g] and tanh_c (B, H) are saved for backward;
 * hc (B, 2H) = [h' | c'];  bwd: dhc (B, 2H) -> dlin (B, 4H), dc_prev (B, H). */
int pdn_rnn_cell_fwd_f32(const float* lin, float* y, int64_t n, int act, void* stream);
int pdn_rnn_cell_bwd_f32(const float* lin, const float* y, const float* dy, float* dlin, int64_t n, int act,
                         void* stream);
int pdn_lstm_cell_fwd_f32(const float* lin, const float* c, float* gates, float* tanh_c, float* hc, int64_t B, int H,
                          void* stream);
int pdn_lstm_cell_bwd_f32(const float* dhc, const float* gates, const float* tanh_c, const float* c, float* dlin,
                          float* dc_prev, int64_t B, int H, void* stream);

/* Persistent GRU sequence (hidden size 32): the Python time loop of nn/modules/rnn.py:640-708 over
 * GRUCell.forward (:537-544) inside ONE launch per direction.  A wave64 owns 32 sequences for all T steps and
 * keeps their hidden state in MFMA accumulator registers (transposed formulation: the new h feeds the next
 * step's MFMA as its B operand unchanged); the caller hoists the input projections g1x = x Wx1 (+ b1),
 * g2x = x Wx2 (+ b2) of all steps into two GEMMs and forms the weight gradients from dG1 / dG2 the same way. */
int pdn_gru_seq_supported(int hidden);
int pdn_gru_seq_fwd_f32(const float* g1x, const float* g2x, const float* h0, const float* wh1, const float* wh2,
                        float* z, float* r, float* rh, float* n, float* out, int T, int B, int H, void* stream);
int pdn_gru_seq_bwd_f32(const float* g, const float* z, const float* r, const float* n, const float* out,
                        const float* h0, const float* wh1, const float* wh2, float* dg1, float* dg2, float* dh0, int T,
                        int B, int H, void* stream);

/* ---- embedding: `weight[ids]` (nn/functional.py:14-20) and its gradient
 * `full = zeros; full[key] = grad` (tensor.py:937-940: scatter-ASSIGN, last write wins).
 * scatter mode 0: assign, 1: assign-last accumulated into dW, 2: atomic scatter-add.
 * row_owner (V floats) / owner_tag: optional data-parallel filter -- a row is written only when
 * row_owner[id] == owner_tag, i.e. when this rank holds the last occurrence of the id in the
 * concatenated global batch (NULL = no filter). */
int pdn_embedding_gather_f32(const float* W, int64_t V, int D, int64_t w_row_stride,
                             const int64_t* ids, int64_t n, float* out, int* err_flag,
                             void* stream);
int pdn_embedding_scatter_f32(const float* g, const int64_t* ids, int64_t n, float* dW, int64_t V,
                              int D, int mode, const float* row_owner, float owner_tag,
                              void* workspace, int64_t workspace_bytes, void* stream);
int64_t pdn_embedding_scatter_workspace_bytes(int64_t V);
/* `x[range(N), idx]` and its scatter-assign gradient (nn/functional.py:371) */
int pdn_take_cols_f32(const float* x, int64_t n, int64_t C, int64_t x_row_stride,
                      const int64_t* idx, float* out, int* err_flag, void* stream);
int pdn_put_cols_f32(const float* g, const int64_t* idx, float* dx, int64_t n, int64_t C,
                     void* stream);

/* ---- cross entropy with integer targets (nn/functional.py:364-381), fused:
 * loss_row[n] = logsumexp(x[n,:]) - x[n,t_n]; loss_out = mean or sum of loss_row.
 * bwd: dlogits = (softmax(x) - onehot) * gscale * (upstream ? upstream[0] : 1); may alias x. */
int pdn_cross_entropy_fwd_f32(const float* logits, const int64_t* targets, int64_t rows, int V,
                              int mean, float* loss_row, float* lse_row, float* loss_out,
                              int* err_flag, void* stream);
int pdn_cross_entropy_bwd_f32(const float* logits, const int64_t* targets, const float* lse_row,
                              const float* upstream, float gscale, float* dlogits, int64_t rows,
                              int V, void* stream);
/* Backward of `linear -> cross entropy` (llm/llama/model.py:179 feeding nn/functional.py:364-381) without the
 * (rows x V) gradient of the logits in memory: both products form
 *   dlogits[t][v] = (exp(logits[t][v] - lse[t]) - [v == targets[t]]) * gscale * (upstream ? upstream[0] : 1)
 * from the saved logits and the row statistics of pdn_cross_entropy_fwd_f32 as they consume it:
 *   dx (rows x in) = dlogits W^T (+ dx_residual);  dW (in x V) = dw_beta dW + x^T dlogits;
 *   dbias (V) = db_beta dbias + column sums of dlogits.   W (in x V) row-major; any of dx / dW / dbias may be null.
 * in = 288 only (pdn_linear_ce_supported); other shapes use pdn_cross_entropy_bwd_f32 + pdn_gemm_f32. */
int pdn_linear_ce_supported(int64_t rows, int V, int in_features);
int64_t pdn_linear_ce_workspace_bytes(int64_t rows, int V, int in_features);
int pdn_linear_ce_backward_f32(const float* x, int64_t ldx, const float* logits, const float* lse,
                               const int64_t* targets, float gscale, const float* upstream, const float* W,
                               float* dx, const float* dx_residual, float* dW, float dw_beta, float* dbias,
                               float db_beta, int64_t rows, int V, int in_features, void* workspace,
                               int64_t workspace_bytes, void* stream);

/* ---- Conv2d building blocks (nn/functional.py:194-339).  im2col folds the zero padding of
 * __pad2d (:241-245) into the gather and writes the reference's exact layout
 * (N, C, kh, kw, oh, ow) -- bit-exact with `as_strided(...).copy()` (:211-222).  col2im is the
 * adjoint of `xp.add.at` on the overlapping view (:224-232) as a deterministic gather.
 * Pooling (mode 0 = max, 1 = avg) works on the zero-padded input like the reference; max
 * backward sends the gradient to every tied position (core/tensor.py:744-750).
 * col is (N, col_rows, oh*ow): rows [0, C*k*k) are the reference layout, the rest pad the
 * contraction (zeros; row C*k*k = ones when ones_row, pairing with a bias column of the packed
 * weight so that `+ bias` (functional.py:279) and its gradient ride inside the GEMMs). */
int pdn_im2col2d_f32(const float* x, int N, int C, int H, int W, int k, int stride, int pad,
                     float* col, int col_rows, int ones_row, void* stream);
int pdn_col2im2d_f32(const float* dcol, int N, int C, int H, int W, int k, int stride, int pad,
                     float* dx, int col_rows, void* stream);
/* Direct (implicit-GEMM) convolution for small-image / small-channel shapes (the zero-padded image
 * and the weights fit in LDS): the im2col matrix of nn/functional.py:211-222 exists only as LDS
 * addresses, MFMA operands are read straight from the staged image, the NCHW result (+ bias, the
 * `+ self.bias` of nn/modules/conv.py:99-103) leaves the accumulators once.  bwd_data is the
 * `np.add.at` col2im of :224-232 fused with its GEMM (stride 1); bwd_weight forms dW (O, C, k, k)
 * and db (O) in one pass (deterministic two-stage reduction through `workspace`).
 * pdn_conv2d_direct_supported -> bitmask 1 fwd | 2 bwd_data | 4 bwd_weight; other shapes use
 * pdn_im2col2d_f32 + pdn_gemm_f32 (+ pdn_col2im2d_f32). */
int pdn_conv2d_direct_supported(int C, int H, int W, int O, int k, int stride, int pad);
int pdn_conv2d_fwd_f32(const float* x, const float* w, const float* bias, float* y, int N, int C, int H,
                       int W, int O, int k, int stride, int pad, void* stream);
int pdn_conv2d_bwd_data_f32(const float* dy, const float* w, float* dx, int N, int C, int H, int W, int O,
                            int k, int stride, int pad, void* stream);
int pdn_conv2d_bwd_weight_f32(const float* x, const float* dy, float* dw, float* db, int accumulate, int N,
                              int C, int H, int W, int O, int k, int stride, int pad, void* workspace,
                              int64_t workspace_bytes, void* stream);
int64_t pdn_conv2d_bwd_weight_workspace_bytes(int N, int C, int H, int W, int O, int k, int stride, int pad);
/* conv -> relu -> max_pool(2, 2) as ONE node (the chain of examples/pydynet/mnist.py:92-95; functional.py:31-32,
 * 254-339): the full-resolution conv output and its gradient never exist in HBM.  Forward writes the pooled map and
 * a hit map of one BIT per conv output position, (N, O, OH * OW / 32) 32-bit words, bit (p & 31) of word p >> 5 for
 * position p = oy * OW + ox -- set when the position receives the pooled gradient: relu(y) equals the window maximum
 * (ties all pass, tensor.py:808-815) AND y >= 0 (relu'(0) = 1, the reference's maximum(0., x) quirk).  The backward forms expand the pooled gradient through the mask while it is staged into LDS.
 * pdn_conv2d_relu_pool_supported -> bitmask 1 forward | 2 data gradient | 4 weight gradient;
 * pdn_pool_mask_expand_f32 materialises the expanded gradient (rows, OH, OW) for shapes only the plain kernels take.
 * Workspace of the weight gradient: pdn_conv2d_bwd_weight_workspace_bytes. */
int pdn_conv2d_relu_pool_supported(int C, int H, int W, int O, int k, int stride, int pad);
int pdn_conv2d_relu_pool_fwd_f32(const float* x, const float* w, const float* bias, float* pooled, unsigned* mask,
                                 int N, int C, int H, int W, int O, int k, int stride, int pad, void* stream);
int pdn_conv2d_relu_pool_bwd_data_f32(const float* dpooled, const unsigned* mask, const float* w, float* dx, int N,
                                      int C, int H, int W, int O, int k, int stride, int pad, void* stream);
int pdn_conv2d_relu_pool_bwd_weight_f32(const float* x, const float* dpooled, const unsigned* mask, float* dw,
                                        float* db, int accumulate, int N, int C, int H, int W, int O, int k, int stride,
                                        int pad, void* workspace, int64_t workspace_bytes, void* stream);
int pdn_pool_mask_expand_f32(const float* dpooled, const unsigned* mask, float* dy, int64_t rows, int OH, int OW,
                             void* stream);
int pdn_pool2d_fwd_f32(const float* x, int N, int C, int H, int W, int k, int stride, int pad,
                       int mode, float* y, void* stream);
int pdn_pool2d_bwd_f32(const float* x, const float* y, const float* dy, int N, int C, int H, int W,
                       int k, int stride, int pad, int mode, float* dx, void* stream);

/* Forward and backward of the same loss in one pass over HBM: dlogits = (softmax - onehot) *
 * gscale is written while the row is still on chip.  Backward then only applies the upstream
 * scalar with pdn_scale_by_device_scalar_f32, which reads it on the device and leaves the
 * buffer untouched when it is exactly 1 (the `loss.backward()` case) -- no host sync.
 * dlogits_colsum (nullable, (V,)): column sums of dlogits, i.e. the bias gradient of the Linear
 * that produced the logits (nn/modules/linear.py:41, tensor.py:360-370 un-broadcast sum) formed
 * while the rows stream through; needs pdn_cross_entropy_colsum_workspace_bytes(rows, V) bytes of
 * workspace, which is 0 when the shape cannot take the fused path (then pass NULL). */
int pdn_cross_entropy_fwd_bwd_f32(const float* logits, const int64_t* targets, int64_t rows, int V,
                                  int mean, float gscale, float* loss_row, float* lse_row,
                                  float* loss_out, float* dlogits, float* dlogits_colsum,
                                  void* workspace, int64_t workspace_bytes, int* err_flag,
                                  void* stream);
int64_t pdn_cross_entropy_colsum_workspace_bytes(int64_t rows, int V);
int pdn_scale_by_device_scalar_f32(float* x, int64_t n, const float* scalar_dev, void* stream);

/* ---- reference LayerNorm / BatchNorm1d (nn/modules/norm.py:60-74, 203-218): statistics per COLUMN
 * of x (rows, cols) -- the reference's LayerNorm reduces over the leading axes -- biased variance,
 * y = (x - mean) * rstd * w + b; running_{mean,var} (nullable) <- (1-m) * running + m * stat.
 * mean / rstd (cols,) are saved for the backward: db = sum dy, dw = sum dy*xhat,
 * dx = w * rstd * (dy - db/R - xhat * dw/R); dx / dw / db are each optional, dw/db (+)= when
 * accumulate.  Workspace: pdn_colnorm_workspace_bytes. */
int pdn_colnorm_fwd_f32(const float* x, const float* w, const float* b, float* y, float* mean, float* rstd,
                        float* running_mean, float* running_var, float momentum, float eps,
                        int64_t rows, int cols, void* workspace, int64_t workspace_bytes, void* stream);
int pdn_colnorm_bwd_f32(const float* x, const float* w, const float* mean, const float* rstd,
                        const float* dy, float* dx, float* dw, float* db, int accumulate, int64_t rows,
                        int cols, void* workspace, int64_t workspace_bytes, void* stream);
int64_t pdn_colnorm_workspace_bytes(int64_t rows, int cols);

/* ---- GRU cell (nn/modules/rnn.py:537-544): the gate algebra between the four GEMMs.
 *   [z, r] = sigmoid(g1) with g1 = x Wx1 + h Wh1 + b1 (B, 2H);  rh = r * h
 *   n = tanh(g2) with g2 = x Wx2 + rh Wh2 + b2 (B, H);          h' = (1 - z) h + z n
 * backward: out_bwd gives dg2, the z half of dg1 and dh = dh' (1 - z); after drh = dg2 Wh2^T,
 * gates_bwd fills the r half of dg1 and adds drh * r to dh.  sigmoid / tanh are the reference's
 * piecewise forms (core/tensor.py:996-1019). */
int pdn_gru_gates_fwd_f32(const float* g1, const float* h, float* z, float* r, float* rh, int64_t B,
                          int H, void* stream);
int pdn_gru_out_fwd_f32(const float* g2, const float* z, const float* h, float* n, float* hnew,
                        int64_t B, int H, void* stream);
int pdn_gru_out_bwd_f32(const float* dhnew, const float* z, const float* n, const float* h, float* dg2,
                        float* dg1, float* dh, int64_t B, int H, void* stream);
int pdn_gru_gates_bwd_f32(const float* drh, const float* r, const float* h, float* dg1, float* dh,
                          int64_t B, int H, void* stream);

/* ---- Adam.step for all parameters in one launch (optim/optimizer.py:185-196).
 * chunk_table_dev: device int64[nchunks][5] = {p, g, m, v addresses, n elements}.
 * step = lr*sqrt(1-b2^t)/(1-b1^t) computed on the host as the reference does. */
int pdn_adam_multi_f32(const int64_t* chunk_table_dev, int nchunks, float step, float beta1,
                       float beta2, float one_minus_beta1, float one_minus_beta2, float eps,
                       float weight_decay, float grad_scale, void* stream);
/* the same step replayable from a hipGraph: {t, lr} live on the device as doubles in state_dev, a 1-thread
 * kernel writes step = lr * sqrt(1-b2^t)/(1-b1^t) to step_dev and advances t, the update reads it there */
int pdn_adam_multi_tick_f32(const int64_t* chunk_table_dev, int nchunks, double* state_dev, float* step_dev,
                            float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                            void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PDN_HIP_H */
