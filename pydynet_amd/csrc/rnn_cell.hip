// Fused pointwise halves of the RNN and LSTM cells (gfx950, fp32, HBM-bound streams).
//
//   RNNCell.forward   nn/modules/rnn.py:35-47     h' = act(x Wx + h Wh + b), act = tanh | relu
//   LSTMCell.forward  nn/modules/rnn.py:244-262   lin = x Wx + h Wh + b;  [f, i, o] = sigmoid(lin[:, :3H]),
//                                                 g = tanh(lin[:, 3H:]);  c' = f c + i g;  h' = o tanh(c')
// The two GEMMs of a cell (the second accumulating into the first's output, bias in its epilogue) stay on
// the MFMA GEMM; everything after them -- 2 tape nodes for the RNN cell, 12 for the LSTM cell (two hsplits,
// sigmoid, two tanh, three mul, add) -- is ONE kernel forward and ONE backward here.  sigmoid / tanh are the
// reference's overflow-safe piecewise forms (core/tensor.py:999-1003, 1012-1016).
#include "common.h"

__device__ __forceinline__ float rc_sigmoid(float x) {
  return x > 0.f ? 1.f / (1.f + expf(-x)) : 1.f - 1.f / (1.f + expf(x));
}
__device__ __forceinline__ float rc_tanh(float x) {
  return x > 0.f ? 2.f / (1.f + expf(-2.f * x)) - 1.f : 1.f - 2.f / (1.f + expf(2.f * x));
}

// act 0 = tanh, 1 = relu (maximum(0., x): the gradient passes where out == x, i.e. also at x == 0)
__global__ void rnn_cell_fwd_kernel(const float* __restrict__ lin, float* __restrict__ y, int64_t n, int act) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = act == 0 ? rc_tanh(lin[i]) : fmaxf(0.f, lin[i]);
}
__global__ void rnn_cell_bwd_kernel(const float* __restrict__ lin, const float* __restrict__ y,
                                    const float* __restrict__ dy, float* __restrict__ dlin, int64_t n, int act) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dlin[i] = act == 0 ? (1.f - y[i] * y[i]) * dy[i] : (y[i] == lin[i] ? dy[i] : 0.f);
}

// lin (B, 4H) -> gates (B, 4H) = [f | i | o | tanh g] (saved), tc (B, H) = tanh(c'), hc (B, 2H) = [h' | c']
__global__ void lstm_cell_fwd_kernel(const float* __restrict__ lin, const float* __restrict__ c,
                                     float* __restrict__ gates, float* __restrict__ tc, float* __restrict__ hc,
                                     int64_t n, int H) {
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = idx / H;
    const int j = (int)(idx - b * H);
    const float* l = lin + b * 4 * H;
    const float f = rc_sigmoid(l[j]), i = rc_sigmoid(l[H + j]), o = rc_sigmoid(l[2 * H + j]), g = rc_tanh(l[3 * H + j]);
    float* gs = gates + b * 4 * H;
    gs[j] = f; gs[H + j] = i; gs[2 * H + j] = o; gs[3 * H + j] = g;
    const float cn = f * c[idx] + i * g, t = rc_tanh(cn);
    tc[idx] = t;
    hc[b * 2 * H + j] = o * t;
    hc[b * 2 * H + H + j] = cn;
  }
}

// dhc (B, 2H) = [dh' | dc' from later consumers]  ->  dlin (B, 4H), dc_prev (B, H)
__global__ void lstm_cell_bwd_kernel(const float* __restrict__ dhc, const float* __restrict__ gates,
                                     const float* __restrict__ tc, const float* __restrict__ c,
                                     float* __restrict__ dlin, float* __restrict__ dc_prev, int64_t n, int H) {
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = idx / H;
    const int j = (int)(idx - b * H);
    const float* gs = gates + b * 4 * H;
    const float f = gs[j], i = gs[H + j], o = gs[2 * H + j], g = gs[3 * H + j], t = tc[idx];
    const float dh = dhc[b * 2 * H + j];
    const float dc = dhc[b * 2 * H + H + j] + dh * o * (1.f - t * t);
    float* dl = dlin + b * 4 * H;
    dl[j] = dc * c[idx] * f * (1.f - f);
    dl[H + j] = dc * g * i * (1.f - i);
    dl[2 * H + j] = dh * t * o * (1.f - o);
    dl[3 * H + j] = dc * i * (1.f - g * g);
    dc_prev[idx] = dc * f;
  }
}

static inline int rc_grid(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 4096) b = 4096; if (b < 1) b = 1;
  return (int)b;
}

extern "C" {

int pdn_rnn_cell_fwd_f32(const float* lin, float* y, int64_t n, int act, void* stream) {
  if (n == 0) return PDN_OK;
  PDN_CHECK_ARG(lin && y && (act == 0 || act == 1), "pdn_rnn_cell_fwd_f32: bad arguments");
  hipLaunchKernelGGL(rnn_cell_fwd_kernel, dim3(rc_grid(n)), dim3(256), 0, (hipStream_t)stream, lin, y, n, act);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

int pdn_rnn_cell_bwd_f32(const float* lin, const float* y, const float* dy, float* dlin, int64_t n, int act,
                         void* stream) {
  if (n == 0) return PDN_OK;
  PDN_CHECK_ARG(lin && y && dy && dlin && (act == 0 || act == 1), "pdn_rnn_cell_bwd_f32: bad arguments");
  hipLaunchKernelGGL(rnn_cell_bwd_kernel, dim3(rc_grid(n)), dim3(256), 0, (hipStream_t)stream, lin, y, dy, dlin, n, act);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

int pdn_lstm_cell_fwd_f32(const float* lin, const float* c, float* gates, float* tanh_c, float* hc, int64_t B, int H,
                          void* stream) {
  if (B == 0 || H == 0) return PDN_OK;
  PDN_CHECK_ARG(lin && c && gates && tanh_c && hc, "pdn_lstm_cell_fwd_f32: null operand");
  hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(rc_grid(B * H)), dim3(256), 0, (hipStream_t)stream, lin, c, gates,
                     tanh_c, hc, B * H, H);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

int pdn_lstm_cell_bwd_f32(const float* dhc, const float* gates, const float* tanh_c, const float* c, float* dlin,
                          float* dc_prev, int64_t B, int H, void* stream) {
  if (B == 0 || H == 0) return PDN_OK;
  PDN_CHECK_ARG(dhc && gates && tanh_c && c && dlin && dc_prev, "pdn_lstm_cell_bwd_f32: null operand");
  hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(rc_grid(B * H)), dim3(256), 0, (hipStream_t)stream, dhc, gates, tanh_c,
                     c, dlin, dc_prev, B * H, H);
  PDN_LAUNCH_CHECK();
  return PDN_OK;
}

}  // extern "C"
