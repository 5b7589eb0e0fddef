// csrc/conv_quad.hip: 3 x 3 / stride 1 / pad 1 convolutions of compile-time geometry (the LeNet shapes of
// examples/pydynet/mnist.py:82-98), called from the entry points of csrc/conv_direct.hip when the shape matches.
#pragma once
bool conv_quad_fwd_supported(int C, int H, int W, int O, int k, int stride, int pad);
int conv_quad_relu_pool_fwd(const float* x, const float* w, const float* bias, float* pooled, unsigned* mask, int N, int C,
                            int H, int W, int O, void* stream);
bool conv_quad_dgrad_supported(int C, int H, int W, int O, int k, int stride, int pad);
int conv_quad_relu_pool_bwd_data(const float* dpooled, const unsigned* mask, const float* w, float* dx, int N, int C, int H,
                                 int W, int O, void* stream);
bool conv_quad_wgrad_supported(int C, int H, int W, int O, int k, int stride, int pad);
int conv_quad_relu_pool_bwd_weight(const float* x, const float* dpooled, const unsigned* mask, float* dw, float* db,
                                   int accumulate, int N, int C, int H, int W, int O, void* workspace,
                                   int64_t workspace_bytes, void* stream);
