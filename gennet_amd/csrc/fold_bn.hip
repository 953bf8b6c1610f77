// Inference-phase BatchNormalization folded into the preceding convolution (generator.predict, bbhMahoGANy.py:1248, :1330):
//   BN_infer(conv(x; W, b)) = conv(x; W * scale, b * scale + shift),   scale = gamma / sqrt(moving_var + eps), shift = beta - moving_mean * scale
// so the predict path runs conv + activation in one kernel and never writes the pre-BN tensor.  Used in the inference phase only
// (the training phase normalises with batch statistics); equal to the unfolded result up to one fp32 rounding per weight.
#include "common.h"

namespace gn {

__global__ __launch_bounds__(256) void fold_bn_kernel(const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, float* __restrict__ w_out, float* __restrict__ b_out, size_t rows, int C4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;      // one float4 of one weight row; rows = taps * Cin
  const size_t total = rows * C4;
  if (i < total) {
    const int c4 = (int)(i % C4);
    const float4 s = reinterpret_cast<const float4*>(scale)[c4];
    float4 v = reinterpret_cast<const float4*>(w)[i];
    v.x *= s.x; v.y *= s.y; v.z *= s.z; v.w *= s.w;
    reinterpret_cast<float4*>(w_out)[i] = v;
  }
  if (i < (size_t)C4) {
    const float4 s = reinterpret_cast<const float4*>(scale)[i], t = reinterpret_cast<const float4*>(shift)[i];
    float4 v = b ? reinterpret_cast<const float4*>(b)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    v.x = fmaf(v.x, s.x, t.x); v.y = fmaf(v.y, s.y, t.y); v.z = fmaf(v.z, s.z, t.z); v.w = fmaf(v.w, s.w, t.w);
    reinterpret_cast<float4*>(b_out)[i] = v;
  }
}

}  // namespace gn

extern "C" int gn_conv_fold_bn(const float* w, const float* bias, const float* scale, const float* shift, float* w_out, float* bias_out, size_t rows, int Cout,
                               void* stream) {
  GN_REQUIRE(w && scale && shift && w_out && bias_out, "conv_fold_bn: null pointer");
  GN_REQUIRE(rows > 0 && Cout > 0 && Cout % 4 == 0, "conv_fold_bn: bad shape (rows %zu, Cout %d; Cout must be a multiple of 4)", rows, Cout);
  const size_t total = rows * (size_t)(Cout / 4);
  hipLaunchKernelGGL(gn::fold_bn_kernel, dim3(gn::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, bias, scale, shift, w_out, bias_out, rows, Cout / 4);
  return gn::check_launch("conv_fold_bn");
}
