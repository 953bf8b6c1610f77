// BatchNormalization apply + activation + Dropout with the keep-mask GENERATED in the same pass (generator hidden layers,
// bbhMahoGANy.py:235-239, :251-255, ...): y = mask ? act(x * scale + shift) / (1 - rate) : 0, mask written out for the backward
// pass.  The draw is gn_dropout_mask's, bit for bit (Philox4x32-10, element k uses counter offset + k / 4, lane k % 4; keep iff
// u >= rate): one float4 of activations = one Philox call.  The pass is HBM-bound, so the ten Philox rounds ride for free -- the
// separate mask kernel (ALU-bound, ~1 % of the training step) and one read of the mask disappear.
#include "common.h"

namespace gn {

__global__ void bn_apply_dropgen_kernel(const float4* __restrict__ x, const float4* __restrict__ scale, const float4* __restrict__ shift,
                                        uchar4* __restrict__ mask_out, float4* __restrict__ y, size_t n4, int C4, int act, float p, float rate,
                                        float keep_scale, uint64_t seed, uint64_t offset, const uint64_t* __restrict__ base) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  if (base) offset += *base;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const int c = (int)(i % C4);
    const float4 v = x[i], sc = scale[c], sh = shift[c];
    const Philox4 r = philox4x32_10(offset + i, seed);
    uchar4 m;
    m.x = u01_24(r.v[0]) >= rate ? 1 : 0; m.y = u01_24(r.v[1]) >= rate ? 1 : 0;
    m.z = u01_24(r.v[2]) >= rate ? 1 : 0; m.w = u01_24(r.v[3]) >= rate ? 1 : 0;
    float4 o;
    o.x = act_apply(fmaf(v.x, sc.x, sh.x), act, p); o.y = act_apply(fmaf(v.y, sc.y, sh.y), act, p);
    o.z = act_apply(fmaf(v.z, sc.z, sh.z), act, p); o.w = act_apply(fmaf(v.w, sc.w, sh.w), act, p);
    o.x = m.x ? o.x * keep_scale : 0.f; o.y = m.y ? o.y * keep_scale : 0.f;
    o.z = m.z ? o.z * keep_scale : 0.f; o.w = m.w ? o.w * keep_scale : 0.f;
    mask_out[i] = m;
    y[i] = o;
  }
}

}  // namespace gn

extern "C" int gn_bn_apply_dropgen(const float* x, const float* scale, const float* shift, uint8_t* mask_out, float* y, size_t rows, int C, int act, float p,
                                   float rate, uint64_t seed, uint64_t offset, void* stream) {
  GN_REQUIRE(x && scale && shift && mask_out && y, "bn_apply_dropgen: null pointer");
  GN_REQUIRE(C > 0 && C % 4 == 0, "bn_apply_dropgen: C %d must be a positive multiple of 4", C);
  GN_REQUIRE(rate >= 0.f && rate < 1.f, "bn_apply_dropgen: bad rate %f", rate);
  const size_t n4 = rows * (size_t)(C / 4);
  if (!n4) return GN_OK;
  size_t blocks = (n4 + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(gn::bn_apply_dropgen_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)x, (const float4*)scale,
                     (const float4*)shift, (uchar4*)mask_out, (float4*)y, n4, C / 4, act, p, rate, 1.0f / (1.0f - rate), seed, offset, gn::rng_base());
  return gn::check_launch("bn_apply_dropgen");
}
