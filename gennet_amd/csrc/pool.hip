// MaxPooling2D(pool_size=(2,1)) of the discriminator's `maxpool = True` configuration (bbhMahoGANy.py:426, :444, :453, ...): the maximum over pairs of rows
// along H of an (B, H, R) tensor (R = W * C floats per row; strides = pool_size, padding 'valid': an odd last row is dropped).  HBM-bound: one read of x,
// half a write.  Backward routes dy to the row that held the maximum -- on a tie to the FIRST row of the pair, as TensorFlow's max-pool gradient does
// (its kernels keep the first maximum; ties are common here: both rows of a pair zeroed by the Dropout in front) -- and zero to the other (and to a
// dropped last row).
#include "common.h"

namespace gn {

__global__ __launch_bounds__(256) void maxpool_h2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t total, int H, int Ho, int R) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;      // one float of y (B, Ho, R)
  if (i >= total) return;
  const int r = (int)(i % (size_t)R);
  const size_t q = i / (size_t)R;
  const int ho = (int)(q % (size_t)Ho);
  const size_t b = q / (size_t)Ho;
  const float* p = x + ((b * H + 2 * (size_t)ho) * R + r);
  const float a = p[0], c = p[R];
  y[i] = c > a ? c : a;
}

__global__ __launch_bounds__(256) void maxpool_h2_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx, size_t total, int H,
                                                             int Ho, int R) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;      // one float of dx (B, H, R)
  if (i >= total) return;
  const int r = (int)(i % (size_t)R);
  const size_t q = i / (size_t)R;
  const int h = (int)(q % (size_t)H);
  const size_t b = q / (size_t)H;
  const int ho = h >> 1;
  float v = 0.f;
  if (ho < Ho) {
    const float* p = x + ((b * H + 2 * (size_t)ho) * R + r);
    const bool second = p[R] > p[0];
    if (second == (bool)(h & 1)) v = dy[(b * Ho + ho) * R + r];
  }
  dx[i] = v;
}

}  // namespace gn

extern "C" int gn_maxpool_h2_fwd(const float* x, float* y, int B, int H, int R, void* stream) {
  GN_REQUIRE(x && y, "maxpool_h2_fwd: null pointer");
  GN_REQUIRE(B >= 0 && H >= 2 && R > 0, "maxpool_h2_fwd: bad shape (H %d >= 2)", H);
  const size_t total = (size_t)B * (H / 2) * R;
  if (total == 0) return GN_OK;
  hipLaunchKernelGGL(gn::maxpool_h2_fwd_kernel, dim3(gn::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, total, H, H / 2, R);
  return gn::check_launch("maxpool_h2_fwd");
}

extern "C" int gn_maxpool_h2_bwd(const float* dy, const float* x, float* dx, int B, int H, int R, void* stream) {
  GN_REQUIRE(dy && x && dx, "maxpool_h2_bwd: null pointer");
  GN_REQUIRE(B >= 0 && H >= 2 && R > 0, "maxpool_h2_bwd: bad shape (H %d >= 2)", H);
  const size_t total = (size_t)B * H * R;
  if (total == 0) return GN_OK;
  hipLaunchKernelGGL(gn::maxpool_h2_bwd_kernel, dim3(gn::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, dy, x, dx, total, H, H / 2, R);
  return gn::check_launch("maxpool_h2_bwd");
}
