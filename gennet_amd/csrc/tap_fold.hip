// Conv1D with more than 5 taps (the reference's own `filtsize = 5 # 10 is best`, bbhMahoGANy.py:228; the 16-tap layers of its saved Keras models) on the
// <= 5-tap matrix-core kernels, without a new kernel family: with G = ceil(k/5) tap groups of h = ceil(k/G) taps,
//   y[b,m,:] = sum_{t<k} x[b, s*m + t - pl, :] W[t]  =  sum_{t<h} x2[b, s*m + t, :] W2[t],
//   x2[b, j, g*Cin:(g+1)*Cin] = x[b, j - pl + g*h, :]   (zero outside 0 <= row < L),  j in [0, L + pl),  g in [0, G)
//   W2[t, g*Cin:(g+1)*Cin, :] = W[t + g*h]               (zero where t + g*h >= k)
// i.e. every further group of taps becomes a further group of input channels of an h-tap convolution over the shifted input, and the partial sums
// accumulate inside the kernel's own K loop (k = 10, unit stride: a 5-tap layer over 2*Cin channels, so it takes the transform-domain kernels).  The left
// padding is materialised in x2 (pad_left of the h-tap conv is 0): a row of x2 left of the input still carries x[j - pl + g*h].  The gradients come back
// the same way: dW from the h-tap weight gradient over x2 (tapunfold_dw), dx[b,l,:] = sum_g dx2[b, l + pl - g*h, g*Cin:(g+1)*Cin] (tapunfold_dx).
// One pass over x each way (HBM-bound, (1 + G) x the input's bytes); the h-tap convolution on G*Cin channels does G*h >= k taps' worth of multiplies.
#include "common.h"

namespace gn {

static inline int tap_groups(int k) { return (k + 4) / 5; }
static inline int tap_group_len(int k) { return (k + tap_groups(k) - 1) / tap_groups(k); }

__global__ __launch_bounds__(256) void tapfold_x_kernel(const float* __restrict__ x, float* __restrict__ x2, size_t total, int L, int L2, int Cin, int G, int h, int pl) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;      // one float of x2 (B, L2, G*Cin)
  if (i >= total) return;
  const int c2 = (int)(i % (size_t)(G * Cin));
  const size_t r = i / (size_t)(G * Cin);
  const int j = (int)(r % (size_t)L2);
  const size_t b = r / (size_t)L2;
  const int g = c2 / Cin, c = c2 - g * Cin;
  const int row = j - pl + g * h;
  x2[i] = (row >= 0 && row < L) ? x[(b * L + row) * Cin + c] : 0.f;
}

__global__ __launch_bounds__(256) void tapunfold_dx_kernel(const float* __restrict__ dx2, float* __restrict__ dx, size_t total, int L, int L2, int Cin, int G, int h, int pl) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;      // one float of dx (B, L, Cin)
  if (i >= total) return;
  const int c = (int)(i % (size_t)Cin);
  const size_t r = i / (size_t)Cin;
  const int l = (int)(r % (size_t)L);
  const size_t b = r / (size_t)L;
  float v = 0.f;
  for (int g = 0; g < G; ++g) {                                  // fixed order: group 0 first
    const int row = l + pl - g * h;
    if (row >= 0) v += dx2[(b * L2 + row) * ((size_t)G * Cin) + (size_t)g * Cin + c];
  }
  dx[i] = v;
}

// DIR 0: W (k, Cin, Cout) -> W2 (h, G*Cin, Cout);  DIR 1: dW2 -> dW (the padded taps are dropped)
template <int DIR, typename TS, typename TD>
__global__ __launch_bounds__(256) void tapfold_w_kernel(TS src, TD dst, size_t total, int k, int G, int h, int Cin, int Cout) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;      // one float of W2
  if (i >= total) return;
  const int co = (int)(i % (size_t)Cout);
  const size_t r = i / (size_t)Cout;
  const int c2 = (int)(r % (size_t)(G * Cin));
  const int t = (int)(r / (size_t)(G * Cin));
  const int g = c2 / Cin, ci = c2 - g * Cin, tap = t + g * h;
  const size_t j = ((size_t)tap * Cin + ci) * Cout + co;
  if (DIR == 0) dst[i] = tap < k ? src[j] : 0.f;
  else if (tap < k) dst[j] = src[i];
}

}  // namespace gn

#define GN_TAPFOLD_SHAPE(who) GN_REQUIRE(Cin > 0 && k >= 6 && k <= 40, who ": bad shape (6 <= k <= 40)")

extern "C" int gn_conv1d_tap_groups(int k, int* groups, int* taps) {
  GN_REQUIRE(groups && taps && k >= 1, "conv1d_tap_groups: bad arguments");
  *groups = gn::tap_groups(k);
  *taps = gn::tap_group_len(k);
  return GN_OK;
}

extern "C" int gn_conv1d_tapfold_x(const float* x, float* x2, int B, int L, int Cin, int k, int pad_left, void* stream) {
  GN_REQUIRE(x && x2, "conv1d_tapfold_x: null pointer");
  GN_TAPFOLD_SHAPE("conv1d_tapfold_x");
  GN_REQUIRE(B >= 0 && L > 0 && pad_left >= 0 && pad_left < k, "conv1d_tapfold_x: bad shape");
  const int G = gn::tap_groups(k), h = gn::tap_group_len(k), L2 = L + pad_left;
  const size_t total = (size_t)B * L2 * G * Cin;
  if (total == 0) return GN_OK;
  hipLaunchKernelGGL(gn::tapfold_x_kernel, dim3(gn::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, x2, total, L, L2, Cin, G, h, pad_left);
  return gn::check_launch("conv1d_tapfold_x");
}

extern "C" int gn_conv1d_tapunfold_dx(const float* dx2, float* dx, int B, int L, int Cin, int k, int pad_left, void* stream) {
  GN_REQUIRE(dx2 && dx, "conv1d_tapunfold_dx: null pointer");
  GN_TAPFOLD_SHAPE("conv1d_tapunfold_dx");
  GN_REQUIRE(B >= 0 && L > 0 && pad_left >= 0 && pad_left < k, "conv1d_tapunfold_dx: bad shape");
  const int G = gn::tap_groups(k), h = gn::tap_group_len(k), L2 = L + pad_left;
  const size_t total = (size_t)B * L * Cin;
  if (total == 0) return GN_OK;
  hipLaunchKernelGGL(gn::tapunfold_dx_kernel, dim3(gn::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, dx2, dx, total, L, L2, Cin, G, h, pad_left);
  return gn::check_launch("conv1d_tapunfold_dx");
}

extern "C" int gn_conv1d_tapfold_w(const float* w, float* w2, int k, int Cin, int Cout, void* stream) {
  GN_REQUIRE(w && w2, "conv1d_tapfold_w: null pointer");
  GN_TAPFOLD_SHAPE("conv1d_tapfold_w");
  GN_REQUIRE(Cout > 0, "conv1d_tapfold_w: bad shape");
  const int G = gn::tap_groups(k), h = gn::tap_group_len(k);
  const size_t total = (size_t)h * G * Cin * Cout;
  hipLaunchKernelGGL((gn::tapfold_w_kernel<0, const float*, float*>), dim3(gn::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, w2, total, k, G, h, Cin, Cout);
  return gn::check_launch("conv1d_tapfold_w");
}

extern "C" int gn_conv1d_tapunfold_dw(const float* dw2, float* dw, int k, int Cin, int Cout, void* stream) {
  GN_REQUIRE(dw2 && dw, "conv1d_tapunfold_dw: null pointer");
  GN_TAPFOLD_SHAPE("conv1d_tapunfold_dw");
  GN_REQUIRE(Cout > 0, "conv1d_tapunfold_dw: bad shape");
  const int G = gn::tap_groups(k), h = gn::tap_group_len(k);
  const size_t total = (size_t)h * G * Cin * Cout;
  hipLaunchKernelGGL((gn::tapfold_w_kernel<1, const float*, float*>), dim3(gn::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, dw2, dw, total, k, G, h, Cin, Cout);
  return gn::check_launch("conv1d_tapunfold_dw");
}
