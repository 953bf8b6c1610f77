// PReLU (keras.layers.advanced_activations.PReLU; bbhMahoGANy.py:39, reachable through act = 'prelu' at :237-286, :315-325):
//   y[b, f] = x > 0 ? x : alpha[f] * x          alpha has the shape of one sample (no shared axes), zeros at initialisation
// Keras writes it as relu(x) - alpha * relu(-x); its gradient is 0 at x == 0 exactly (both relus sit on their kink), so
//   dx = dy * (x > 0 ? 1 : x < 0 ? alpha : 0)    dalpha[f] = sum_b dy[b, f] * min(x[b, f], 0)
// HBM-bound streaming: one thread per float4 of features, loop over the batch (coalesced across features; alpha and the
// dalpha accumulators stay in registers).  The batch sum runs in b order in fp32, like the other bias-type reductions.
#include "common.h"

namespace gn {

__global__ __launch_bounds__(256) void prelu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ alpha, float* __restrict__ y, int B, size_t F4) {
  const size_t f = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (f >= F4) return;
  const float4 a = reinterpret_cast<const float4*>(alpha)[f];
  for (int b = blockIdx.y; b < B; b += gridDim.y) {
    float4 v = reinterpret_cast<const float4*>(x)[(size_t)b * F4 + f];
    v.x = v.x > 0.f ? v.x : a.x * v.x; v.y = v.y > 0.f ? v.y : a.y * v.y;
    v.z = v.z > 0.f ? v.z : a.z * v.z; v.w = v.w > 0.f ? v.w : a.w * v.w;
    reinterpret_cast<float4*>(y)[(size_t)b * F4 + f] = v;
  }
}

__device__ __forceinline__ float prelu_dx(float g, float x, float a) { return x > 0.f ? g : (x < 0.f ? g * a : 0.f); }

__global__ __launch_bounds__(256) void prelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ alpha,
                                                        float* __restrict__ dx, float* __restrict__ dalpha, int B, size_t F4) {
  const size_t f = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (f >= F4) return;
  const float4 a = reinterpret_cast<const float4*>(alpha)[f];
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int b = 0; b < B; ++b) {
    const float4 g = reinterpret_cast<const float4*>(dy)[(size_t)b * F4 + f];
    const float4 v = reinterpret_cast<const float4*>(x)[(size_t)b * F4 + f];
    float4 o;
    o.x = prelu_dx(g.x, v.x, a.x); o.y = prelu_dx(g.y, v.y, a.y); o.z = prelu_dx(g.z, v.z, a.z); o.w = prelu_dx(g.w, v.w, a.w);
    if (dx) reinterpret_cast<float4*>(dx)[(size_t)b * F4 + f] = o;
    s.x = fmaf(g.x, fminf(v.x, 0.f), s.x); s.y = fmaf(g.y, fminf(v.y, 0.f), s.y);
    s.z = fmaf(g.z, fminf(v.z, 0.f), s.z); s.w = fmaf(g.w, fminf(v.w, 0.f), s.w);
  }
  if (dalpha) reinterpret_cast<float4*>(dalpha)[f] = s;
}

}  // namespace gn

extern "C" {

int gn_prelu_fwd(const float* x, const float* alpha, float* y, int B, size_t F, void* stream) {
  GN_REQUIRE(x && alpha && y, "prelu_fwd: null pointer");
  GN_REQUIRE(B >= 0 && F > 0 && F % 4 == 0, "prelu_fwd: bad shape (B %d, F %zu; F must be a multiple of 4)", B, F);
  if (B == 0) return GN_OK;
  const unsigned gx = gn::cdiv(F / 4, 256);
  unsigned gy = 1;                                   // few features: spread the batch over blockIdx.y to fill the chip
  while (gx * gy < 2048 && gy < (unsigned)B) gy *= 2;
  hipLaunchKernelGGL(gn::prelu_fwd_kernel, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, x, alpha, y, B, F / 4);
  return gn::check_launch("prelu_fwd");
}

int gn_prelu_bwd(const float* dy, const float* x, const float* alpha, float* dx, float* dalpha, int B, size_t F, void* stream) {
  GN_REQUIRE(dy && x && alpha && (dx || dalpha), "prelu_bwd: null pointer");
  GN_REQUIRE(B > 0 && F > 0 && F % 4 == 0, "prelu_bwd: bad shape (B %d, F %zu; F must be a multiple of 4)", B, F);
  hipLaunchKernelGGL(gn::prelu_bwd_kernel, dim3(gn::cdiv(F / 4, 256)), dim3(256), 0, (hipStream_t)stream, dy, x, alpha, dx, dalpha, B, F / 4);
  return gn::check_launch("prelu_bwd");
}

}  // extern "C"
