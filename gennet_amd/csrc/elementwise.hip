// Streaming (HBM-bound) kernels: activations, dropout, upsample, subtract-stack, RNG fills, gather, BatchNorm passes,
// losses and the fused Adam update.  All are float4-vectorised grid-stride loops sized to ~8 blocks/CU.
#include "common.h"

namespace gn {

static inline unsigned stream_grid(size_t n_items, int block = 256) {
  size_t g = (n_items + block - 1) / block;
  if (g > 256 * 8) g = 256 * 8;
  if (g < 1) g = 1;
  return (unsigned)g;
}

// ---------------------------------------------------------------------------------------------
// activations
// ---------------------------------------------------------------------------------------------
__global__ void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, int act, float p) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 v = reinterpret_cast<const float4*>(x)[i];
    v.x = act_apply(v.x, act, p); v.y = act_apply(v.y, act, p); v.z = act_apply(v.z, act, p); v.w = act_apply(v.w, act, p);
    reinterpret_cast<float4*>(y)[i] = v;
  }
  for (size_t i = (n4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = act_apply(x[i], act, p);
}

__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx, size_t n, int act, float p) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 g = reinterpret_cast<const float4*>(dy)[i];
    const float4 v = reinterpret_cast<const float4*>(y)[i];
    float4 o;
    o.x = g.x * act_grad_from_y(v.x, act, p); o.y = g.y * act_grad_from_y(v.y, act, p);
    o.z = g.z * act_grad_from_y(v.z, act, p); o.w = g.w * act_grad_from_y(v.w, act, p);
    reinterpret_cast<float4*>(dx)[i] = o;
  }
  for (size_t i = (n4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dx[i] = dy[i] * act_grad_from_y(y[i], act, p);
}

// fused backward of [activation -> inverted dropout] expressed through the layer output y (post-dropout):
// dx = mask ? dy * keep_scale * act'(y / keep_scale) : 0
__global__ void act_dropout_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, const uint8_t* __restrict__ mask, float* __restrict__ dx,
                                       size_t n, int act, float p, float keep_scale) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t n4 = n >> 2;
  const float inv = 1.0f / keep_scale;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 g = reinterpret_cast<const float4*>(dy)[i];
    const float4 v = reinterpret_cast<const float4*>(y)[i];
    const uchar4 k = reinterpret_cast<const uchar4*>(mask)[i];
    float4 o;
    o.x = k.x ? g.x * keep_scale * act_grad_from_y(v.x * inv, act, p) : 0.f; o.y = k.y ? g.y * keep_scale * act_grad_from_y(v.y * inv, act, p) : 0.f;
    o.z = k.z ? g.z * keep_scale * act_grad_from_y(v.z * inv, act, p) : 0.f; o.w = k.w ? g.w * keep_scale * act_grad_from_y(v.w * inv, act, p) : 0.f;
    reinterpret_cast<float4*>(dx)[i] = o;
  }
  for (size_t i = (n4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    dx[i] = mask[i] ? dy[i] * keep_scale * act_grad_from_y(y[i] * inv, act, p) : 0.f;
}
int act_dropout_bwd(const float* dy, const float* y, const uint8_t* mask, float* dx, size_t n, int act, float p, float rate, hipStream_t s) {
  if (n == 0) return GN_OK;
  hipLaunchKernelGGL(act_dropout_bwd_kernel, dim3(stream_grid(n / 4 + 1)), dim3(256), 0, s, dy, y, mask, dx, n, act, p, 1.0f / (1.0f - rate));
  return check_launch("act_dropout_bwd");
}

int act_fwd(const float* x, float* y, size_t n, int act, float p, hipStream_t s) {
  if (n == 0) return GN_OK;
  hipLaunchKernelGGL(act_fwd_kernel, dim3(stream_grid(n / 4 + 1)), dim3(256), 0, s, x, y, n, act, p);
  return check_launch("act_fwd");
}
int act_bwd(const float* dy, const float* y, float* dx, size_t n, int act, float p, hipStream_t s) {
  if (n == 0) return GN_OK;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(stream_grid(n / 4 + 1)), dim3(256), 0, s, dy, y, dx, n, act, p);
  return check_launch("act_bwd");
}

// ---------------------------------------------------------------------------------------------
// dropout: one Philox call yields 4 uniforms -> 4 consecutive mask bytes
// ---------------------------------------------------------------------------------------------
__global__ void dropout_mask_kernel(uint8_t* __restrict__ mask, size_t n, float rate, uint64_t seed, uint64_t offset, const uint64_t* __restrict__ base) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t n4 = (n + 3) >> 2;
  if (base) offset += *base;
  const bool aligned = (reinterpret_cast<uintptr_t>(mask) & 3) == 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const Philox4 r = philox4x32_10(offset + i, seed);
    if (aligned && 4 * i + 3 < n) {                          // the four bytes as one dword store (byte stores: 4 instructions, a quarter of each sector)
      unsigned w = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) w |= (u01_24(r.v[e]) >= rate ? 1u : 0u) << (8 * e);
      reinterpret_cast<unsigned*>(mask)[i] = w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const size_t k = 4 * i + e;
        if (k < n) mask[k] = u01_24(r.v[e]) >= rate ? 1 : 0;
      }
    }
  }
}

__global__ void dropout_apply_kernel(const float* __restrict__ x, const uint8_t* __restrict__ mask, float* __restrict__ y, size_t n, float scale) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = mask[i] ? x[i] * scale : 0.f;
}

int dropout_mask(uint8_t* mask, size_t n, float rate, uint64_t seed, uint64_t offset, hipStream_t s) {
  if (n == 0) return GN_OK;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(stream_grid(n / 4 + 1)), dim3(256), 0, s, mask, n, rate, seed, offset, rng_base());
  return check_launch("dropout_mask");
}
int dropout_apply(const float* x, const uint8_t* mask, float* y, size_t n, float rate, hipStream_t s) {
  if (n == 0) return GN_OK;
  hipLaunchKernelGGL(dropout_apply_kernel, dim3(stream_grid(n)), dim3(256), 0, s, x, mask, y, n, 1.0f / (1.0f - rate));
  return check_launch("dropout_apply");
}

// ---------------------------------------------------------------------------------------------
// UpSampling1D(2), MyLayer stack, gather, axpy, RNG fills
// ---------------------------------------------------------------------------------------------
__global__ void upsample2_fwd_kernel(const float4* __restrict__ x, float4* __restrict__ y, size_t rows, int C4) {
  const size_t n = rows * C4, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const size_t r = i / C4, c = i % C4;
    const float4 v = x[i];
    y[(2 * r) * C4 + c] = v;
    y[(2 * r + 1) * C4 + c] = v;
  }
}
__global__ void upsample2_bwd_kernel(const float4* __restrict__ dy, float4* __restrict__ dx, size_t rows, int C4) {
  const size_t n = rows * C4, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const size_t r = i / C4, c = i % C4;
    const float4 a = dy[(2 * r) * C4 + c], b = dy[(2 * r + 1) * C4 + c];
    dx[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  }
}
int upsample2_fwd(const float* x, float* y, int B, int L, int C, hipStream_t s) {
  if (C % 4) { set_error("upsample2: C %d %% 4 != 0", C); return GN_EINVAL; }
  const size_t rows = (size_t)B * L;
  if (!rows) return GN_OK;
  hipLaunchKernelGGL(upsample2_fwd_kernel, dim3(stream_grid(rows * (C / 4))), dim3(256), 0, s, (const float4*)x, (float4*)y, rows, C / 4);
  return check_launch("upsample2_fwd");
}
int upsample2_bwd(const float* dy, float* dx, int B, int L, int C, hipStream_t s) {
  if (C % 4) { set_error("upsample2: C %d %% 4 != 0", C); return GN_EINVAL; }
  const size_t rows = (size_t)B * L;
  if (!rows) return GN_OK;
  hipLaunchKernelGGL(upsample2_bwd_kernel, dim3(stream_grid(rows * (C / 4))), dim3(256), 0, s, (const float4*)dy, (float4*)dx, rows, C / 4);
  return check_launch("upsample2_bwd");
}

__global__ void subtract_stack_fwd_kernel(const float* __restrict__ x, const float* __restrict__ ev, float2* __restrict__ img, size_t total, int n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const float v = x[i];
    img[i] = make_float2(v, ev[i % n] - v);
  }
}
__global__ void subtract_stack_bwd_kernel(const float2* __restrict__ d, float* __restrict__ dx, size_t total) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const float2 v = d[i];
    dx[i] = v.x - v.y;
  }
}
// user-defined Layer.call of the form K.stack([a0*x + b0, a1*x + b1], axis=2) (keras/backend.py lowers it here); MyLayer is (1, 0, -1, event)
__global__ void affine_stack_fwd_kernel(const float* __restrict__ x, const float* __restrict__ b0, const float* __restrict__ b1, float a0, float a1,
                                        float2* __restrict__ img, size_t total, int n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const float v = x[i];
    const int t = (int)(i % n);
    img[i] = make_float2(a0 * v + (b0 ? b0[t] : 0.f), a1 * v + (b1 ? b1[t] : 0.f));
  }
}
__global__ void affine_stack_bwd_kernel(const float2* __restrict__ d, float a0, float a1, float* __restrict__ dx, size_t total) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const float2 v = d[i];
    dx[i] = a0 * v.x + a1 * v.y;
  }
}
int affine_stack_fwd(const float* x, const float* b0, const float* b1, float a0, float a1, float* img, int B, int n, hipStream_t s) {
  const size_t total = (size_t)B * n;
  if (!total) return GN_OK;
  hipLaunchKernelGGL(affine_stack_fwd_kernel, dim3(stream_grid(total)), dim3(256), 0, s, x, b0, b1, a0, a1, (float2*)img, total, n);
  return check_launch("affine_stack_fwd");
}
int affine_stack_bwd(const float* dimg, float a0, float a1, float* dx, int B, int n, hipStream_t s) {
  const size_t total = (size_t)B * n;
  if (!total) return GN_OK;
  hipLaunchKernelGGL(affine_stack_bwd_kernel, dim3(stream_grid(total)), dim3(256), 0, s, (const float2*)dimg, a0, a1, dx, total);
  return check_launch("affine_stack_bwd");
}
int subtract_stack_fwd(const float* x, const float* ev, float* img, int B, int n, hipStream_t s) {
  const size_t total = (size_t)B * n;
  if (!total) return GN_OK;
  hipLaunchKernelGGL(subtract_stack_fwd_kernel, dim3(stream_grid(total)), dim3(256), 0, s, x, ev, (float2*)img, total, n);
  return check_launch("subtract_stack_fwd");
}
int subtract_stack_bwd(const float* dimg, float* dx, int B, int n, hipStream_t s) {
  const size_t total = (size_t)B * n;
  if (!total) return GN_OK;
  hipLaunchKernelGGL(subtract_stack_bwd_kernel, dim3(stream_grid(total)), dim3(256), 0, s, (const float2*)dimg, dx, total);
  return check_launch("subtract_stack_bwd");
}

// bbhMahoGANy.py:1268-1289: discriminator batch [real | fake] as width-2 images, fake half in reversed sample order
__global__ void assemble_d_batch_kernel(const float* __restrict__ real, const float* __restrict__ noise, const float* __restrict__ fake,
                                        const float* __restrict__ ev, float2* __restrict__ sX, int B, int n) {
  const size_t total = (size_t)2 * B * n, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int row = (int)(i / n), t = (int)(i % n);
    if (row < B) {
      sX[i] = make_float2(real[(size_t)row * n + t], noise[(size_t)row * n + t]);
    } else {
      const float f = fake[(size_t)(2 * B - 1 - row) * n + t];
      sX[i] = make_float2(f, ev[t] - f);
    }
  }
}
int assemble_d_batch(const float* real, const float* noise, const float* fake, const float* ev, float* sX, int B, int n, hipStream_t s) {
  if (!B || !n) return GN_OK;
  hipLaunchKernelGGL(assemble_d_batch_kernel, dim3(stream_grid((size_t)2 * B * n)), dim3(256), 0, s, real, noise, fake, ev, (float2*)sX, B, n);
  return check_launch("assemble_d_batch");
}

__global__ void gather_rows_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx, float* __restrict__ out, size_t rows, int width) {
  const size_t total = rows * width, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const size_t r = i / width, c = i % width;
    out[i] = src[(size_t)idx[r] * width + c];
  }
}
int gather_rows(const float* src, const int64_t* idx, float* out, int rows, int width, hipStream_t s) {
  const size_t total = (size_t)rows * width;
  if (!total) return GN_OK;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(stream_grid(total)), dim3(256), 0, s, src, idx, out, (size_t)rows, width);
  return check_launch("gather_rows");
}

__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float a, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = fmaf(a, x[i], y[i]);
}
int axpy(float* y, const float* x, float a, size_t n, hipStream_t s) {
  if (!n) return GN_OK;
  hipLaunchKernelGGL(axpy_kernel, dim3(stream_grid(n)), dim3(256), 0, s, y, x, a, n);
  return check_launch("axpy");
}

__global__ void fill_uniform_kernel(float* __restrict__ out, size_t n, float lo, float hi, uint64_t seed, uint64_t offset, const uint64_t* __restrict__ base) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t n4 = (n + 3) >> 2;
  if (base) offset += *base;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const Philox4 r = philox4x32_10(offset + i, seed);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const size_t k = 4 * i + e;
      if (k < n) out[k] = lo + (hi - lo) * u01_24(r.v[e]);
    }
  }
}
// Box-Muller on two 24-bit uniforms per pair; u1 in (0,1] so log is finite
__global__ void fill_normal_kernel(float* __restrict__ out, size_t n, float mean, float sd, uint64_t seed, uint64_t offset, const uint64_t* __restrict__ base,
                                   const float* __restrict__ sd_dev) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t n4 = (n + 3) >> 2;
  if (base) offset += *base;
  if (sd_dev) sd = *sd_dev;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const Philox4 r = philox4x32_10(offset + i, seed);
    float z[4];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float u1 = 1.0f - u01_24(r.v[2 * e]);
      const float u2 = u01_24(r.v[2 * e + 1]);
      const float rad = sqrtf(-2.0f * logf(u1));
      float sn, cs;
      sincosf(6.283185307179586f * u2, &sn, &cs);
      z[2 * e] = rad * cs;
      z[2 * e + 1] = rad * sn;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const size_t k = 4 * i + e;
      if (k < n) out[k] = mean + sd * z[e];
    }
  }
}
int fill_uniform(float* out, size_t n, float lo, float hi, uint64_t seed, uint64_t offset, hipStream_t s) {
  if (!n) return GN_OK;
  hipLaunchKernelGGL(fill_uniform_kernel, dim3(stream_grid(n / 4 + 1)), dim3(256), 0, s, out, n, lo, hi, seed, offset, rng_base());
  return check_launch("fill_uniform");
}
int fill_normal(float* out, size_t n, float mean, float sd, uint64_t seed, uint64_t offset, hipStream_t s, const float* sd_dev) {
  if (!n) return GN_OK;
  hipLaunchKernelGGL(fill_normal_kernel, dim3(stream_grid(n / 4 + 1)), dim3(256), 0, s, out, n, mean, sd, seed, offset, rng_base(), sd_dev);
  return check_launch("fill_normal");
}

// ---------------------------------------------------------------------------------------------
// column reductions in fp64 (BatchNorm statistics, bias gradients):  x viewed as (rows, C), C % 4 == 0
// grid = (column blocks, row chunks); thread = one float4 column group x one row lane; partials [chunk][NV][C] fp64.
// MODE 0: sum x                (bias gradient)
// MODE 1: sum x, sum x^2       (BN forward statistics)
// MODE 2: sum g, sum g*xhat    (BN backward statistics; g = dy through dropout and activation)
// ---------------------------------------------------------------------------------------------

// value of g for one element (shared by backward pass 1 and 2)
__device__ __forceinline__ float bn_bwd_g(float dy, float y, uint8_t keep, int act, float p, float keep_scale) {
  if (!keep) return 0.f;
  const float yact = y / keep_scale;  // undo the inverted-dropout scale to recover the activation output
  return dy * keep_scale * act_grad_from_y(yact, act, p);
}

// the same with the activation output itself (recomputed from the pre-BN tensor) instead of the stored, dropout-scaled layer output
__device__ __forceinline__ float bn_bwd_g_act(float dy, float yact, uint8_t keep, int act, float p, float keep_scale) {
  if (!keep) return 0.f;
  return dy * keep_scale * act_grad_from_y(yact, act, p);
}

// LazyDy (common.h): the 4 channels 4q..4q+3 of one row of the data gradient of a 1-filter stride-1 conv with k <= 5 taps, from its
// output gradient g and kernel; wq holds the thread's kernel columns (taps past k are zero).
__device__ __forceinline__ void lazy_dy_taps(const LazyDy& z, int C, int q, float wq[5][4]) {
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < z.k) w4 = *reinterpret_cast<const float4*>(z.w + (size_t)j * C + 4 * q);
    wq[j][0] = w4.x; wq[j][1] = w4.y; wq[j][2] = w4.z; wq[j][3] = w4.w;
  }
}
// A wave's window on g: lane l holds g[b, base + l] (0 outside [0, Lout)).  With C / 4 lanes per row a multiple of 64 the row is the same
// for all lanes of a wave, rows advance along the segment, and one 64-wide load serves ~60 / RL rows; the k values of a row come out
// of it by v_readlane.  (k vector loads of one address per row cost the address path as much as the 16-byte row loads themselves
// and doubled the kernels' time; scalar loads are not available next to the kernel's own global stores.)
struct LazyWin {
  float win;
  int base;
  unsigned b;
};
template <bool UNI>
__device__ __forceinline__ void lazy_dy4(const LazyDy& z, unsigned b, int t, const float wq[5][4], float v[4], LazyWin& w) {
  float gv[5];
  if (UNI) {
    b = __builtin_amdgcn_readfirstlane(b);
    t = __builtin_amdgcn_readfirstlane(t);
    const int uhi = t + z.pad_left, ulo = uhi - (z.k - 1);
    if (b != w.b || ulo < w.base || uhi >= w.base + 64) {              // wave-uniform
      w.b = b;
      w.base = ulo;
      const int idx = ulo + (int)(threadIdx.x & 63);
      w.win = (idx >= 0 && idx < z.Lout) ? z.g[(size_t)b * z.Lout + idx] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 5; ++j)
      gv[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w.win), max(uhi - j - w.base, 0)));      // taps past k: weight 0
  } else {
    const float* gb = z.g + (size_t)b * z.Lout;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int u = t - j + z.pad_left;
      const int uc = min(max(u, 0), z.Lout - 1);
      const float g = gb[uc];
      gv[j] = (u == uc) ? g : 0.f;
    }
  }
  v[0] = v[1] = v[2] = v[3] = 0.f;
#pragma unroll
  for (int j = 0; j < 5; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaf(gv[j], wq[j][e], v[e]);
}
// (segment, position) of row r; the row loops then step both without dividing
__device__ __forceinline__ void lazy_dy_pos(const LazyDy& z, size_t r, unsigned* b, int* t) {
  *b = (unsigned)(r / (unsigned)z.L);
  *t = (int)(r - (size_t)*b * z.L);
}
__device__ __forceinline__ void lazy_dy_step(const LazyDy& z, int step, unsigned* b, int* t) {
  *t += step;
  while (*t >= z.L) { *t -= z.L; ++*b; }
}

template <int MODE>
__global__ __launch_bounds__(256) void colred_kernel(ColRedArgs a) {
  constexpr int NV = MODE == 0 ? 1 : 2;
  const int NQ = a.C >> 2;
  const int NQc = NQ < 256 ? NQ : 256;
  const int RL = 256 / NQc;
  const int tid = threadIdx.x, ql = tid % NQc, rl = tid / NQc;
  const int q = blockIdx.x * NQc + ql;
  double s[NV][4];
#pragma unroll
  for (int v = 0; v < NV; ++v)
#pragma unroll
    for (int e = 0; e < 4; ++e) s[v][e] = 0.0;
  const size_t r_lo = (size_t)blockIdx.y * a.rows_per_chunk;
  const size_t r_hi = r_lo + a.rows_per_chunk < a.rows ? r_lo + a.rows_per_chunk : a.rows;
  if (rl < RL && q < NQ) {
    float mu[4] = {0, 0, 0, 0}, is[4] = {0, 0, 0, 0}, sc[4] = {0, 0, 0, 0}, sh[4] = {0, 0, 0, 0};
    if (MODE == 2) {
      const float4 m4 = *reinterpret_cast<const float4*>(a.mean + 4 * q), i4 = *reinterpret_cast<const float4*>(a.invstd + 4 * q);
      mu[0] = m4.x; mu[1] = m4.y; mu[2] = m4.z; mu[3] = m4.w;
      is[0] = i4.x; is[1] = i4.y; is[2] = i4.z; is[3] = i4.w;
      if (a.scale) {
        const float4 c4 = *reinterpret_cast<const float4*>(a.scale + 4 * q), h4 = *reinterpret_cast<const float4*>(a.shift + 4 * q);
        sc[0] = c4.x; sc[1] = c4.y; sc[2] = c4.z; sc[3] = c4.w;
        sh[0] = h4.x; sh[1] = h4.y; sh[2] = h4.z; sh[3] = h4.w;
      }
    }
    float wq[5][4];
    LazyWin lw = {0.f, 0, 0xffffffffu};
    const bool lazy = MODE == 2 && a.lz.g != nullptr;
    unsigned lb = 0;
    int lt = 0;
    if (lazy) {
      lazy_dy_taps(a.lz, a.C, q, wq);
      lazy_dy_pos(a.lz, r_lo + rl, &lb, &lt);
    }
    // U rows per trip with all their loads issued before the first use.  Measured on the generator's largest BatchNormalization
    // (1 M rows x 1024 channels): U = 4 is SLOWER than U = 1 (146 VGPRs, 3 waves per SIMD: 6.6 against 6.2 ms for the backward pair) --
    // the pass is bound by its arithmetic (tanh recomputation, fp64 sums), not by load latency.
    constexpr int U = 2;
    for (size_t r = r_lo + rl; r < r_hi; r += (size_t)RL * U) {
      float4 v4[U], x4[U], y4[U];
      uchar4 m4[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const size_t ru = r + (size_t)u * RL;
        ok[u] = ru < r_hi;
        const size_t o = (ok[u] ? ru : r) * a.C + 4 * q;
        if (!lazy) v4[u] = *reinterpret_cast<const float4*>(a.a + o);
        if (MODE == 2) {
          x4[u] = *reinterpret_cast<const float4*>(a.xpre + o);
          if (!a.scale) y4[u] = *reinterpret_cast<const float4*>(a.y + o);
          m4[u] = a.mask ? *reinterpret_cast<const uchar4*>(a.mask + o) : make_uchar4(1, 1, 1, 1);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (!ok[u]) break;
        float v[4];
        if (lazy) {
          if (((NQc | NQ) & 63) == 0) lazy_dy4<true>(a.lz, lb, lt, wq, v, lw);
          else lazy_dy4<false>(a.lz, lb, lt, wq, v, lw);
          lazy_dy_step(a.lz, RL, &lb, &lt);
        } else {
          v[0] = v4[u].x; v[1] = v4[u].y; v[2] = v4[u].z; v[3] = v4[u].w;
        }
        if (MODE == 0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) s[0][e] += (double)v[e];
        } else if (MODE == 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            s[0][e] += (double)v[e];
            s[1][e] += (double)v[e] * (double)v[e];
          }
        } else {
          const float xv[4] = {x4[u].x, x4[u].y, x4[u].z, x4[u].w};
          float yv[4];
          if (a.scale) {        // activation output recomputed from the pre-BN tensor: one 4-byte read per element less
#pragma unroll
            for (int e = 0; e < 4; ++e) yv[e] = act_apply(fmaf(xv[e], sc[e], sh[e]), a.act, a.act_param);
          } else {
            yv[0] = y4[u].x; yv[1] = y4[u].y; yv[2] = y4[u].z; yv[3] = y4[u].w;
          }
          const uint8_t k[4] = {m4[u].x, m4[u].y, m4[u].z, m4[u].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float g = a.scale ? bn_bwd_g_act(v[e], yv[e], k[e], a.act, a.act_param, a.keep_scale)
                                    : bn_bwd_g(v[e], yv[e], k[e], a.act, a.act_param, a.keep_scale);
            const float xh = (xv[e] - mu[e]) * is[e];
            s[0][e] += (double)g;
            s[NV - 1][e] += (double)g * (double)xh;
          }
        }
      }
    }
  }
  __shared__ double red[256 * 4];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
#pragma unroll
    for (int e = 0; e < 4; ++e) red[tid * 4 + e] = s[v][e];
    __syncthreads();
    if (rl == 0 && q < NQ) {
      double t[4] = {red[ql * 4], red[ql * 4 + 1], red[ql * 4 + 2], red[ql * 4 + 3]};
      for (int k = 1; k < RL; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] += red[(k * NQc + ql) * 4 + e];
      double* d = a.part + ((size_t)blockIdx.y * NV + v) * a.C + 4 * q;
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] = t[e];
    }
    __syncthreads();
  }
}

// sum the chunk partials: block = 32 columns x 8 chunk lanes; lane l adds chunks l, l+8, ... then the 8 lane sums are added in
// lane order (fixed order -> bitwise reproducible)
template <typename OUT, int COLS>
__global__ __launch_bounds__(256) void colred_final_kernel(const double* __restrict__ part, OUT* __restrict__ out, size_t n, int chunks) {
  constexpr int LANES = 256 / COLS;
  const int col = threadIdx.x % COLS, lane = threadIdx.x / COLS;
  const size_t i = (size_t)blockIdx.x * COLS + col;
  double s = 0.0;
  if (i < n)
    for (int k = lane; k < chunks; k += LANES) s += part[(size_t)k * n + i];
  __shared__ double red[LANES][COLS + 1];
  red[lane][col] = s;
  __syncthreads();
  if (lane == 0 && i < n) {
    double t = red[0][col];
    for (int l = 1; l < LANES; ++l) t += red[l][col];
    out[i] = (OUT)t;
  }
}
// 32 columns x 8 lanes per block; with many partial rows and few columns (the per-block partials of the conv epilogue: 4096 rows x 2048
// columns ran on 64 blocks) 8 columns x 32 lanes, four times the blocks and a quarter of the serial adds per thread
template <typename OUT>
static void colred_final_launch(const double* part, OUT* out, size_t n, int chunks, hipStream_t s) {
  if (chunks >= 256 && n <= 16384) hipLaunchKernelGGL((colred_final_kernel<OUT, 8>), dim3(cdiv(n, 8)), dim3(256), 0, s, part, out, n, chunks);
  else hipLaunchKernelGGL((colred_final_kernel<OUT, 32>), dim3(cdiv(n, 32)), dim3(256), 0, s, part, out, n, chunks);
}

static int colred_chunks(size_t rows, int C) {
  const int NQ = C / 4, NQc = NQ < 256 ? NQ : 256, RL = 256 / NQc;
  const int gx = (NQ + NQc - 1) / NQc;
  int chunks = (1024 + gx - 1) / gx;
  const size_t max_chunks = (rows + (size_t)RL * 4 - 1) / ((size_t)RL * 4);
  if ((size_t)chunks > max_chunks) chunks = (int)max_chunks;
  if (chunks < 1) chunks = 1;
  return chunks;
}
size_t colred_workspace_bytes(size_t rows, int C) { return (size_t)colred_chunks(rows, C) * 2 * C * sizeof(double); }

// out_f64 (NV*C doubles) or out_f32 (MODE 0 only) receives the reduced sums
int colred_run(int mode, ColRedArgs a, void* ws, size_t ws_bytes, double* out_f64, float* out_f32, hipStream_t s) {
  if (a.C % 4) { set_error("column reduction: C %d %% 4 != 0", a.C); return GN_EINVAL; }
  if (a.rows == 0) { set_error("column reduction: no rows"); return GN_EINVAL; }
  const int chunks = colred_chunks(a.rows, a.C);
  const int NV = mode == 0 ? 1 : 2;
  if (ws_bytes < (size_t)chunks * NV * a.C * sizeof(double)) { set_error("column reduction: workspace too small"); return GN_EWORKSPACE; }
  a.part = (double*)ws;
  a.rows_per_chunk = (int)((a.rows + chunks - 1) / chunks);
  const int NQ = a.C / 4, NQc = NQ < 256 ? NQ : 256;
  dim3 grid((NQ + NQc - 1) / NQc, chunks);
  if (mode == 0) hipLaunchKernelGGL(colred_kernel<0>, grid, dim3(256), 0, s, a);
  else if (mode == 1) hipLaunchKernelGGL(colred_kernel<1>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(colred_kernel<2>, grid, dim3(256), 0, s, a);
  int rc = check_launch("colred");
  if (rc) return rc;
  const size_t n = (size_t)NV * a.C;
  if (out_f32) colred_final_launch((const double*)ws, out_f32, n, chunks, s);
  else colred_final_launch((const double*)ws, out_f64, n, chunks, s);
  return check_launch("colred_final");
}

int colred_finalize(const double* part, double* out_f64, size_t n, int chunks, hipStream_t s) {
  colred_final_launch(part, out_f64, n, chunks, s);
  return check_launch("colred_final");
}
int colred_finalize_f32(const double* part, float* out_f32, size_t n, int chunks, hipStream_t s) {
  colred_final_launch(part, out_f32, n, chunks, s);
  return check_launch("colred_final");
}

// ---------------------------------------------------------------------------------------------
// BatchNorm finalize / apply / backward-apply
// ---------------------------------------------------------------------------------------------
// moving statistics, two forms of TF's assign_moving_average (fp32 variables, like TF's):
//   zd_step == 0 : plain EMA (zero_debias=False):  v -= (v - value) * (1 - m)
//   zd_step >= 1 : zero_debias=True (keras 2.2.4's TF backend): biased -= (biased - value) * (1 - m);  v -= v - biased / (1 - m^step)
//                  with `biased` a shadow accumulator that starts at zero and zd_step the already incremented local_step
__global__ void bn_finalize_kernel(const double* __restrict__ sums, double count, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float eps, float momentum, float* __restrict__ mm, float* __restrict__ mv, float* __restrict__ bm,
                                   float* __restrict__ bv, float zd_step, float* __restrict__ scale,
                                   float* __restrict__ shift, float* __restrict__ smean, float* __restrict__ sinv, int C, const int32_t* __restrict__ zd_step_dev) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (zd_step_dev) zd_step = (float)*zd_step_dev;
  const double mean = sums[c] / count;
  double var = sums[C + c] / count - mean * mean;
  if (var < 0) var = 0;
  const float meanf = (float)mean, varf = (float)var;
  const float inv = 1.0f / sqrtf(varf + eps);
  const float sc = gamma[c] * inv;
  scale[c] = sc;
  shift[c] = beta[c] - meanf * sc;
  smean[c] = meanf;
  sinv[c] = inv;
  if (mm) {
    const float corr = (float)(count / (count - (1.0 + (double)eps)));
    const float decay = (float)(1.0 - (double)momentum);
    const float varc = varf * corr;
    if (bm) {
      const float nbm = bm[c] - (bm[c] - meanf) * decay;
      const float nbv = bv[c] - (bv[c] - varc) * decay;
      bm[c] = nbm;
      bv[c] = nbv;
      const float unb = 1.0f - powf(1.0f - decay, zd_step);
      mm[c] = mm[c] - (mm[c] - nbm / unb);
      mv[c] = mv[c] - (mv[c] - nbv / unb);
    } else {
      mm[c] = mm[c] - (mm[c] - meanf) * decay;
      mv[c] = mv[c] - (mv[c] - varc) * decay;
    }
  }
}
int bn_finalize(const double* sums, double count, const float* gamma, const float* beta, float eps, float momentum, float* mm, float* mv,
                float* bm, float* bv, float zd_step, float* scale, float* shift, float* smean, float* sinv, int C, hipStream_t s, const int32_t* zd_step_dev) {
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, sums, count, gamma, beta, eps, momentum, mm, mv, bm, bv, zd_step, scale, shift,
                     smean, sinv, C, zd_step_dev);
  return check_launch("bn_finalize");
}

__global__ void bn_infer_coeffs_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mm,
                                       const float* __restrict__ mv, float eps, float* __restrict__ scale, float* __restrict__ shift, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = gamma[c] / sqrtf(mv[c] + eps);
  scale[c] = sc;
  shift[c] = beta[c] - mm[c] * sc;
}
int bn_infer_coeffs(const float* gamma, const float* beta, const float* mm, const float* mv, float eps, float* scale, float* shift, int C, hipStream_t s) {
  hipLaunchKernelGGL(bn_infer_coeffs_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, gamma, beta, mm, mv, eps, scale, shift, C);
  return check_launch("bn_infer_coeffs");
}

__global__ void bn_apply_kernel(const float4* __restrict__ x, const float4* __restrict__ scale, const float4* __restrict__ shift,
                                const uchar4* __restrict__ mask, float4* __restrict__ y, size_t n4, int C4, int act, float p, float keep_scale) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const int c = (int)(i % C4);
    const float4 v = x[i], sc = scale[c], sh = shift[c];
    float4 o;
    o.x = act_apply(fmaf(v.x, sc.x, sh.x), act, p); o.y = act_apply(fmaf(v.y, sc.y, sh.y), act, p);
    o.z = act_apply(fmaf(v.z, sc.z, sh.z), act, p); o.w = act_apply(fmaf(v.w, sc.w, sh.w), act, p);
    if (mask) {
      const uchar4 m = mask[i];
      o.x = m.x ? o.x * keep_scale : 0.f; o.y = m.y ? o.y * keep_scale : 0.f;
      o.z = m.z ? o.z * keep_scale : 0.f; o.w = m.w ? o.w * keep_scale : 0.f;
    }
    y[i] = o;
  }
}
int bn_apply(const float* x, const float* scale, const float* shift, const uint8_t* mask, float* y, size_t rows, int C, int act, float p, float rate, hipStream_t s) {
  if (C % 4) { set_error("bn_apply: C %d %% 4 != 0", C); return GN_EINVAL; }
  const size_t n4 = rows * (C / 4);
  if (!n4) return GN_OK;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(stream_grid(n4)), dim3(256), 0, s, (const float4*)x, (const float4*)scale, (const float4*)shift,
                     (const uchar4*)mask, (float4*)y, n4, C / 4, act, p, 1.0f / (1.0f - rate));
  return check_launch("bn_apply");
}

__global__ void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ x, const uint8_t* __restrict__ mask,
                                    const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ invstd,
                                    const double* __restrict__ dsums, double count, float* __restrict__ dx, size_t n, int C, int act, float p, float keep_scale) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int c = (int)(i % C);
    const float g = bn_bwd_g(dy[i], y[i], mask ? mask[i] : (uint8_t)1, act, p, keep_scale);
    const float inv = invstd[c];
    const float xh = (x[i] - mean[c]) * inv;
    const float mg = (float)(dsums[c] / count), mgx = (float)(dsums[C + c] / count);
    dx[i] = gamma[c] * inv * (g - mg - xh * mgx);
  }
}
// C % 4 == 0: a thread owns ONE group of four channels (its seven per-channel constants stay in registers: no modulo, no fp64
// division per element) and walks the rows of its chunk with 16-byte loads / stores; with scale / shift the activation output
// is recomputed from x instead of read (17 -> 13 bytes per element).  Same arithmetic per element as the scalar kernel.
__global__ __launch_bounds__(256) void bn_bwd_apply_v4_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ x,
                                                              const uint8_t* __restrict__ mask, const float* __restrict__ gamma, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, const double* __restrict__ dsums, double count,
                                                              const float* __restrict__ scale, const float* __restrict__ shift, float* __restrict__ dx,
                                                              size_t rows, int C, int rows_per_chunk, int act, float p, float keep_scale, LazyDy lz) {
  const int NQ = C >> 2;
  const int NQc = NQ < 256 ? NQ : 256;
  const int RL = 256 / NQc;
  const int tid = threadIdx.x, ql = tid % NQc, rl = tid / NQc;
  const int qblocks = (NQ + NQc - 1) / NQc;
  const int q = (blockIdx.x % qblocks) * NQc + ql;
  const size_t r_lo = (size_t)(blockIdx.x / qblocks) * rows_per_chunk;
  const size_t r_hi = r_lo + rows_per_chunk < rows ? r_lo + rows_per_chunk : rows;
  if (rl >= RL || q >= NQ) return;
  float gi[4], mu[4], is[4], mg[4], mgx[4], sc[4] = {0, 0, 0, 0}, sh[4] = {0, 0, 0, 0};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = 4 * q + e;
    is[e] = invstd[c]; mu[e] = mean[c]; gi[e] = gamma[c] * is[e];
    mg[e] = (float)(dsums[c] / count); mgx[e] = (float)(dsums[C + c] / count);
    if (scale) { sc[e] = scale[c]; sh[e] = shift[c]; }
  }
  float wq[5][4];
  LazyWin lw = {0.f, 0, 0xffffffffu};
  unsigned lb = 0;
  int lt = 0;
  if (lz.g) {
    lazy_dy_taps(lz, C, q, wq);
    lazy_dy_pos(lz, r_lo + rl, &lb, &lt);
  }
  constexpr int U = 2;                                   // rows per trip, loads first (see colred_kernel: more is slower)
  const bool uni = ((NQc | NQ) & 63) == 0;
  for (size_t r = r_lo + rl; r < r_hi; r += (size_t)RL * U) {
    float4 d4[U], x4[U], y4[U];
    uchar4 m4[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t ru = r + (size_t)u * RL;
      ok[u] = ru < r_hi;
      const size_t o = (ok[u] ? ru : r) * C + 4 * q;
      x4[u] = *reinterpret_cast<const float4*>(x + o);
      if (!lz.g) d4[u] = *reinterpret_cast<const float4*>(dy + o);
      if (!scale) y4[u] = *reinterpret_cast<const float4*>(y + o);
      m4[u] = mask ? *reinterpret_cast<const uchar4*>(mask + o) : make_uchar4(1, 1, 1, 1);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!ok[u]) break;
      const size_t o = (r + (size_t)u * RL) * C + 4 * q;
      const float xv[4] = {x4[u].x, x4[u].y, x4[u].z, x4[u].w};
      float dv[4];
      if (lz.g) {
        if (uni) lazy_dy4<true>(lz, lb, lt, wq, dv, lw);
        else lazy_dy4<false>(lz, lb, lt, wq, dv, lw);
        lazy_dy_step(lz, RL, &lb, &lt);
      } else {
        dv[0] = d4[u].x; dv[1] = d4[u].y; dv[2] = d4[u].z; dv[3] = d4[u].w;
      }
      float yv[4] = {0, 0, 0, 0};
      if (!scale) { yv[0] = y4[u].x; yv[1] = y4[u].y; yv[2] = y4[u].z; yv[3] = y4[u].w; }
      const uint8_t k[4] = {m4[u].x, m4[u].y, m4[u].z, m4[u].w};
      float ov[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float g = scale ? bn_bwd_g_act(dv[e], act_apply(fmaf(xv[e], sc[e], sh[e]), act, p), k[e], act, p, keep_scale)
                              : bn_bwd_g(dv[e], yv[e], k[e], act, p, keep_scale);
        const float xh = (xv[e] - mu[e]) * is[e];
        ov[e] = gi[e] * (g - mg[e] - xh * mgx[e]);
      }
      *reinterpret_cast<float4*>(dx + o) = make_float4(ov[0], ov[1], ov[2], ov[3]);
    }
  }
}
__global__ void bn_param_grads_kernel(const double* __restrict__ dsums_local, float* __restrict__ dgamma, float* __restrict__ dbeta, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  dbeta[c] = (float)dsums_local[c];
  dgamma[c] = (float)dsums_local[C + c];
}
int bn_bwd_apply(const float* dy, const float* y, const float* x, const uint8_t* mask, const float* gamma, const float* mean, const float* invstd,
                 const double* dsums_global, double count, const double* dsums_local, float* dx, float* dgamma, float* dbeta, size_t rows, int C,
                 int act, float p, float rate, const float* scale, const float* shift, hipStream_t s, const LazyDy* lz) {
  const size_t n = rows * C;
  if (!n) return GN_OK;
  LazyDy z = {};
  if (lz) z = *lz;
  if (z.g && (C % 4 || !scale)) { set_error("bn_bwd_apply: the on-the-fly conv gradient needs C %% 4 == 0 and scale / shift"); return GN_EINVAL; }
  if (C % 4 == 0) {
    const int NQ = C / 4, NQc = NQ < 256 ? NQ : 256, RL = 256 / NQc, qblocks = cdiv(NQ, NQc);
    size_t chunks = 8192 / qblocks;                       // ~8k blocks: 32 per CU
    if (chunks < 1) chunks = 1;
    size_t rpc = (rows + chunks - 1) / chunks;
    rpc = ((rpc + RL - 1) / RL) * RL;
    if (rpc < (size_t)RL) rpc = RL;
    chunks = (rows + rpc - 1) / rpc;
    hipLaunchKernelGGL(bn_bwd_apply_v4_kernel, dim3((unsigned)(chunks * qblocks)), dim3(256), 0, s, dy, y, x, mask, gamma, mean, invstd, dsums_global, count, scale,
                       shift, dx, rows, C, (int)rpc, act, p, 1.0f / (1.0f - rate), z);
  } else {
    if (!y) { set_error("bn_bwd_apply: C %d %% 4 != 0 needs the stored layer output y", C); return GN_EINVAL; }
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(stream_grid(n)), dim3(256), 0, s, dy, y, x, mask, gamma, mean, invstd, dsums_global, count, dx, n, C, act, p,
                       1.0f / (1.0f - rate));
  }
  int rc = check_launch("bn_bwd_apply");
  if (rc) return rc;
  hipLaunchKernelGGL(bn_param_grads_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, dsums_local, dgamma, dbeta, C);
  return check_launch("bn_param_grads");
}

// ---------------------------------------------------------------------------------------------
// losses: single block (B is a batch size, a few thousand at most)
// ---------------------------------------------------------------------------------------------
template <int KIND>  // 0 = BCE, 1 = MSE
__global__ __launch_bounds__(256) void loss_kernel(const float* __restrict__ p, const float* __restrict__ y, float* __restrict__ dp, float* __restrict__ out,
                                                   int B, int Bglobal) {
  const float eps = 1e-7f;
  float lsum = 0.f, hits = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) {
    const float pv = p[i], yv = y[i];
    if (KIND == 0) {
      const float pc = fminf(fmaxf(pv, eps), 1.f - eps);
      const float z = logf(pc / (1.f - pc));
      lsum += fmaxf(z, 0.f) - z * yv + log1pf(expf(-fabsf(z)));
      const bool inside = (pv >= eps) && (pv <= 1.f - eps);
      const float sg = 1.f / (1.f + expf(-z));
      dp[i] = inside ? (sg - yv) / (pc * (1.f - pc)) / (float)Bglobal : 0.f;
    } else {
      const float d = pv - yv;
      lsum += d * d;
      dp[i] = 2.f * d / (float)Bglobal;
    }
    hits += (rintf(pv) == yv) ? 1.f : 0.f;
  }
  __shared__ float r0[256], r1[256];
  r0[threadIdx.x] = lsum; r1[threadIdx.x] = hits;
  __syncthreads();
  for (int sft = 128; sft >= 1; sft >>= 1) {
    if (threadIdx.x < sft) { r0[threadIdx.x] += r0[threadIdx.x + sft]; r1[threadIdx.x] += r1[threadIdx.x + sft]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = r0[0] / (float)Bglobal; out[1] = r1[0]; }
}
int loss_run(int kind, const float* p, const float* y, float* dp, float* out, int B, int Bglobal, hipStream_t s) {
  if (B < 1 || Bglobal < B) { set_error("loss: bad batch sizes %d / %d", B, Bglobal); return GN_EINVAL; }
  if (kind == 0) hipLaunchKernelGGL(loss_kernel<0>, dim3(1), dim3(256), 0, s, p, y, dp, out, B, Bglobal);
  else hipLaunchKernelGGL(loss_kernel<1>, dim3(1), dim3(256), 0, s, p, y, dp, out, B, Bglobal);
  return check_launch("loss");
}

// ---------------------------------------------------------------------------------------------
// Adam (keras form), one fused pass over the flat parameter segment
// ---------------------------------------------------------------------------------------------
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n, float lr_t, float b1,
                            float b2, float eps, const float* __restrict__ lr_t_dev) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  if (lr_t_dev) lr_t = *lr_t_dev;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] -= lr_t * mi / (sqrtf(vi) + eps);
  }
}
int adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr_t, float b1, float b2, float eps, hipStream_t s, const float* lr_t_dev) {
  if (!n) return GN_OK;
  hipLaunchKernelGGL(adam_kernel, dim3(stream_grid(n)), dim3(256), 0, s, p, g, m, v, n, lr_t, b1, b2, eps, lr_t_dev);
  return check_launch("adam");
}

// ---------------------------------------------------------------------------------------------
// weight layout helpers: conv transpose, width-2 Conv2D fold/unfold
// ---------------------------------------------------------------------------------------------
__global__ void transpose_w_kernel(const float* __restrict__ w, float* __restrict__ wt, int k, int Cin, int Cout) {
  __shared__ float tile[32][33];
  const int j = blockIdx.z;
  const int c0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, n = n0 + tx;
    tile[r][tx] = (c < Cin && n < Cout) ? w[((size_t)j * Cin + c) * Cout + n] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int n = n0 + r, c = c0 + tx;
    if (n < Cout && c < Cin) wt[((size_t)j * Cout + n) * Cin + c] = tile[tx][r];
  }
}
int transpose_w(const float* w, float* wt, int k, int Cin, int Cout, hipStream_t s) {
  hipLaunchKernelGGL(transpose_w_kernel, dim3(cdiv(Cout, 32), cdiv(Cin, 32), k), dim3(256), 0, s, w, wt, k, Cin, Cout);
  return check_launch("transpose_w");
}

__global__ void conv2d_w2_fold_kernel(const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ wf, float* __restrict__ bf, int kh, int Cin, int Cout) {
  const size_t total = (size_t)kh * 2 * Cin * 2 * Cout, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int co = (int)(i % (2 * Cout));
    const int ci = (int)((i / (2 * Cout)) % (2 * Cin));
    const int h = (int)(i / ((size_t)4 * Cin * Cout));
    const int wo = co / Cout, c2 = co % Cout, wi = ci / Cin, c = ci % Cin;
    wf[i] = w[(((size_t)h * 5 + (wi - wo + 2)) * Cin + c) * Cout + c2];
  }
  if (bias && blockIdx.x == 0)
    for (int o = threadIdx.x; o < 2 * Cout; o += blockDim.x) bf[o] = bias[o % Cout];
}
__global__ void conv2d_w2_unfold_kernel(const float* __restrict__ dwf, const float* __restrict__ dbf, float* __restrict__ dw, float* __restrict__ db, int kh, int Cin, int Cout) {
  const size_t total = (size_t)kh * 5 * Cin * Cout, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c2 = (int)(i % Cout);
    const int c = (int)((i / Cout) % Cin);
    const int kw = (int)((i / ((size_t)Cin * Cout)) % 5);
    const int h = (int)(i / ((size_t)5 * Cin * Cout));
    float s = 0.f;
    for (int wi = 0; wi < 2; ++wi) {
      const int wo = wi + 2 - kw;
      if (wo < 0 || wo > 1) continue;
      s += dwf[((size_t)h * 2 * Cin + wi * Cin + c) * (2 * Cout) + wo * Cout + c2];
    }
    dw[i] = s;
  }
  if (db && blockIdx.x == 0)
    for (int o = threadIdx.x; o < Cout; o += blockDim.x) db[o] = dbf[o] + dbf[Cout + o];
}
int conv2d_w2_fold(const float* w, const float* bias, float* wf, float* bf, int kh, int Cin, int Cout, hipStream_t s) {
  hipLaunchKernelGGL(conv2d_w2_fold_kernel, dim3(stream_grid((size_t)kh * 4 * Cin * Cout)), dim3(256), 0, s, w, bias, wf, bf, kh, Cin, Cout);
  return check_launch("conv2d_w2_fold");
}
int conv2d_w2_unfold(const float* dwf, const float* dbf, float* dw, float* db, int kh, int Cin, int Cout, hipStream_t s) {
  hipLaunchKernelGGL(conv2d_w2_unfold_kernel, dim3(stream_grid((size_t)kh * 5 * Cin * Cout)), dim3(256), 0, s, dwf, dbf, dw, db, kh, Cin, Cout);
  return check_launch("conv2d_w2_unfold");
}

// UpSampling1D(2) -> Conv1D(k=5, 'same') folded into a 3-tap stride-1 conv on the un-upsampled input:
//   stride 2:  y[t]    = W0 x[t-1] + (W1+W2) x[t] + (W3+W4) x[t+1]                                   wf (3, Cin, Cout)
//   stride 1:  y[2s]   = (W0+W1) x[s-1] + (W2+W3) x[s] + W4 x[s+1]   (columns [0, Cout) of wf)         wf (3, Cin, 2*Cout)
//              y[2s+1] = W0 x[s-1] + (W1+W2) x[s] + (W3+W4) x[s+1]   (columns [Cout, 2*Cout))
// the (Lin, 2*Cout) output of the stride-1 form IS the (2*Lin, Cout) tensor in memory.  tap k of W lands on folded tap UP2_TAB[phase][k].
__device__ __constant__ int UP2_TAB[2][5] = {{0, 0, 1, 1, 2}, {0, 1, 1, 2, 2}};
__global__ void up2_fold_kernel(const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ wf, float* __restrict__ bf, int Cin, int Cout, int stride) {
  const int phases = stride == 1 ? 2 : 1, Cf = phases * Cout;
  const size_t total = (size_t)3 * Cin * Cf, step = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
    const int o = (int)(i % Cf);
    const int c = (int)((i / Cf) % Cin);
    const int j = (int)(i / ((size_t)Cin * Cf));
    const int ph = stride == 1 ? o / Cout : 1, n = o % Cout;
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 5; ++k)
      if (UP2_TAB[ph][k] == j) v += w[((size_t)k * Cin + c) * Cout + n];
    wf[i] = v;
  }
  if (bias && blockIdx.x == 0)
    for (int o = threadIdx.x; o < Cf; o += blockDim.x) bf[o] = bias[o % Cout];
}
__global__ void up2_unfold_kernel(const float* __restrict__ dwf, const float* __restrict__ dbf, float* __restrict__ dw, float* __restrict__ db, int Cin, int Cout, int stride) {
  const int Cf = (stride == 1 ? 2 : 1) * Cout;
  const size_t total = (size_t)5 * Cin * Cout, step = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
    const int n = (int)(i % Cout);
    const int c = (int)((i / Cout) % Cin);
    const int k = (int)(i / ((size_t)Cin * Cout));
    float v = dwf[((size_t)UP2_TAB[1][k] * Cin + c) * Cf + (stride == 1 ? Cout : 0) + n];
    if (stride == 1) v = dwf[((size_t)UP2_TAB[0][k] * Cin + c) * Cf + n] + v;
    dw[i] = v;
  }
  if (db && blockIdx.x == 0)
    for (int o = threadIdx.x; o < Cout; o += blockDim.x) db[o] = stride == 1 ? dbf[o] + dbf[Cout + o] : dbf[o];
}
int up2_fold(const float* w, const float* bias, float* wf, float* bf, int Cin, int Cout, int stride, hipStream_t s) {
  hipLaunchKernelGGL(up2_fold_kernel, dim3(stream_grid((size_t)6 * Cin * Cout)), dim3(256), 0, s, w, bias, wf, bf, Cin, Cout, stride);
  return check_launch("up2_fold");
}
int up2_unfold(const float* dwf, const float* dbf, float* dw, float* db, int Cin, int Cout, int stride, hipStream_t s) {
  hipLaunchKernelGGL(up2_unfold_kernel, dim3(stream_grid((size_t)5 * Cin * Cout)), dim3(256), 0, s, dwf, dbf, dw, db, Cin, Cout, stride);
  return check_launch("up2_unfold");
}

}  // namespace gn
