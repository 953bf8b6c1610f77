// Shared helpers for the gfx950 kernel library (libgennet_hip.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/gennet_hip.h"

// ---------------------------------------------------------------------------------------------------------------------------------------------
// LDS-DMA (buffer_load_dwordx4 ... lds, global_load_lds_dwordx4) CLOBBERS v[0:3] on gfx950.  The instruction's VDATA / VDST field is unused and encoded as
// v0; found in round 5 (profiles/r05_winograd_gate.txt): with the accumulator of point 0 / column tile 0 living in v[2:5], conv_wino_kernel returned, once in a
// few launches, a wrong v[2:3] in one wave -- no instruction of the kernel writes those registers -- as soon as its vector instructions were scheduled in runs
// (the spread schedules never showed it in thousands of launches, so the write depends on timing; an MFMA + VALU + ds_read loop alone does not reproduce it,
// scripts/mfma_valu_hazard.hip).  Declaring v0-v3 clobbered at every LDS-DMA statement (the register allocator then keeps nothing live there) removed it in the
// failing schedule: 160 clean launches against a failure in nearly every 20.  Every LDS-DMA in this library goes through these two macros.
// ---------------------------------------------------------------------------------------------------------------------------------------------
#define GN_LDS_DMA_CLOBBER() asm volatile("" ::: "v0", "v1", "v2", "v3")
#define gn_buffer_load_lds(...) do { __builtin_amdgcn_raw_ptr_buffer_load_lds(__VA_ARGS__); GN_LDS_DMA_CLOBBER(); } while (0)
#define gn_global_load_lds(...) do { __builtin_amdgcn_global_load_lds(__VA_ARGS__); GN_LDS_DMA_CLOBBER(); } while (0)

namespace gn {

void set_error(const char* fmt, ...);
int check_launch(const char* what);

// Event-based timing of the MFMA kernels (bench.py roofline leg).  No-ops unless gn_prof_enable(1).
// Step-varying scalars of a captured hipGraph live in device memory (a by-value kernel argument would be frozen into the graph):
// rng_base() = the device word every Philox kernel adds to its counter offset (NULL outside graph capture; gn_set_rng_base).
const uint64_t* rng_base();
void prof_begin(hipStream_t s);
void prof_end(hipStream_t s, double flop, int kind, double bytes = 0.0);  // kind 0: conv_mfma (fwd, dgrad), 1: wgrad_mfma, 2: bf16x3 conv, 3: fused synthesiser, 4: fused noise chain, 5: transform-domain conv F(2,5) (flop = executed = 0.6 algorithmic), 6: transform-domain wgrad, 7: transform-domain stride-2 conv (0.7), 8: transform-domain stride-2 wgrad (0.7)


// ---------------------------------------------------------------------------------------------
// Generic tap description shared by all convolution kernels:
//   y[b, out_stride*m + out_off, n] = act(bias[n] + sum_j sum_c x[b, in_stride*m + off[j], c] * w[widx[j], c, n])
// covers Conv1D forward (off[j] = j - pad_left), its data gradient (per output phase when stride = 2) and Dense (1 tap).
// ---------------------------------------------------------------------------------------------
struct ConvTaps {
  int ntaps;
  int in_stride;
  int off[8];
  int widx[8];
  int out_stride, out_off;
  // merged two-phase launch (stride-2 data gradient, conv_pipe_try_merged): taps with even index j accumulate the output rows
  // out_stride*m + out_off, taps with odd j the rows out_stride*m + out_off_odd
  int out_off_odd;
};

struct ConvArgs {
  const float* x;
  const float* w;
  const float* bias;
  float* y;
  int B, Lin, Cin, Cout;
  int M;   // output positions m per batch element (this phase)
  int Ly;  // rows per batch element in y
  ConvTaps t;
  int act;
  float act_param;
  const uint8_t* mask;  // optional dropout keep-mask, same shape as y: y = mask ? act(.) * keep_scale : 0 (fused Dropout)
  float keep_scale;
  // gradient epilogue (data-gradient launches): out *= d act(prev)/d pre-activation, expressed through the PRODUCER layer's output
  // gy (same shape as y), optionally through its dropout: out = gmask ? out * gscale * act'(gy/gscale) : 0
  const float* gy;
  const uint8_t* gmask;
  int gact;
  float gparam, gscale;
  // optional BatchNorm statistics of the OUTPUT, accumulated in the epilogue (pipelined kernel, linear epilogue): per-block fp64
  // partials stat_part[(b*m_tiles + m_tile)][2][Cout] = (sum y, sum y^2) over the block's valid rows, reduced in fixed order into
  // stat_sums[2*Cout]; *stat_done is set when the launched kernel produced them (the caller falls back to a separate pass otherwise)
  double* stat_part;
  double* stat_sums;
  int* stat_done;
};

struct WgradArgs {
  const float* x;
  const float* dy;
  float* part;  // [splits][ntaps][Cin][Cout]
  int B, Lin, Cin, Cout, M;
  int ntaps, in_stride;
  int off[8];
  int chunks_per_split;   // K-chunks (32 rows of one batch element) per split: whole batch elements when the batch fills the chip, parts of one below
  double* db_part;  // optional [splits][Cout]: per-split column sums of dy (the bias gradient), written by the blocks of Cin-tile 0
  float* db;        // optional: where wgrad_mfma_dispatch puts the bias gradient when the kernel it selects can sum it on the way
  int db_done;      // set by the dispatcher when db has been written
  // opt-in conv math (gn_set_conv_math 'bf16x3'): the workspace of the split planes; NULL on the default path (wgrad_bf16x3.hip)
  void* split_ws;
  size_t split_ws_bytes;
};

struct WgradSmallArgs {
  const float* x;
  const float* dy;
  float* part;
  int B, Lin, Cin, Cout, M;
  int ntaps, in_stride;
  int off[8];
  int rows_per_chunk;
};

// The gradient arriving at a BatchNormalization output when the layer feeds ONLY a Conv1D(1 filter, k <= 5 taps, stride 1): that conv's
// data gradient dz[b,t,c] = sum_j g[b, t - j + pad_left] * w[j,c], computed where it is consumed instead of written and re-read.
struct LazyDy {
  const float* g;   // (B, Lout): the conv's output gradient (after its own activation backward); NULL = not lazy
  const float* w;   // (k, C): the conv's kernel
  int L, Lout, k, pad_left;
};

struct ColRedArgs {
  const float* a;        // x (MODE 0/1) or dy (MODE 2)
  const float* y;        // MODE 2: layer output (post act, post dropout)
  const float* xpre;     // MODE 2: BN input
  const uint8_t* mask;   // MODE 2: dropout keep mask or NULL
  const float* mean;
  const float* invstd;
  const float* scale;    // MODE 2, optional: the forward pass' gamma*invstd and beta - mean*gamma*invstd: with them the activation output is
  const float* shift;    //   RECOMPUTED from xpre (act(fma(x, scale, shift)), bit-identical to the forward) and y is not read
  double* part;
  size_t rows;
  int C;
  int rows_per_chunk;
  int act;
  float act_param;
  float keep_scale;      // 1/(1-rate)
  LazyDy lz;             // MODE 2: lz.g != NULL -> a is not read
};

// conv_mfma.hip
int conv_mfma_dispatch(const ConvArgs& a, hipStream_t s);
size_t wgrad_workspace_bytes(int B, int M, int Cin, int Cout, int ntaps);
int wgrad_mfma_dispatch(WgradArgs& a, float* dw, size_t ws_bytes, hipStream_t s);
// conv_pipe.hip / wgrad_pipe.hip (hand-scheduled variants selected by the two dispatchers above)
int conv_pipe_try(const ConvArgs& a, bool tall, hipStream_t s, bool* launched);
int conv_pipe_try_merged(const ConvArgs& a, hipStream_t s, bool* launched);
void wgrad_pipe_launch(const WgradArgs& a, dim3 grid, bool narrow, hipStream_t s);
// conv_bf16x3.hip (experimental bf16 x 3 operand-split convolution, opt-in)
size_t conv_bf16x3_workspace_bytes(int B, int Lin, int Cin, int Cout, int w_taps);
bool conv_bf16x3_supported(const ConvArgs& a);
int conv_bf16x3_split(const ConvArgs& a, int w_taps, void* ws, size_t ws_bytes, bool split_x, bool split_w, hipStream_t s);
int conv_bf16x3_run(const ConvArgs& a, int w_taps, void* ws, hipStream_t s);
int conv_bf16x3_merged_kind(const ConvArgs& a);            // 0: not the merged two-phase shape
int conv_bf16x3_run_merged(const ConvArgs& a, void* ws, hipStream_t s);
// conv_wino.hip (transform-domain F(2,5) fp32 convolution for the unit-stride 5-tap launches)
size_t conv_wino_workspace_bytes(int Cin, int Cout);
bool conv_wino_supported(const ConvArgs& a);
int conv_wino_run(const ConvArgs& a, void* ws, size_t ws_bytes, hipStream_t s);
// conv_wino_s2.hip (the stride-2 5-tap layers: F(2,3) + F(2,2) on the even / odd rows; forward and the merged two-phase data gradient)
size_t conv_wino_s2_workspace_bytes(int Cin, int Cout);
int conv_wino_s2_kind(const ConvArgs& a);                 // 1 forward, 2 merged data gradient, 0 not supported
int conv_wino_s2_run(const ConvArgs& a, void* ws, size_t ws_bytes, hipStream_t s);
// wgrad_wino.hip (the transposed form for the weight gradient of the same layers)
void wgrad_split_plan(int B, int M, int Cin, int Cout, int TC, int TN, int* splits, int* chunks_per_split);      // conv_mfma.hip: K-chunks of 32 rows
bool wgrad_wino_supported(const WgradArgs& a);
size_t wgrad_wino_workspace_bytes(int B, int M, int Cin, int Cout);
int wgrad_wino_run(WgradArgs& a, float* dw, size_t ws_bytes, hipStream_t s);
// wgrad_wino_s2.hip (... of the stride-2 layers: transposed F(2,3) + F(2,2))
bool wgrad_wino_s2_supported(const WgradArgs& a);
size_t wgrad_wino_s2_workspace_bytes(int B, int M, int Cin, int Cout);
int wgrad_wino_s2_run(WgradArgs& a, float* dw, size_t ws_bytes, hipStream_t s);
// wgrad_bf16x3.hip (the same split for the weight gradient, opt-in)
size_t wgrad_bf16x3_workspace_bytes(int B, int M, int Cin, int Cout, int in_stride);
bool wgrad_bf16x3_supported(const WgradArgs& a);
int wgrad_bf16x3_run(const WgradArgs& a, int splits, void* ws, size_t ws_bytes, hipStream_t s);
// small_conv.hip
int conv_smallcin_dispatch(const ConvArgs& a, hipStream_t s);
int conv_smallcout_dispatch(const ConvArgs& a, hipStream_t s);
size_t wgrad_small_workspace_bytes(int B, int M, int Cin, int Cout, int ntaps);
int wgrad_small_dispatch(WgradSmallArgs a, float* dw, size_t ws_bytes, hipStream_t s);
int dense_small_fwd(const float* x, const float* w, const float* bias, float* y, int B, int in, int out, int act, float p, hipStream_t s);
int dense_small_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int B, int in, int out, hipStream_t s,
                    int gact = 0, float gparam = 0.f, const uint8_t* gmask = nullptr, float gscale = 1.f);
// elementwise.hip
int act_fwd(const float* x, float* y, size_t n, int act, float p, hipStream_t s);
int act_bwd(const float* dy, const float* y, float* dx, size_t n, int act, float p, hipStream_t s);
int act_dropout_bwd(const float* dy, const float* y, const uint8_t* mask, float* dx, size_t n, int act, float p, float rate, hipStream_t s);
int dropout_mask(uint8_t* mask, size_t n, float rate, uint64_t seed, uint64_t offset, hipStream_t s);
int dropout_apply(const float* x, const uint8_t* mask, float* y, size_t n, float rate, hipStream_t s);
int upsample2_fwd(const float* x, float* y, int B, int L, int C, hipStream_t s);
int upsample2_bwd(const float* dy, float* dx, int B, int L, int C, hipStream_t s);
int subtract_stack_fwd(const float* x, const float* ev, float* img, int B, int n, hipStream_t s);
int subtract_stack_bwd(const float* dimg, float* dx, int B, int n, hipStream_t s);
int affine_stack_fwd(const float* x, const float* b0, const float* b1, float a0, float a1, float* img, int B, int n, hipStream_t s);
int affine_stack_bwd(const float* dimg, float a0, float a1, float* dx, int B, int n, hipStream_t s);
int assemble_d_batch(const float* real, const float* noise, const float* fake, const float* ev, float* sX, int B, int n, hipStream_t s);
int gather_rows(const float* src, const int64_t* idx, float* out, int rows, int width, hipStream_t s);
int axpy(float* y, const float* x, float a, size_t n, hipStream_t s);
int fill_uniform(float* out, size_t n, float lo, float hi, uint64_t seed, uint64_t offset, hipStream_t s);
int fill_normal(float* out, size_t n, float mean, float sd, uint64_t seed, uint64_t offset, hipStream_t s, const float* sd_dev = nullptr);
size_t colred_workspace_bytes(size_t rows, int C);
int colred_run(int mode, ColRedArgs a, void* ws, size_t ws_bytes, double* out_f64, float* out_f32, hipStream_t s);
int colred_finalize(const double* part, double* out_f64, size_t n, int chunks, hipStream_t s);   // out[i] = sum_k part[k*n + i], fixed order
int colred_finalize_f32(const double* part, float* out_f32, size_t n, int chunks, hipStream_t s);
int bn_finalize(const double* sums, double count, const float* gamma, const float* beta, float eps, float momentum, float* mm, float* mv,
                float* bm, float* bv, float zd_step, float* scale, float* shift, float* smean, float* sinv, int C, hipStream_t s,
                const int32_t* zd_step_dev = nullptr);
int bn_infer_coeffs(const float* gamma, const float* beta, const float* mm, const float* mv, float eps, float* scale, float* shift, int C, hipStream_t s);
int bn_apply(const float* x, const float* scale, const float* shift, const uint8_t* mask, float* y, size_t rows, int C, int act, float p, float rate, hipStream_t s);
int bn_bwd_apply(const float* dy, const float* y, const float* x, const uint8_t* mask, const float* gamma, const float* mean, const float* invstd,
                 const double* dsums_global, double count, const double* dsums_local, float* dx, float* dgamma, float* dbeta, size_t rows, int C,
                 int act, float p, float rate, const float* scale, const float* shift, hipStream_t s, const LazyDy* lz = nullptr);
int loss_run(int kind, const float* p, const float* y, float* dp, float* out, int B, int Bglobal, hipStream_t s);
int adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr_t, float b1, float b2, float eps, hipStream_t s, const float* lr_t_dev = nullptr);
int transpose_w(const float* w, float* wt, int k, int Cin, int Cout, hipStream_t s);
int conv2d_w2_fold(const float* w, const float* bias, float* wf, float* bf, int kh, int Cin, int Cout, hipStream_t s);
int conv2d_w2_unfold(const float* dwf, const float* dbf, float* dw, float* db, int kh, int Cin, int Cout, hipStream_t s);
int up2_fold(const float* w, const float* bias, float* wf, float* bf, int Cin, int Cout, int stride, hipStream_t s);
int up2_unfold(const float* dwf, const float* dbf, float* dw, float* db, int Cin, int Cout, int stride, hipStream_t s);

static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// Opt a kernel in to up to 160 KiB of dynamic LDS, once per DEVICE (the attribute is per device and function; `done` is the
// call site's bit mask of devices already served, so a process that drives several GPUs stays correct).
static inline void allow_big_lds(const void* kernel, unsigned long long* done) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(*done & bit)) {
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    *done |= bit;
  }
}

#define GN_REQUIRE(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      gn::set_error(__VA_ARGS__);        \
      return GN_EINVAL;                  \
    }                                    \
  } while (0)

// ---------------------------------------------------------------------------------------------
// activation epilogues (fwd on the pre-activation, bwd through the OUTPUT value)
// ---------------------------------------------------------------------------------------------
// tanh for every kernel of the library (forward epilogues, BN apply passes, and the backward passes that RECOMPUTE the activation from
// the pre-BN tensor: all must agree bit for bit, so there is exactly one implementation).  |x| < 0.35: odd Taylor polynomial through
// x^11 (next term 4e-9); else 1 - 2 / (exp(2|x|) + 1) on v_exp_f32 / v_rcp_f32.  Absolute error <= ~1e-7 (a few ulp at 0.35, less
// elsewhere), NaN in -> NaN out, +-inf -> +-1.  libdevice's tanhf is ~4x the instructions and made the BN backward passes
// arithmetic-bound (0.5-1.0 ms of 4.4-5.6 ms on the generator's largest layer).
__device__ __forceinline__ float gn_tanhf(float x) {
  const float ax = fabsf(x);
  const float e = __builtin_amdgcn_exp2f(ax * 2.8853900817779268f);          // exp(2|x|); +inf past |x| ~ 44 -> big = 1
  const float big = 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
  const float x2 = ax * ax;
  float q = fmaf(x2, -1382.f / 155925.f, 62.f / 2835.f);
  q = fmaf(x2, q, -17.f / 315.f);
  q = fmaf(x2, q, 2.f / 15.f);
  q = fmaf(x2, q, -1.f / 3.f);
  const float small = fmaf(ax * x2, q, ax);
  return copysignf(ax < 0.35f ? small : big, x);
}

__device__ __forceinline__ float act_apply(float x, int act, float p) {
  switch (act) {
    case GN_ACT_RELU: return fmaxf(x, 0.f);
    case GN_ACT_RELU_MAX: return fminf(fmaxf(x, 0.f), p);
    case GN_ACT_LEAKY: return x > 0.f ? x : p * x;
    case GN_ACT_TANH: return gn_tanhf(x);
    case GN_ACT_SIGMOID: return 1.f / (1.f + expf(-x));
    default: return x;
  }
}

__device__ __forceinline__ float act_grad_from_y(float y, int act, float p) {
  switch (act) {
    case GN_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case GN_ACT_RELU_MAX: return (y > 0.f && y < p) ? 1.f : 0.f;
    case GN_ACT_LEAKY: return y > 0.f ? 1.f : p;
    case GN_ACT_TANH: return 1.f - y * y;
    case GN_ACT_SIGMOID: return y * (1.f - y);
    default: return 1.f;
  }
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011), counter-based: stateless, so every rank / kernel derives
// its stream from (seed, offset) alone.
// ---------------------------------------------------------------------------------------------
struct Philox4 {
  uint32_t v[4];
};

__host__ __device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }

__host__ __device__ __forceinline__ Philox4 philox4x32_10(uint64_t counter, uint64_t seed) {
  uint32_t c0 = (uint32_t)counter, c1 = (uint32_t)(counter >> 32), c2 = 0, c3 = 0;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    uint32_t h0 = mulhi32(M0, c0), l0 = M0 * c0;
    uint32_t h1 = mulhi32(M1, c2), l1 = M1 * c2;
    uint32_t n0 = h1 ^ c1 ^ k0, n1 = l1, n2 = h0 ^ c3 ^ k1, n3 = l0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  Philox4 o;
  o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
  return o;
}

// uniform in [0,1) with 24 random bits (exactly representable in fp32)
__host__ __device__ __forceinline__ float u01_24(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }
// uniform in (0,1) with 53-ish bits for fp64 Box-Muller
__host__ __device__ __forceinline__ double u01_53(uint32_t hi, uint32_t lo) {
  uint64_t x = (((uint64_t)hi << 32) | lo) >> 11;
  return ((double)x + 0.5) * (1.0 / 9007199254740992.0);
}

}  // namespace gn
