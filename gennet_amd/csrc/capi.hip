// extern "C" surface of libgennet_hip.so (see include/gennet_hip.h).  Argument checking, tap-table construction and
// kernel-family dispatch live here; no torch types, no allocation, no synchronisation.
#include <stdarg.h>
#include <algorithm>
#include <mutex>
#include <vector>
#include <stdlib.h>
#include "common.h"

namespace gn {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return GN_ELAUNCH;
  }
  return GN_OK;
}

// ---- profiling: HIP events around the MFMA launches, on the stream they are launched on ------------------
struct ProfRec {
  hipEvent_t a, b;
  double flop;
  int kind;
  double bytes;
};
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static std::vector<hipEvent_t> g_pool;
static hipEvent_t g_cur;
static std::mutex g_prof_mu;

static hipEvent_t get_event() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}

static thread_local const uint64_t* g_rng_base = nullptr;
const uint64_t* rng_base() { return g_rng_base; }

void prof_begin(hipStream_t s) {
  if (!g_prof_on) return;
  g_cur = get_event();
  (void)hipEventRecord(g_cur, s);
}

void prof_end(hipStream_t s, double flop, int kind, double bytes) {
  if (!g_prof_on) return;
  hipEvent_t b = get_event();
  (void)hipEventRecord(b, s);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof.push_back({g_cur, b, flop, kind, bytes});
}

static void fwd_taps(ConvTaps* t, int k, int stride, int pad_left) {
  t->ntaps = k;
  t->in_stride = stride;
  for (int j = 0; j < k; ++j) {
    t->off[j] = j - pad_left;
    t->widx[j] = j;
  }
  t->out_stride = 1;
  t->out_off = 0;
}

// Conv arithmetic (gn_set_conv_math): 0 = direct fp32 MFMA kernels only; 1 = opt-in bf16 x 3 operand split for the launches it supports and that
// are large enough to gain from it; 2 = transform-domain fp32 (Cook-Toom F(2,5), conv_wino.hip) for every unit-stride 5-tap launch it supports --
// decided by the layer's shape alone, never by the batch size, so a batch and its chunks take the same kernel and agree bit for bit.  Everything
// else stays on the direct fp32 kernels.
static int g_conv_math = 0;
static void* g_conv_ws = nullptr;
static size_t g_conv_ws_bytes = 0;

// The output phases of one strided data gradient read the same dy and the same kernel: under the opt-in split math the first phase splits them (ALL k
// taps of the kernel, so that the planes' layout does not depend on the phase), the later phases reuse the planes (8 of the 28 split passes of a
// BASELINE step are such repeats).  Set by dgrad_impl around its phase loop.
static int g_phase_w_taps = 0;
static bool g_phase_have_split = false;
struct PhaseScope {
  explicit PhaseScope(int w_taps) { g_phase_w_taps = w_taps; g_phase_have_split = false; }
  ~PhaseScope() { g_phase_w_taps = 0; g_phase_have_split = false; }
};

// Size gate of the opt-in split: below ~50 GFLOP a launch does not keep the 256-row blocks of the split kernels busy for more than a round or two and
// the split pass in front of it is pure latency (at the reference script's own batch 8 every launch is below it: the opt-in then changes nothing there
// instead of costing 1 %).  GN_BF16X3_MIN_GFLOP: A/B switch.
static bool split_worth_it(int B, int M, int ntaps, int Cin, int Cout) {
  static const double min_gflop = getenv("GN_BF16X3_MIN_GFLOP") ? atof(getenv("GN_BF16X3_MIN_GFLOP")) : 50.0;
  return 2.0 * B * (double)M * ntaps * Cin * Cout >= min_gflop * 1e9;
}

static int conv_dispatch(const ConvArgs& a, hipStream_t s) {
  if (a.Cin <= 4) return conv_smallcin_dispatch(a, s);
  if (a.Cout <= 4) return conv_smallcout_dispatch(a, s);
  if (g_conv_math == 2 && a.Cin >= 32 && conv_wino_supported(a) && conv_wino_workspace_bytes(a.Cin, a.Cout) <= g_conv_ws_bytes)
    return conv_wino_run(a, g_conv_ws, g_conv_ws_bytes, s);
  if (g_conv_math == 2 && conv_wino_s2_kind(a) == 1 && conv_wino_s2_workspace_bytes(a.Cin, a.Cout) <= g_conv_ws_bytes)
    return conv_wino_s2_run(a, g_conv_ws, g_conv_ws_bytes, s);                 // stride-2 forward: F(2,3) + F(2,2)
  // size threshold of the opt-in split: the split pass costs ~10 bytes per input element per launch, the conv gains ~0.02 ps per element and output
  // channel, so small-Cout layers gain little and small-Cin layers (few K chunks) lose to the prologue
  constexpr int min_cin = 256, min_cout = 256;
  if (g_conv_math == 1 && a.Cin >= min_cin && a.Cout >= min_cout && conv_bf16x3_supported(a) &&
      (g_phase_w_taps > 0 ? split_worth_it(a.B, (a.Ly + 1) / 2, g_phase_w_taps, a.Cin, a.Cout) :      // (every phase of one data gradient decides alike)
                            split_worth_it(a.B, a.M, a.t.ntaps, a.Cin, a.Cout))) {
    int w_taps = 0;
    for (int j = 0; j < a.t.ntaps; ++j) w_taps = std::max(w_taps, a.t.widx[j] + 1);
    const bool phased = g_phase_w_taps > 0;
    if (phased) w_taps = g_phase_w_taps;
    // which kernel a launch takes depends on its shape alone: a workspace too small for it is an error, not a silent change of arithmetic
    if (conv_bf16x3_workspace_bytes(a.B, a.Lin, a.Cin, a.Cout, w_taps) > g_conv_ws_bytes) {
      set_error("conv (bf16x3 math): the split operands of this launch need %zu bytes, the workspace has %zu -- raise GENNET_CONV_WS_GB / ops.set_conv_math(workspace_gb=)",
                conv_bf16x3_workspace_bytes(a.B, a.Lin, a.Cin, a.Cout, w_taps), g_conv_ws_bytes);
      return GN_EWORKSPACE;
    }
    if (!(phased && g_phase_have_split)) {
      int rc = conv_bf16x3_split(a, w_taps, g_conv_ws, g_conv_ws_bytes, true, true, s);
      if (rc) return rc;
      g_phase_have_split = phased;
    }
    return conv_bf16x3_run(a, w_taps, g_conv_ws, s);
  }
  return conv_mfma_dispatch(a, s);
}

static int set_conv_math_impl(int mode, void* workspace, size_t workspace_bytes) {
  GN_REQUIRE(mode >= 0 && mode <= 2, "set_conv_math: mode %d (0 = direct fp32, 1 = bf16x3, 2 = transform-domain fp32)", mode);
  GN_REQUIRE(mode == 0 || workspace, "set_conv_math: modes 1 and 2 need a device workspace");
  g_conv_math = mode;
  g_conv_ws = mode ? workspace : nullptr;
  g_conv_ws_bytes = mode ? workspace_bytes : 0;
  return GN_OK;
}

// small C (< 4 or not a multiple of 4) column sums: fp64 block partials + fp64 atomics
__global__ void colsum_anyc_kernel(const float* __restrict__ x, double* __restrict__ acc, size_t n, int C) {
  double s[4] = {0, 0, 0, 0};
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int c = (int)(i % C);
    const double v = (double)x[i];
    s[0] += c == 0 ? v : 0.0; s[1] += c == 1 ? v : 0.0; s[2] += c == 2 ? v : 0.0; s[3] += c == 3 ? v : 0.0;
  }
  __shared__ double red[4][256];
  for (int c = 0; c < 4; ++c) red[c][threadIdx.x] = s[c];
  __syncthreads();
  for (int sft = 128; sft >= 1; sft >>= 1) {
    if (threadIdx.x < sft)
      for (int c = 0; c < 4; ++c) red[c][threadIdx.x] += red[c][threadIdx.x + sft];
    __syncthreads();
  }
  if (threadIdx.x < C) atomicAdd(&acc[threadIdx.x], red[threadIdx.x][0]);
}
__global__ void f64_to_f32_small_kernel(const double* __restrict__ a, float* __restrict__ o, int n) {
  if ((int)threadIdx.x < n) o[threadIdx.x] = (float)a[threadIdx.x];
}

// db[c] = sum over rows of dy[row, c]; ws needs colred_workspace_bytes(rows, C) (C % 4 == 0) or 32 bytes otherwise
static int bias_grad(const float* dy, float* db, size_t rows, int C, void* ws, size_t ws_bytes, hipStream_t s) {
  if (C % 4 == 0) {
    ColRedArgs r = {};
    r.a = dy; r.rows = rows; r.C = C;
    return colred_run(0, r, ws, ws_bytes, nullptr, db, s);
  }
  if (C > 4) { set_error("bias_grad: C %d unsupported", C); return GN_EINVAL; }
  if (ws_bytes < 32) { set_error("bias_grad: workspace too small"); return GN_EWORKSPACE; }
  (void)hipMemsetAsync(ws, 0, 32, s);
  size_t g = (rows * C + 255) / 256;
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(colsum_anyc_kernel, dim3((unsigned)g), dim3(256), 0, s, dy, (double*)ws, rows * C, C);
  hipLaunchKernelGGL(f64_to_f32_small_kernel, dim3(1), dim3(64), 0, s, (const double*)ws, db, C);
  return check_launch("bias_grad");
}

static size_t bias_grad_ws(size_t rows, int C) { return C % 4 == 0 ? colred_workspace_bytes(rows, C) : 32; }

}  // namespace gn

using namespace gn;

extern "C" {

const char* gn_last_error(void) { return g_err; }
int gn_version(void) { return 100; }

int gn_prof_enable(int on) {
  g_prof_on = on != 0;
  return GN_OK;
}
int gn_prof_reset(void) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& r : g_prof) {
    g_pool.push_back(r.a);
    g_pool.push_back(r.b);
  }
  g_prof.clear();
  return GN_OK;
}
int gn_prof_collect(int kind, double* out) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  double ms = 0, flop = 0, cnt = 0, bytes = 0;
  for (auto& r : g_prof) {
    if (kind >= 0 && r.kind != kind) continue;
    cnt += 1;
    if (hipEventSynchronize(r.b) != hipSuccess) { set_error("prof: event sync failed"); return GN_ELAUNCH; }
    float t = 0;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) { set_error("prof: elapsed failed"); return GN_ELAUNCH; }
    ms += t;
    flop += r.flop;
    bytes += r.bytes;
  }
  out[0] = cnt;
  out[1] = ms;
  out[2] = flop;
  out[3] = bytes;
  return GN_OK;
}

// ---------------------------------------------------------------------------------------------------------
int gn_conv1d_fwd(const float* x, const float* w, const float* bias, float* y, int B, int L, int Cin, int Cout, int k, int stride, int pad_left, int Lout,
                  int act, float act_param, void* stream) {
  GN_REQUIRE(x && w && y, "conv1d_fwd: null pointer");
  GN_REQUIRE(B >= 0 && L > 0 && Cin > 0 && Cout > 0 && k >= 1 && k <= 8 && stride >= 1 && Lout > 0, "conv1d_fwd: bad shape");
  GN_REQUIRE(pad_left >= 0 && stride * (Lout - 1) + k - pad_left <= L + k, "conv1d_fwd: Lout %d inconsistent with L %d k %d stride %d", Lout, L, k, stride);
  if (B == 0) return GN_OK;
  ConvArgs a = {};
  a.x = x; a.w = w; a.bias = bias; a.y = y;
  a.B = B; a.Lin = L; a.Cin = Cin; a.Cout = Cout; a.M = Lout; a.Ly = Lout;
  fwd_taps(&a.t, k, stride, pad_left);
  a.act = act; a.act_param = act_param;
  return conv_dispatch(a, (hipStream_t)stream);
}

int gn_set_conv_math(int mode, void* workspace, size_t workspace_bytes) { return set_conv_math_impl(mode, workspace, workspace_bytes); }

size_t gn_conv1d_bf16x3_workspace(int B, int L, int Cin, int Cout, int k) { return conv_bf16x3_workspace_bytes(B, L, Cin, Cout, k); }

size_t gn_conv1d_fwd_stats_workspace(int B, int Lout, int Cout) {
  // per-block partials: one row of 2 * Cout doubles per (batch element, row tile); the smallest row tile any launch_conv_pipe instantiation uses
  // is 64 rows (the narrow-wave blocks with one wave in M, conv_pipe.hip conv_pipe_try: nwm == 1)
  const size_t fused = (size_t)B * (size_t)((Lout + 63) / 64) * 2 * (size_t)Cout * sizeof(double);
  const size_t plain = colred_workspace_bytes((size_t)B * Lout, Cout);
  return (fused > plain ? fused : plain) + 256;
}

int gn_conv1d_fwd_stats(const float* x, const float* w, const float* bias, float* y, double* sums, void* ws, size_t ws_bytes, int B, int L, int Cin, int Cout, int k,
                        int stride, int pad_left, int Lout, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  GN_REQUIRE(x && w && y && sums && ws, "conv1d_fwd_stats: null pointer");
  GN_REQUIRE(B > 0 && L > 0 && Cin > 0 && Cout > 0 && Cout % 4 == 0 && k >= 1 && k <= 8 && stride >= 1 && Lout > 0, "conv1d_fwd_stats: bad shape");
  GN_REQUIRE(pad_left >= 0 && stride * (Lout - 1) + k - pad_left <= L + k, "conv1d_fwd_stats: Lout %d inconsistent with L %d k %d stride %d", Lout, L, k, stride);
  GN_REQUIRE(ws_bytes >= gn_conv1d_fwd_stats_workspace(B, Lout, Cout), "conv1d_fwd_stats: workspace too small");
  ConvArgs a = {};
  a.x = x; a.w = w; a.bias = bias; a.y = y;
  a.B = B; a.Lin = L; a.Cin = Cin; a.Cout = Cout; a.M = Lout; a.Ly = Lout;
  fwd_taps(&a.t, k, stride, pad_left);
  a.act = GN_ACT_LINEAR; a.act_param = 0.f;
  int done = 0;
  a.stat_part = (double*)ws; a.stat_sums = sums; a.stat_done = &done;
  int rc = conv_dispatch(a, s);
  if (rc || done) return rc;
  ColRedArgs r = {};                                     // the launched kernel had no statistics epilogue: one separate pass over y
  r.a = y; r.rows = (size_t)B * Lout; r.C = Cout;
  return colred_run(1, r, ws, ws_bytes, sums, nullptr, s);
}

int gn_conv1d_fwd_bf16x3(const float* x, const float* w, const float* bias, float* y, void* ws, size_t ws_bytes, int B, int L, int Cin, int Cout, int k, int stride,
                         int pad_left, int Lout, int act, float act_param, int resplit, void* stream) {
  GN_REQUIRE(x && w && y && ws, "conv1d_fwd_bf16x3: null pointer");
  GN_REQUIRE(B >= 0 && L > 0 && Cin > 0 && Cout > 0 && k >= 1 && k <= 5 && stride >= 1 && Lout > 0 && pad_left >= 0, "conv1d_fwd_bf16x3: bad shape");
  if (B == 0) return GN_OK;
  ConvArgs a = {};
  a.x = x; a.w = w; a.bias = bias; a.y = y;
  a.B = B; a.Lin = L; a.Cin = Cin; a.Cout = Cout; a.M = Lout; a.Ly = Lout;
  fwd_taps(&a.t, k, stride, pad_left);
  a.act = act; a.act_param = act_param;
  GN_REQUIRE(conv_bf16x3_supported(a), "conv1d_fwd_bf16x3: needs Cin %% 16 == 0, Cout %% 64 == 0, k <= 5, stride 1");
  if (resplit) {
    int rc = conv_bf16x3_split(a, k, ws, ws_bytes, true, true, (hipStream_t)stream);
    if (rc) return rc;
  }
  return conv_bf16x3_run(a, k, ws, (hipStream_t)stream);
}

size_t gn_conv1d_wino_workspace(int Cin, int Cout) { return conv_wino_workspace_bytes(Cin, Cout); }

int gn_conv1d_fwd_wino(const float* x, const float* w, const float* bias, float* y, void* ws, size_t ws_bytes, int B, int L, int Cin, int Cout, int k, int stride,
                       int pad_left, int Lout, int act, float act_param, void* stream) {
  GN_REQUIRE(x && w && y && ws, "conv1d_fwd_wino: null pointer");
  GN_REQUIRE(B >= 0 && L > 0 && Cin > 0 && Cout > 0 && k == 5 && stride == 1 && Lout > 0 && pad_left >= 0, "conv1d_fwd_wino: 5 taps, unit stride");
  if (B == 0) return GN_OK;
  ConvArgs a = {};
  a.x = x; a.w = w; a.bias = bias; a.y = y;
  a.B = B; a.Lin = L; a.Cin = Cin; a.Cout = Cout; a.M = Lout; a.Ly = Lout;
  fwd_taps(&a.t, k, stride, pad_left);
  a.act = act; a.act_param = act_param;
  return conv_wino_run(a, ws, ws_bytes, (hipStream_t)stream);
}

int gn_conv1d_fwd_dropout(const float* x, const float* w, const float* bias, const uint8_t* mask, float* y, int B, int L, int Cin, int Cout, int k, int stride,
                          int pad_left, int Lout, int act, float act_param, float rate, void* stream) {
  GN_REQUIRE(x && w && y && mask, "conv1d_fwd_dropout: null pointer");
  GN_REQUIRE(B >= 0 && L > 0 && Cin > 0 && Cout > 4 && Cout % 4 == 0 && k >= 1 && k <= 5 && stride >= 1 && Lout > 0 && pad_left >= 0, "conv1d_fwd_dropout: bad shape");
  GN_REQUIRE(rate >= 0.f && rate < 1.f, "conv1d_fwd_dropout: bad rate %f", rate);
  if (B == 0) return GN_OK;
  ConvArgs a = {};
  a.x = x; a.w = w; a.bias = bias; a.y = y;
  a.B = B; a.Lin = L; a.Cin = Cin; a.Cout = Cout; a.M = Lout; a.Ly = Lout;
  fwd_taps(&a.t, k, stride, pad_left);
  a.act = act; a.act_param = act_param;
  a.mask = mask; a.keep_scale = 1.0f / (1.0f - rate);
  return conv_dispatch(a, (hipStream_t)stream);
}

int gn_conv1d_transpose_w(const float* w, float* wt, int k, int Cin, int Cout, void* stream) {
  GN_REQUIRE(w && wt && k >= 1 && Cin > 0 && Cout > 0, "transpose_w: bad arguments");
  return transpose_w(w, wt, k, Cin, Cout, (hipStream_t)stream);
}

static int dgrad_impl(const float* dy, const float* wt, float* dx, int B, int L, int Cin, int Cout, int k, int stride, int pad_left, int Lout, const float* gy,
                      const uint8_t* gmask, int gact, float gparam, float grate, void* stream);

int gn_conv1d_dgrad(const float* dy, const float* wt, float* dx, int B, int L, int Cin, int Cout, int k, int stride, int pad_left, int Lout, void* stream) {
  return dgrad_impl(dy, wt, dx, B, L, Cin, Cout, k, stride, pad_left, Lout, nullptr, nullptr, GN_ACT_LINEAR, 0.f, 0.f, stream);
}

int gn_conv1d_dgrad_fused(const float* dy, const float* wt, float* dx, int B, int L, int Cin, int Cout, int k, int stride, int pad_left, int Lout,
                          const float* y_prev, const uint8_t* mask_prev, int act_prev, float act_param_prev, float rate_prev, void* stream) {
  GN_REQUIRE(y_prev, "conv1d_dgrad_fused: y_prev is NULL");
  GN_REQUIRE(Cin > 4 && Cout > 4, "conv1d_dgrad_fused: only the MFMA path fuses the producer's activation gradient (Cin %d, Cout %d)", Cin, Cout);
  GN_REQUIRE(rate_prev >= 0.f && rate_prev < 1.f, "conv1d_dgrad_fused: bad rate");
  return dgrad_impl(dy, wt, dx, B, L, Cin, Cout, k, stride, pad_left, Lout, y_prev, mask_prev, act_prev, act_param_prev, mask_prev ? rate_prev : 0.f, stream);
}

static int dgrad_impl(const float* dy, const float* wt, float* dx, int B, int L, int Cin, int Cout, int k, int stride, int pad_left, int Lout, const float* gy,
                      const uint8_t* gmask, int gact, float gparam, float grate, void* stream) {
  GN_REQUIRE(dy && wt && dx, "conv1d_dgrad: null pointer");
  GN_REQUIRE(B >= 0 && L > 0 && Cin > 0 && Cout > 0 && k >= 1 && k <= 8 && stride >= 1 && Lout > 0 && pad_left >= 0, "conv1d_dgrad: bad shape");
  if (B == 0) return GN_OK;
  // dx[b, tau, ci] = sum_{k', co} dy[b, t, co] * wt[k', co, ci]  with  stride*t + k' - pad_left == tau.
  // Output phase p = tau mod stride uses the taps with (p + pad_left - k') divisible by stride, at dy row m + (p+pad_left-k')/stride.
  if (stride == 2 && k == 5 && L >= 2 && Cin > 4 && Cout > 4) {
    // both output phases in one launch where the pipelined kernel takes it (conv_pipe_try_merged): tap kk belongs to phase (kk + pad_left) & 1
    ConvArgs a = {};
    a.x = dy; a.w = wt; a.bias = nullptr; a.y = dx;
    a.B = B; a.Lin = Lout; a.Cin = Cout; a.Cout = Cin;
    a.M = (L + 1) / 2; a.Ly = L;
    a.t.in_stride = 1; a.t.out_stride = 2; a.t.ntaps = 5;
    for (int kk = 0; kk < 5; ++kk) {
      const int p = (kk + pad_left) & 1;
      const int d = p + pad_left - kk;                       // even by construction
      a.t.off[kk] = (d >= 0) ? d / 2 : -((-d) / 2);
      a.t.widx[kk] = kk;
    }
    a.t.out_off = pad_left & 1;                              // phase of the even taps (kk = 0, 2, 4)
    a.t.out_off_odd = 1 - (pad_left & 1);
    a.act = GN_ACT_LINEAR;
    a.gy = gy; a.gmask = gmask; a.gact = gact; a.gparam = gparam; a.gscale = 1.0f / (1.0f - grate);
    if (g_conv_math == 2 && conv_wino_s2_kind(a) == 2 && conv_wino_s2_workspace_bytes(a.Cin, a.Cout) <= g_conv_ws_bytes)
      return conv_wino_s2_run(a, g_conv_ws, g_conv_ws_bytes, (hipStream_t)stream);      // both phases in the transform domain: F(2,3) and F(2,2) over the same dy rows
    // opt-in split math: one launch for both phases where the 256-row blocks fill (the x fragments of tap pairs that read the same rows are read once)
    constexpr int min_cin = 256, min_cout = 256;
    static const bool no_merge = getenv("GN_BF16X3_NO_MERGE") != nullptr;            // A/B switch: the two phase launches (tests/test_bf16x3_gpu.py)
    const bool split_ok = g_conv_math == 1 && a.Cin >= min_cin && a.Cout >= min_cout && split_worth_it(a.B, a.M, 5, a.Cin, a.Cout);
    if (split_ok && !no_merge && conv_bf16x3_merged_kind(a)) {
      if (conv_bf16x3_workspace_bytes(a.B, a.Lin, a.Cin, a.Cout, 5) > g_conv_ws_bytes) {
        set_error("conv data gradient (bf16x3 math): the split operands need %zu bytes, the workspace has %zu", conv_bf16x3_workspace_bytes(a.B, a.Lin, a.Cin, a.Cout, 5),
                  g_conv_ws_bytes);
        return GN_EWORKSPACE;
      }
      int rc = conv_bf16x3_split(a, 5, g_conv_ws, g_conv_ws_bytes, true, true, (hipStream_t)stream);
      if (rc) return rc;
      return conv_bf16x3_run_merged(a, g_conv_ws, (hipStream_t)stream);
    }
    if (!split_ok) {
      // the exact kernel's merged form (it takes the small launches): a launch the split leaves alone must find it exactly as on the default path
      bool launched = false;
      int rc = conv_pipe_try_merged(a, (hipStream_t)stream, &launched);
      if (rc || launched) return rc;
    }
  }
  PhaseScope phases(stride > 1 ? k : 0);
  for (int p = 0; p < stride; ++p) {
    if (p >= L) break;
    ConvArgs a = {};
    a.x = dy; a.w = wt; a.bias = nullptr; a.y = dx;
    a.B = B; a.Lin = Lout; a.Cin = Cout; a.Cout = Cin;
    a.M = (L - p + stride - 1) / stride; a.Ly = L;
    a.t.in_stride = 1; a.t.out_stride = stride; a.t.out_off = p;
    int nt = 0;
    for (int kk = 0; kk < k; ++kk) {
      const int d = p + pad_left - kk;
      if (((d % stride) + stride) % stride) continue;
      a.t.off[nt] = (d >= 0) ? d / stride : -((-d) / stride);
      a.t.widx[nt] = kk;
      ++nt;
    }
    GN_REQUIRE(nt > 0, "conv1d_dgrad: phase %d has no taps (k %d < stride %d)", p, k, stride);
    a.t.ntaps = nt;
    a.act = GN_ACT_LINEAR;
    a.gy = gy; a.gmask = gmask; a.gact = gact; a.gparam = gparam; a.gscale = 1.0f / (1.0f - grate);
    int rc = conv_dispatch(a, (hipStream_t)stream);
    if (rc) return rc;
  }
  return GN_OK;
}

size_t gn_conv1d_wgrad_workspace(int B, int L, int Cin, int Cout, int k, int stride, int Lout) {
  (void)L;
  size_t w = (Cin <= 4 || Cout <= 4) ? wgrad_small_workspace_bytes(B, Lout, Cin, Cout, k) : wgrad_workspace_bytes(B, Lout, Cin, Cout, k);
  if (k == 5 && stride == 1 && Cin % 64 == 0 && Cout % 64 == 0) w = std::max(w, wgrad_wino_workspace_bytes(B, Lout, Cin, Cout));      // six point slabs per split
  if (k == 5 && stride == 2 && Cin % 64 == 0 && Cout % 64 == 0) w = std::max(w, wgrad_wino_s2_workspace_bytes(B, Lout, Cin, Cout));   // seven
  size_t b = bias_grad_ws((size_t)B * Lout, Cout);
  return (w > b ? w : b) + 256;
}

int gn_conv1d_wgrad(const float* x, const float* dy, float* dw, float* db, void* ws, size_t ws_bytes, int B, int L, int Cin, int Cout, int k, int stride,
                    int pad_left, int Lout, void* stream) {
  GN_REQUIRE(x && dy && dw && ws, "conv1d_wgrad: null pointer");
  GN_REQUIRE(B > 0 && L > 0 && Cin > 0 && Cout > 0 && k >= 1 && k <= 5 && stride >= 1 && Lout > 0 && pad_left >= 0, "conv1d_wgrad: bad shape");
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (Cin <= 4 || Cout <= 4) {
    WgradSmallArgs a = {};
    a.x = x; a.dy = dy; a.part = (float*)ws;
    a.B = B; a.Lin = L; a.Cin = Cin; a.Cout = Cout; a.M = Lout; a.ntaps = k; a.in_stride = stride;
    for (int j = 0; j < k; ++j) a.off[j] = j - pad_left;
    rc = wgrad_small_dispatch(a, dw, ws_bytes, s);
  } else {
    WgradArgs a = {};
    a.x = x; a.dy = dy; a.part = (float*)ws;
    a.B = B; a.Lin = L; a.Cin = Cin; a.Cout = Cout; a.M = Lout; a.ntaps = k; a.in_stride = stride;
    for (int j = 0; j < k; ++j) a.off[j] = j - pad_left;
    a.db = db;
    if (g_conv_math == 2 && wgrad_wino_supported(a) && ws_bytes >= wgrad_wino_workspace_bytes(B, Lout, Cin, Cout)) {
      rc = wgrad_wino_run(a, dw, ws_bytes, s);                 // transform-domain weight gradient; the bias gradient takes the separate pass below
      if (rc) return rc;
      return db ? bias_grad(dy, db, (size_t)B * Lout, Cout, ws, ws_bytes, s) : GN_OK;
    }
    if (g_conv_math == 2 && wgrad_wino_s2_supported(a) && ws_bytes >= wgrad_wino_s2_workspace_bytes(B, Lout, Cin, Cout)) {
      rc = wgrad_wino_s2_run(a, dw, ws_bytes, s);              // ... of a stride-2 layer
      if (rc) return rc;
      return db ? bias_grad(dy, db, (size_t)B * Lout, Cout, ws, ws_bytes, s) : GN_OK;
    }
    constexpr int min_cin = 256, min_cout = 256;
    if (g_conv_math == 1 && Cin >= min_cin && Cout >= min_cout && split_worth_it(B, Lout, k, Cin, Cout)) {
      a.split_ws = g_conv_ws;
      a.split_ws_bytes = g_conv_ws_bytes;
    }
    rc = wgrad_mfma_dispatch(a, dw, ws_bytes, s);
    if (!rc && a.db_done) return GN_OK;                  // the weight-gradient kernel summed the bias gradient on the way
  }
  if (rc) return rc;
  if (db) rc = bias_grad(dy, db, (size_t)B * Lout, Cout, ws, ws_bytes, s);
  return rc;
}

int gn_conv2d_w2_fold(const float* w, const float* bias, float* wf, float* biasf, int kh, int Cin, int Cout, void* stream) {
  GN_REQUIRE(w && wf && kh >= 1 && Cin > 0 && Cout > 0, "conv2d_w2_fold: bad arguments");
  return conv2d_w2_fold(w, bias, wf, biasf, kh, Cin, Cout, (hipStream_t)stream);
}
int gn_conv2d_w2_unfold_grad(const float* dwf, const float* dbf, float* dw, float* db, int kh, int Cin, int Cout, void* stream) {
  GN_REQUIRE(dwf && dw && kh >= 1 && Cin > 0 && Cout > 0, "conv2d_w2_unfold_grad: bad arguments");
  return conv2d_w2_unfold(dwf, dbf, dw, db, kh, Cin, Cout, (hipStream_t)stream);
}
int gn_conv1d_up2_fold(const float* w, const float* bias, float* wf, float* biasf, int Cin, int Cout, int stride, void* stream) {
  GN_REQUIRE(w && wf && Cin > 0 && Cout > 0 && (stride == 1 || stride == 2), "conv1d_up2_fold: bad arguments (5-tap 'same' conv, stride 1 or 2)");
  return up2_fold(w, bias, wf, biasf, Cin, Cout, stride, (hipStream_t)stream);
}
int gn_conv1d_up2_unfold_grad(const float* dwf, const float* dbf, float* dw, float* db, int Cin, int Cout, int stride, void* stream) {
  GN_REQUIRE(dwf && dw && Cin > 0 && Cout > 0 && (stride == 1 || stride == 2) && (!db || dbf), "conv1d_up2_unfold_grad: bad arguments");
  return up2_unfold(dwf, dbf, dw, db, Cin, Cout, stride, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------
int gn_dense_fwd(const float* x, const float* w, const float* bias, float* y, int B, int in, int out, int act, float act_param, void* stream) {
  GN_REQUIRE(x && w && y && B >= 0 && in > 0 && out > 0, "dense_fwd: bad arguments");
  if (B == 0) return GN_OK;
  if (out <= 4) return dense_small_fwd(x, w, bias, y, B, in, out, act, act_param, (hipStream_t)stream);
  ConvArgs a = {};
  a.x = x; a.w = w; a.bias = bias; a.y = y;
  a.B = 1; a.Lin = B; a.Cin = in; a.Cout = out; a.M = B; a.Ly = B;
  fwd_taps(&a.t, 1, 1, 0);
  a.act = act; a.act_param = act_param;
  return conv_mfma_dispatch(a, (hipStream_t)stream);
}

size_t gn_dense_bwd_workspace(int B, int in, int out) {
  if (out <= 4) return 256;
  size_t w = wgrad_workspace_bytes(1, B, in, out, 1);
  size_t b = bias_grad_ws((size_t)B, out);
  return (w > b ? w : b) + (size_t)in * out * sizeof(float) + 256;
}

int gn_dense_bwd_fused(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int B, int in, int out, const uint8_t* mask_prev,
                       int act_prev, float act_param_prev, float rate_prev, void* stream) {
  GN_REQUIRE(x && w && dy && dx && dw && B > 0 && in > 0 && out >= 1 && out <= 4, "dense_bwd_fused: bad arguments (small-output heads only)");
  GN_REQUIRE(rate_prev >= 0.f && rate_prev < 1.f, "dense_bwd_fused: bad rate");
  return dense_small_bwd(x, w, dy, dx, dw, db, B, in, out, (hipStream_t)stream, act_prev, act_param_prev, mask_prev, 1.0f / (1.0f - (mask_prev ? rate_prev : 0.f)));
}

int gn_dense_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, void* ws, size_t ws_bytes, int B, int in, int out, void* stream) {
  GN_REQUIRE(x && w && dy && dw && B > 0 && in > 0 && out > 0, "dense_bwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (out <= 4) return dense_small_bwd(x, w, dy, dx, dw, db, B, in, out, s);
  GN_REQUIRE(ws && ws_bytes >= gn_dense_bwd_workspace(B, in, out), "dense_bwd: workspace too small");
  const size_t wt_bytes = (size_t)in * out * sizeof(float);
  float* wt = (float*)ws;
  void* ws2 = (char*)ws + wt_bytes;
  const size_t ws2_bytes = ws_bytes - wt_bytes;
  int rc;
  if (dx) {
    rc = transpose_w(w, wt, 1, in, out, s);
    if (rc) return rc;
    ConvArgs a = {};
    a.x = dy; a.w = wt; a.y = dx;
    a.B = 1; a.Lin = B; a.Cin = out; a.Cout = in; a.M = B; a.Ly = B;
    fwd_taps(&a.t, 1, 1, 0);
    a.act = GN_ACT_LINEAR;
    rc = conv_dispatch(a, s);
    if (rc) return rc;
  }
  WgradArgs g = {};
  g.x = x; g.dy = dy; g.part = (float*)ws2;
  g.B = 1; g.Lin = B; g.Cin = in; g.Cout = out; g.M = B; g.ntaps = 1; g.in_stride = 1; g.off[0] = 0;
  rc = wgrad_mfma_dispatch(g, dw, ws2_bytes, s);
  if (rc) return rc;
  if (db) rc = bias_grad(dy, db, (size_t)B, out, ws2, ws2_bytes, s);
  return rc;
}

// ---------------------------------------------------------------------------------------------------------
int gn_act_fwd(const float* x, float* y, size_t n, int act, float p, void* stream) {
  GN_REQUIRE(x && y, "act_fwd: null pointer");
  return act_fwd(x, y, n, act, p, (hipStream_t)stream);
}
int gn_act_bwd(const float* dy, const float* y, float* dx, size_t n, int act, float p, void* stream) {
  GN_REQUIRE(dy && y && dx, "act_bwd: null pointer");
  return act_bwd(dy, y, dx, n, act, p, (hipStream_t)stream);
}
int gn_act_dropout_bwd(const float* dy, const float* y, const uint8_t* mask, float* dx, size_t n, int act, float p, float rate, void* stream) {
  GN_REQUIRE(dy && y && mask && dx && rate >= 0.f && rate < 1.f, "act_dropout_bwd: bad arguments");
  return act_dropout_bwd(dy, y, mask, dx, n, act, p, rate, (hipStream_t)stream);
}
int gn_dropout_mask(uint8_t* mask, size_t n, float rate, uint64_t seed, uint64_t offset, void* stream) {
  GN_REQUIRE(mask && rate >= 0.f && rate < 1.f, "dropout_mask: bad arguments");
  return dropout_mask(mask, n, rate, seed, offset, (hipStream_t)stream);
}
int gn_dropout_apply(const float* x, const uint8_t* mask, float* y, size_t n, float rate, void* stream) {
  GN_REQUIRE(x && mask && y && rate >= 0.f && rate < 1.f, "dropout_apply: bad arguments");
  return dropout_apply(x, mask, y, n, rate, (hipStream_t)stream);
}
int gn_upsample2_fwd(const float* x, float* y, int B, int L, int C, void* stream) {
  GN_REQUIRE(x && y, "upsample2_fwd: null pointer");
  return upsample2_fwd(x, y, B, L, C, (hipStream_t)stream);
}
int gn_upsample2_bwd(const float* dy, float* dx, int B, int L, int C, void* stream) {
  GN_REQUIRE(dy && dx, "upsample2_bwd: null pointer");
  return upsample2_bwd(dy, dx, B, L, C, (hipStream_t)stream);
}
int gn_subtract_stack_fwd(const float* x, const float* event, float* img, int B, int n, void* stream) {
  GN_REQUIRE(x && event && img, "subtract_stack_fwd: null pointer");
  return subtract_stack_fwd(x, event, img, B, n, (hipStream_t)stream);
}
int gn_subtract_stack_bwd(const float* dimg, float* dx, int B, int n, void* stream) {
  GN_REQUIRE(dimg && dx, "subtract_stack_bwd: null pointer");
  return subtract_stack_bwd(dimg, dx, B, n, (hipStream_t)stream);
}
int gn_affine_stack_fwd(const float* x, const float* b0, const float* b1, float a0, float a1, float* img, int B, int n, void* stream) {
  GN_REQUIRE(x && img && B >= 0 && n > 0, "affine_stack_fwd: bad arguments");
  return affine_stack_fwd(x, b0, b1, a0, a1, img, B, n, (hipStream_t)stream);
}
int gn_affine_stack_bwd(const float* dimg, float a0, float a1, float* dx, int B, int n, void* stream) {
  GN_REQUIRE(dimg && dx && B >= 0 && n > 0, "affine_stack_bwd: bad arguments");
  return affine_stack_bwd(dimg, a0, a1, dx, B, n, (hipStream_t)stream);
}
int gn_assemble_d_batch(const float* real, const float* noise, const float* fake, const float* event, float* sX, int B, int n, void* stream) {
  GN_REQUIRE(real && noise && fake && event && sX && B >= 0 && n > 0, "assemble_d_batch: bad arguments");
  return assemble_d_batch(real, noise, fake, event, sX, B, n, (hipStream_t)stream);
}
int gn_fill_uniform(float* out, size_t n, float lo, float hi, uint64_t seed, uint64_t offset, void* stream) {
  GN_REQUIRE(out, "fill_uniform: null pointer");
  return fill_uniform(out, n, lo, hi, seed, offset, (hipStream_t)stream);
}
int gn_fill_normal(float* out, size_t n, float mean, float sd, uint64_t seed, uint64_t offset, void* stream) {
  GN_REQUIRE(out, "fill_normal: null pointer");
  return fill_normal(out, n, mean, sd, seed, offset, (hipStream_t)stream);
}
int gn_gather_rows(const float* src, const int64_t* idx, float* out, int rows, int width, void* stream) {
  GN_REQUIRE(src && idx && out && rows >= 0 && width > 0, "gather_rows: bad arguments");
  return gather_rows(src, idx, out, rows, width, (hipStream_t)stream);
}
int gn_axpy(float* y, const float* x, float a, size_t n, void* stream) {
  GN_REQUIRE(y && x, "axpy: null pointer");
  return axpy(y, x, a, n, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------
size_t gn_bn_stats_workspace(size_t rows, int C) { return colred_workspace_bytes(rows, C) + 256; }

int gn_bn_stats(const float* x, size_t rows, int C, double* sums, void* ws, size_t ws_bytes, void* stream) {
  GN_REQUIRE(x && sums && ws && rows > 0 && C > 0, "bn_stats: bad arguments");
  ColRedArgs r = {};
  r.a = x; r.rows = rows; r.C = C;
  return colred_run(1, r, ws, ws_bytes, sums, nullptr, (hipStream_t)stream);
}
int gn_bn_finalize(const double* sums, double count, const float* gamma, const float* beta, float eps, float momentum, float* moving_mean, float* moving_var,
                   float* scale, float* shift, float* save_mean, float* save_invstd, int C, void* stream) {
  GN_REQUIRE(sums && gamma && beta && scale && shift && save_mean && save_invstd && C > 0 && count > 1.0, "bn_finalize: bad arguments");
  GN_REQUIRE((moving_mean == nullptr) == (moving_var == nullptr), "bn_finalize: moving_mean/moving_var must both be given or both be NULL");
  return bn_finalize(sums, count, gamma, beta, eps, momentum, moving_mean, moving_var, nullptr, nullptr, 0.f, scale, shift, save_mean, save_invstd, C,
                     (hipStream_t)stream);
}

int gn_bn_finalize_zero_debias(const double* sums, double count, const float* gamma, const float* beta, float eps, float momentum, float* moving_mean,
                               float* moving_var, float* biased_mean, float* biased_var, int local_step, float* scale, float* shift, float* save_mean,
                               float* save_invstd, int C, void* stream) {
  GN_REQUIRE(sums && gamma && beta && scale && shift && save_mean && save_invstd && C > 0 && count > 1.0, "bn_finalize_zero_debias: bad arguments");
  GN_REQUIRE(moving_mean && moving_var && biased_mean && biased_var && local_step >= 1,
             "bn_finalize_zero_debias: needs moving_mean/var, the biased accumulators and the incremented local_step (>= 1, got %d)", local_step);
  return bn_finalize(sums, count, gamma, beta, eps, momentum, moving_mean, moving_var, biased_mean, biased_var, (float)local_step, scale, shift, save_mean,
                     save_invstd, C, (hipStream_t)stream);
}
int gn_bn_infer_coeffs(const float* gamma, const float* beta, const float* moving_mean, const float* moving_var, float eps, float* scale, float* shift, int C,
                       void* stream) {
  GN_REQUIRE(gamma && beta && moving_mean && moving_var && scale && shift && C > 0, "bn_infer_coeffs: bad arguments");
  return bn_infer_coeffs(gamma, beta, moving_mean, moving_var, eps, scale, shift, C, (hipStream_t)stream);
}
int gn_bn_apply(const float* x, const float* scale, const float* shift, const uint8_t* mask, float* y, size_t rows, int C, int act, float p, float rate,
                void* stream) {
  GN_REQUIRE(x && scale && shift && y && C > 0, "bn_apply: bad arguments");
  GN_REQUIRE(rate >= 0.f && rate < 1.f && (mask || rate == 0.f), "bn_apply: dropout rate %f without mask", rate);
  return bn_apply(x, scale, shift, mask, y, rows, C, act, p, mask ? rate : 0.f, (hipStream_t)stream);
}
int gn_bn_bwd_stats(const float* dy, const float* y, const float* x, const uint8_t* mask, const float* save_mean, const float* save_invstd, double* dsums, void* ws,
                    size_t ws_bytes, size_t rows, int C, int act, float p, float rate, const float* scale, const float* shift, void* stream) {
  GN_REQUIRE(dy && x && save_mean && save_invstd && dsums && ws && rows > 0 && C > 0, "bn_bwd_stats: bad arguments");
  GN_REQUIRE((scale == nullptr) == (shift == nullptr) && (y || scale), "bn_bwd_stats: needs the layer output y, or scale AND shift to recompute it");
  ColRedArgs r = {};
  r.a = dy; r.y = y; r.xpre = x; r.mask = mask; r.mean = save_mean; r.invstd = save_invstd; r.scale = scale; r.shift = shift;
  r.rows = rows; r.C = C; r.act = act; r.act_param = p; r.keep_scale = 1.0f / (1.0f - (mask ? rate : 0.f));
  return colred_run(2, r, ws, ws_bytes, dsums, nullptr, (hipStream_t)stream);
}
int gn_bn_bwd_apply(const float* dy, const float* y, const float* x, const uint8_t* mask, const float* gamma, const float* save_mean, const float* save_invstd,
                    const double* dsums_global, double count, const double* dsums_local, float* dx, float* dgamma, float* dbeta, size_t rows, int C, int act,
                    float p, float rate, const float* scale, const float* shift, void* stream) {
  GN_REQUIRE(dy && x && gamma && save_mean && save_invstd && dsums_global && dsums_local && dx && dgamma && dbeta && C > 0, "bn_bwd_apply: bad arguments");
  GN_REQUIRE((scale == nullptr) == (shift == nullptr) && (y || scale), "bn_bwd_apply: needs the layer output y, or scale AND shift to recompute it");
  return bn_bwd_apply(dy, y, x, mask, gamma, save_mean, save_invstd, dsums_global, count, dsums_local, dx, dgamma, dbeta, rows, C, act, p, mask ? rate : 0.f, scale,
                      shift, (hipStream_t)stream);
}

static int lazy_dy_check(const char* who, const float* g, const float* w, int L, int Lout, int k, int pad_left, size_t rows, int C, const float* scale,
                         const float* shift, LazyDy* z) {
  GN_REQUIRE(g && w && L > 0 && Lout > 0 && k >= 1 && k <= 5 && pad_left >= 0, "%s: bad conv description (1 filter, 1..5 taps, stride 1)", who);
  GN_REQUIRE(C % 4 == 0 && scale && shift, "%s: needs C %% 4 == 0 and the forward pass' scale / shift", who);
  GN_REQUIRE(rows % (size_t)L == 0 && rows / (size_t)L < 0x7fffffffull, "%s: rows %zu is not a whole number of length-%d segments", who, rows, L);
  z->g = g; z->w = w; z->L = L; z->Lout = Lout; z->k = k; z->pad_left = pad_left;
  return GN_OK;
}
int gn_bn_bwd_stats_conv1(const float* g, const float* w, int L, int Lout, int k, int pad_left, const float* x, const uint8_t* mask, const float* save_mean,
                          const float* save_invstd, double* dsums, void* ws, size_t ws_bytes, size_t rows, int C, int act, float p, float rate,
                          const float* scale, const float* shift, void* stream) {
  GN_REQUIRE(x && save_mean && save_invstd && dsums && ws && rows > 0 && C > 0, "bn_bwd_stats_conv1: bad arguments");
  ColRedArgs r = {};
  int rc = lazy_dy_check("bn_bwd_stats_conv1", g, w, L, Lout, k, pad_left, rows, C, scale, shift, &r.lz);
  if (rc) return rc;
  r.a = nullptr; r.y = nullptr; r.xpre = x; r.mask = mask; r.mean = save_mean; r.invstd = save_invstd; r.scale = scale; r.shift = shift;
  r.rows = rows; r.C = C; r.act = act; r.act_param = p; r.keep_scale = 1.0f / (1.0f - (mask ? rate : 0.f));
  return colred_run(2, r, ws, ws_bytes, dsums, nullptr, (hipStream_t)stream);
}
int gn_bn_bwd_apply_conv1(const float* g, const float* w, int L, int Lout, int k, int pad_left, const float* x, const uint8_t* mask, const float* gamma,
                          const float* save_mean, const float* save_invstd, const double* dsums_global, double count, const double* dsums_local, float* dx,
                          float* dgamma, float* dbeta, size_t rows, int C, int act, float p, float rate, const float* scale, const float* shift, void* stream) {
  GN_REQUIRE(x && gamma && save_mean && save_invstd && dsums_global && dsums_local && dx && dgamma && dbeta && C > 0, "bn_bwd_apply_conv1: bad arguments");
  LazyDy z = {};
  int rc = lazy_dy_check("bn_bwd_apply_conv1", g, w, L, Lout, k, pad_left, rows, C, scale, shift, &z);
  if (rc) return rc;
  return bn_bwd_apply(nullptr, nullptr, x, mask, gamma, save_mean, save_invstd, dsums_global, count, dsums_local, dx, dgamma, dbeta, rows, C, act, p,
                      mask ? rate : 0.f, scale, shift, (hipStream_t)stream, &z);
}

// ---------------------------------------------------------------------------------------------------------
int gn_bce_loss(const float* p, const float* y, float* dp, float* out, int B, int Bglobal, void* stream) {
  GN_REQUIRE(p && y && dp && out, "bce_loss: null pointer");
  return loss_run(0, p, y, dp, out, B, Bglobal, (hipStream_t)stream);
}
int gn_mse_loss(const float* p, const float* y, float* dp, float* out, int B, int Bglobal, void* stream) {
  GN_REQUIRE(p && y && dp && out, "mse_loss: null pointer");
  return loss_run(1, p, y, dp, out, B, Bglobal, (hipStream_t)stream);
}
int gn_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr_t, float b1, float b2, float eps, void* stream) {
  GN_REQUIRE(p && g && m && v, "adam_step: null pointer");
  return adam_step(p, g, m, v, n, lr_t, b1, b2, eps, (hipStream_t)stream);
}

// ---- step-varying scalars from DEVICE memory: what a captured hipGraph of a train step needs (a by-value argument is frozen at capture) ----
int gn_set_rng_base(const uint64_t* base_dev) {
  g_rng_base = base_dev;
  return GN_OK;
}
int gn_adam_step_dyn(float* p, const float* g, float* m, float* v, size_t n, const float* lr_t_dev, float b1, float b2, float eps, void* stream) {
  GN_REQUIRE(p && g && m && v && lr_t_dev, "adam_step_dyn: null pointer");
  return adam_step(p, g, m, v, n, 0.f, b1, b2, eps, (hipStream_t)stream, lr_t_dev);
}
int gn_fill_normal_dyn(float* out, size_t n, float mean, const float* sd_dev, uint64_t seed, uint64_t offset, void* stream) {
  GN_REQUIRE(out && sd_dev, "fill_normal_dyn: null pointer");
  return fill_normal(out, n, mean, 0.f, seed, offset, (hipStream_t)stream, sd_dev);
}
int gn_bn_finalize_zero_debias_dyn(const double* sums, double count, const float* gamma, const float* beta, float eps, float momentum, float* moving_mean,
                                   float* moving_var, float* biased_mean, float* biased_var, const int32_t* local_step_dev, float* scale, float* shift,
                                   float* save_mean, float* save_invstd, int C, void* stream) {
  GN_REQUIRE(sums && gamma && beta && scale && shift && save_mean && save_invstd && C > 0 && count > 1.0, "bn_finalize_zero_debias_dyn: bad arguments");
  GN_REQUIRE(moving_mean && moving_var && biased_mean && biased_var && local_step_dev, "bn_finalize_zero_debias_dyn: needs moving statistics, accumulators, step");
  return bn_finalize(sums, count, gamma, beta, eps, momentum, moving_mean, moving_var, biased_mean, biased_var, 1.f, scale, shift, save_mean, save_invstd, C,
                     (hipStream_t)stream, local_step_dev);
}

}  // extern "C"
