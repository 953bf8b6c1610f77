// Template synthesiser kernels (gw_template_maker.py), fp64 like the numpy reference:
//   chirp_fd_whitened : closed-form FD inspiral-merger-ringdown chirp (this project's own model, see oracle/synth_ref.py)
//                       fused with whiten_data(.,'fd') (:243-286, :518-519)
//   irfft / rfft      : batched real FFTs, one workgroup per series, whole transform resident in LDS (<= 128 KiB of the
//                       160 KiB per CU): N-point real transform = N/2-point complex radix-2 + split/merge pass
//   align_crop        : roll, argmax(hp^2+hc^2) (first maximum), antenna combination, integer slide, crop (:521-565, :695)
//   noise_fd          : gen_noise spectrum with Philox normals (:184-190)
// All HBM-/latency-bound; per template 2*Nf*16 B in, fs*8 B out.
#include "common.h"

namespace gn {

struct ChirpCoeffs {
  double piM, f_merg, f_ring, sigma, f_cut, amp0, t0, wnorm;
  double psi[6];
  double iv_merg;          // 1 / cbrt(piM f_merg): cbrt(f / f_merg) = cbrt(piM f) * iv_merg
};

__constant__ double kF[4][3] = {{2.9740e-1, 4.4810e-2, 9.5560e-2}, {5.9411e-1, 8.9794e-2, 1.9111e-1}, {5.0801e-1, 7.7515e-2, 2.2369e-2}, {8.4845e-1, 1.2848e-1, 2.7299e-1}};
__constant__ double kPsi[6][3] = {{1.7516e-1, 7.9483e-2, -7.2390e-2}, {-5.1571e1, -1.7595e1, 1.3253e1}, {6.5866e2, 1.7803e2, -1.5972e2},
                                  {-3.9031e3, -7.7493e2, 8.8195e2},   {-2.4874e4, -1.4892e3, 4.4588e3}, {2.5196e4, 3.3970e2, -3.9573e3}};
__constant__ int kOrd[6] = {0, 2, 3, 4, 6, 7};

static constexpr double kPi = 3.141592653589793238462643383279502884;
static constexpr double kMtsun = 4.925491025543576e-06;
static constexpr double kMpcSec = 3.085677581491367e22 / 299792458.0;

__device__ ChirpCoeffs chirp_coeffs(double m1, double m2, double dist_mpc) {
  ChirpCoeffs c;
  const double M = m1 + m2;
  const double eta = m1 * m2 / (M * M);
  c.piM = kPi * M * kMtsun;
  double fk[4];
  for (int i = 0; i < 4; ++i) fk[i] = (kF[i][0] * eta * eta + kF[i][1] * eta + kF[i][2]) / c.piM;
  c.f_merg = fk[0]; c.f_ring = fk[1]; c.sigma = fk[2]; c.f_cut = fk[3];
  for (int i = 0; i < 6; ++i) c.psi[i] = (kPsi[i][0] * eta * eta + kPsi[i][1] * eta + kPsi[i][2]) / eta;
  c.amp0 = pow(M * kMtsun, 5.0 / 6.0) / (dist_mpc * kMpcSec * pow(kPi, 2.0 / 3.0)) * sqrt(5.0 * eta / 24.0) * pow(c.f_merg, -7.0 / 6.0);
  const double v = pow(c.piM * c.f_ring, 1.0 / 3.0);
  double dsum = 0.0;
  for (int i = 0; i < 6; ++i) dsum += c.psi[i] * ((kOrd[i] - 5) / 3.0) * pow(v, (double)(kOrd[i] - 5)) / c.f_ring;
  c.t0 = -dsum / (2.0 * kPi);
  c.wnorm = (kPi * c.sigma / 2.0) * pow(c.f_ring / c.f_merg, -2.0 / 3.0);
  c.iv_merg = 1.0 / cbrt(c.piM * c.f_merg);
  return c;
}

__global__ __launch_bounds__(256) void chirp_fd_kernel(const double* __restrict__ m1, const double* __restrict__ m2, const double* __restrict__ scale,
                                                       double2* __restrict__ hp, double2* __restrict__ hc, int Nf, double df, double f_low, double dist_mpc,
                                                       double iota, double phi0) {
  const int b = blockIdx.y;
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= Nf) return;
  const ChirpCoeffs c = chirp_coeffs(m1[b], m2[b], dist_mpc);
  const double f = k * df;
  double2 p = make_double2(0.0, 0.0), x = make_double2(0.0, 0.0);
  if (f >= f_low && f > 0.0 && f < c.f_cut && k > 0) {
    const double v = cbrt(c.piM * f);
    const double v2 = v * v, iv = 1.0 / v;
    const double iv2 = iv * iv;
    const double pw[6] = {iv2 * iv2 * iv, iv2 * iv, iv2, iv, v, v2};
    double phase = 2.0 * kPi * f * c.t0 + 2.0 * phi0;
    for (int i = 0; i < 6; ++i) phase = phase + c.psi[i] * pw[i];
    const double r = f / c.f_merg;
    double shape;
    // r^(-7/6) and r^(-2/3) through cr = cbrt(r) = v * iv_merg (v is there for the phase): no pow per bin
    const double cr = v * c.iv_merg;
    if (f < c.f_merg) shape = 1.0 / (r * sqrt(cr));
    else if (f < c.f_ring) shape = 1.0 / (cr * cr);
    else shape = c.wnorm * ((1.0 / (2.0 * kPi)) * c.sigma / ((f - c.f_ring) * (f - c.f_ring) + 0.25 * c.sigma * c.sigma));
    const double amp = c.amp0 * shape;
    double sn, cs;
    sincos(phase, &sn, &cs);
    const double hr = amp * cs, hi = -amp * sn;           // h = amp * exp(-i phase)
    const double ci = cos(iota);
    const double fp = 0.5 * (1.0 + ci * ci);
    const double w = scale[k];
    p = make_double2(fp * hr * w, fp * hi * w);           // h+ = (1+cos^2 i)/2 h
    x = make_double2(ci * hi * w, -ci * hr * w);          // hx = -i cos(i) h
  }
  hp[(size_t)b * Nf + k] = p;
  hc[(size_t)b * Nf + k] = x;
}

int chirp_fd_whitened(const double* m1, const double* m2, const double* scale, double* hp, double* hc, int nb, int Nf, double df, double f_low,
                      double dist_mpc, double iota, double phi0, hipStream_t s) {
  if (nb == 0) return GN_OK;
  hipLaunchKernelGGL(chirp_fd_kernel, dim3(cdiv(Nf, 256), nb), dim3(256), 0, s, m1, m2, scale, (double2*)hp, (double2*)hc, Nf, df, f_low, dist_mpc, iota, phi0);
  return check_launch("chirp_fd");
}

// ---------------------------------------------------------------------------------------------
// real FFTs.  W[k] = exp(+2 pi i k / N), k = 0..N/2-1 (fp64 table built once by the caller on the host).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ double2 cconj(double2 a) { return make_double2(a.x, -a.y); }

// in-place radix-2 DIT over LDS data already stored in bit-reversed order; SIGN=+1: exp(+i..) (inverse), -1: forward
template <int SIGN>
__device__ void fft_core(double2* d, const double2* __restrict__ W, int M, int logM, int N) {
  for (int s = 1; s <= logM; ++s) {
    const int half = 1 << (s - 1);
    const int wstep = M >> (s - 1);           // table index step: pos * (M/half)
    for (int j = threadIdx.x; j < (M >> 1); j += blockDim.x) {
      const int grp = j >> (s - 1), pos = j & (half - 1);
      const int i0 = (grp << s) + pos, i1 = i0 + half;
      double2 w = W[pos * wstep];
      if (SIGN < 0) w.y = -w.y;
      const double2 a = d[i0], bb = cmul(d[i1], w);
      d[i0] = make_double2(a.x + bb.x, a.y + bb.y);
      d[i1] = make_double2(a.x - bb.x, a.y - bb.y);
    }
    __syncthreads();
  }
}

__device__ __forceinline__ int bitrev(int x, int bits) { return (int)(__brev((unsigned)x) >> (32 - bits)); }

// numpy.fft.irfft(X, n=N): X (nb, M+1) complex -> out (nb, N) real, M = N/2
__global__ __launch_bounds__(1024) void irfft_kernel(const double2* __restrict__ X, double* __restrict__ out, const double2* __restrict__ W, int N, int logM) {
  extern __shared__ __attribute__((aligned(16))) double2 lds[];
  const int M = N >> 1;
  const double2* Xb = X + (size_t)blockIdx.x * (M + 1);
  for (int k = threadIdx.x; k < M; k += blockDim.x) {
    double2 a = Xb[k], b = cconj(Xb[M - k]);
    if (k == 0) { a.y = 0.0; b.y = 0.0; }                 // imaginary parts of DC / Nyquist are ignored
    const double2 e = make_double2(a.x + b.x, a.y + b.y);
    const double2 o = cmul(make_double2(a.x - b.x, a.y - b.y), W[k]);
    lds[bitrev(k, logM)] = make_double2(e.x - o.y, e.y + o.x);   // E + i*O
  }
  __syncthreads();
  fft_core<+1>(lds, W, M, logM, N);
  const double inv = 1.0 / (double)N;
  double2* ob = reinterpret_cast<double2*>(out + (size_t)blockIdx.x * N);
  for (int n = threadIdx.x; n < M; n += blockDim.x) {
    const double2 z = lds[n];
    ob[n] = make_double2(z.x * inv, z.y * inv);
  }
}

// numpy.fft.rfft(x): x (nb, N) real -> X (nb, M+1) complex
__global__ __launch_bounds__(1024) void rfft_kernel(const double* __restrict__ x, double2* __restrict__ X, const double2* __restrict__ W, int N, int logM) {
  extern __shared__ __attribute__((aligned(16))) double2 lds[];
  const int M = N >> 1;
  const double2* xb = reinterpret_cast<const double2*>(x + (size_t)blockIdx.x * N);
  for (int n = threadIdx.x; n < M; n += blockDim.x) lds[bitrev(n, logM)] = xb[n];
  __syncthreads();
  fft_core<-1>(lds, W, M, logM, N);
  double2* Xb = X + (size_t)blockIdx.x * (M + 1);
  for (int k = threadIdx.x; k <= M; k += blockDim.x) {
    const double2 zk = lds[k == M ? 0 : k], zm = cconj(lds[k == 0 ? 0 : M - k]);
    const double2 e = make_double2(0.5 * (zk.x + zm.x), 0.5 * (zk.y + zm.y));
    const double2 d = make_double2(0.5 * (zk.x - zm.x), 0.5 * (zk.y - zm.y));
    double2 w = (k == M) ? make_double2(-1.0, 0.0) : cconj(W[k]);        // exp(-2 pi i k / N)
    const double2 t = cmul(d, w);                                         // -i * t = (t.y, -t.x)
    Xb[k] = make_double2(e.x + t.y, e.y - t.x);
  }
}

static int fft_check(int N, int* logM) {
  if (N < 16 || N > 16384 || (N & (N - 1))) { set_error("real FFT: N %d must be a power of two in [16, 16384]", N); return GN_EINVAL; }
  int l = 0;
  while ((1 << l) < N / 2) ++l;
  *logM = l;
  return GN_OK;
}

int irfft_f64(const double* X, double* out, const double* W, int nb, int N, hipStream_t s) {
  int logM;
  int rc = fft_check(N, &logM);
  if (rc || nb == 0) return rc;
  const size_t lds = (size_t)(N / 2) * sizeof(double2);
  static unsigned long long lds_done = 0;
  allow_big_lds((const void*)irfft_kernel, &lds_done);
  const int threads = N / 4 < 1024 ? (N / 4 < 64 ? 64 : N / 4) : 1024;
  hipLaunchKernelGGL(irfft_kernel, dim3(nb), dim3(threads), lds, s, (const double2*)X, out, (const double2*)W, N, logM);
  return check_launch("irfft");
}

int rfft_f64(const double* x, double* X, const double* W, int nb, int N, hipStream_t s) {
  int logM;
  int rc = fft_check(N, &logM);
  if (rc || nb == 0) return rc;
  const size_t lds = (size_t)(N / 2) * sizeof(double2);
  static unsigned long long lds_done = 0;
  allow_big_lds((const void*)rfft_kernel, &lds_done);
  const int threads = N / 4 < 1024 ? (N / 4 < 64 ? 64 : N / 4) : 1024;
  hipLaunchKernelGGL(rfft_kernel, dim3(nb), dim3(threads), lds, s, x, (double2*)X, (const double2*)W, N, logM);
  return check_launch("rfft");
}

// ---------------------------------------------------------------------------------------------
// align + crop
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void align_crop_kernel(const double* __restrict__ hp, const double* __restrict__ hc, const int32_t* __restrict__ idx,
                                                         double* __restrict__ out, int32_t* __restrict__ ref_out, int N, int roll, int crop0, int crop_len,
                                                         int peak_off, double Fp, double Fc, double g) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const double* p = hp + (size_t)b * N;
  const double* c = hc + (size_t)b * N;
  double best = -1.0;
  int bi = 0x7fffffff;
  for (int n = tid; n < N; n += 256) {
    int s = n + roll;
    if (s >= N) s -= N;
    const double a = p[s], q = c[s];
    const double pw = a * a + q * q;
    if (pw > best) { best = pw; bi = n; }              // ascending n per thread: first maximum kept
  }
  __shared__ double sv[256];
  __shared__ int si[256];
  sv[tid] = best; si[tid] = bi;
  __syncthreads();
  for (int sft = 128; sft >= 1; sft >>= 1) {
    if (tid < sft) {
      const double v2 = sv[tid + sft];
      const int i2 = si[tid + sft];
      if (v2 > sv[tid] || (v2 == sv[tid] && i2 < si[tid])) { sv[tid] = v2; si[tid] = i2; }
    }
    __syncthreads();
  }
  const int ref = si[0];
  if (tid == 0 && ref_out) ref_out[b] = ref;
  long start = (long)ref - idx[b] - peak_off;          // python slice ht[start:]: negative counts from the end, clamped at 0
  if (start < 0) { start += N; if (start < 0) start = 0; }
  double* ob = out + (size_t)b * crop_len;
  for (int n = tid; n < crop_len; n += 256) {
    const long sidx = start + crop0 + n;
    double v = 0.0;
    if (sidx < N) {
      int s = (int)sidx + roll;
      if (s >= N) s -= N;
      const double t1 = p[s] * Fp, t2 = c[s] * Fc;
      v = (t1 + t2) * g;
    }
    ob[n] = v;
  }
}

int align_crop(const double* hp, const double* hc, const int32_t* idx, double* out, int32_t* ref_out, int nb, int N, int roll, int crop0, int crop_len,
               int peak_off, double Fp, double Fc, double g, hipStream_t s) {
  if (nb == 0) return GN_OK;
  if (roll < 0 || roll >= N || crop0 < 0 || crop_len <= 0 || crop0 + crop_len > N) { set_error("align_crop: bad window (N %d roll %d crop %d+%d)", N, roll, crop0, crop_len); return GN_EINVAL; }
  hipLaunchKernelGGL(align_crop_kernel, dim3(nb), dim3(256), 0, s, hp, hc, idx, out, ref_out, N, roll, crop0, crop_len, peak_off, Fp, Fc, g);
  return check_launch("align_crop");
}

// ---------------------------------------------------------------------------------------------
// noise spectrum, scaling, narrowing
// ---------------------------------------------------------------------------------------------
__global__ void noise_fd_kernel(const double* __restrict__ amp, double2* __restrict__ X, int Nf, uint64_t seed, uint64_t offset) {
  // per series 2*Nf normals in the order [re block | im block]; normal pair p = (2p, 2p+1) comes from one Philox call
  const int b = blockIdx.y;
  const int pairs = Nf;                                  // 2*Nf normals = Nf pairs
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= pairs) return;
  const Philox4 r = philox4x32_10(offset + (uint64_t)b * pairs + p, seed);
  const double u1 = u01_53(r.v[0], r.v[1]), u2 = u01_53(r.v[2], r.v[3]);
  const double rad = sqrt(-2.0 * log(u1));
  double sn, cs;
  sincos(2.0 * kPi * u2, &sn, &cs);
  const double z[2] = {rad * cs, rad * sn};
  double* Xd = reinterpret_cast<double*>(X + (size_t)b * Nf);
  for (int e = 0; e < 2; ++e) {
    const int v = 2 * p + e;                             // position in [re block | im block]
    const int f = v < Nf ? v : v - Nf;
    const double val = f == 0 ? 0.0 : amp[f] * z[e];
    Xd[2 * f + (v < Nf ? 0 : 1)] = val;
  }
}
int noise_fd(const double* amp, double* X, int nb, int Nf, uint64_t seed, uint64_t offset, hipStream_t s) {
  if (nb == 0) return GN_OK;
  hipLaunchKernelGGL(noise_fd_kernel, dim3(cdiv(Nf, 256), nb), dim3(256), 0, s, amp, (double2*)X, Nf, seed, offset);
  return check_launch("noise_fd");
}

__global__ void scale_f64_kernel(double* __restrict__ x, double sc, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) x[i] *= sc;
}
__global__ void mul_f64_kernel(double* __restrict__ x, const double* __restrict__ w, size_t n, size_t period, int complex_x) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const size_t e = complex_x ? i >> 1 : i;
    x[i] *= w[e % period];
  }
}
__global__ void f64_to_f32_kernel(const double* __restrict__ x, float* __restrict__ y, double sc, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = (float)(x[i] * sc);
}
static unsigned sgrid(size_t n) {
  size_t g = (n + 255) / 256;
  return (unsigned)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}
int scale_f64(double* x, double sc, size_t n, hipStream_t s) {
  if (!n) return GN_OK;
  hipLaunchKernelGGL(scale_f64_kernel, dim3(sgrid(n)), dim3(256), 0, s, x, sc, n);
  return check_launch("scale_f64");
}
int mul_f64(double* x, const double* w, size_t n, size_t period, int complex_x, hipStream_t s) {
  if (!n) return GN_OK;
  hipLaunchKernelGGL(mul_f64_kernel, dim3(sgrid(n)), dim3(256), 0, s, x, w, n, period, complex_x);
  return check_launch("mul_f64");
}
int f64_to_f32(const double* x, float* y, double sc, size_t n, hipStream_t s) {
  if (!n) return GN_OK;
  hipLaunchKernelGGL(f64_to_f32_kernel, dim3(sgrid(n)), dim3(256), 0, s, x, y, sc, n);
  return check_launch("f64_to_f32");
}

}  // namespace gn

using namespace gn;

extern "C" {

int gn_chirp_fd_whitened(const double* m1, const double* m2, const double* scale, double* out_hp, double* out_hc, int nb, int Nf, double df, double f_low,
                         double dist_mpc, double iota, double phi0, void* stream) {
  GN_REQUIRE(m1 && m2 && scale && out_hp && out_hc && nb >= 0 && Nf > 1 && df > 0, "chirp_fd_whitened: bad arguments");
  return chirp_fd_whitened(m1, m2, scale, out_hp, out_hc, nb, Nf, df, f_low, dist_mpc, iota, phi0, (hipStream_t)stream);
}
int gn_irfft_f64(const double* X, double* out, const double* twiddle, int nb, int N, void* stream) {
  GN_REQUIRE(X && out && twiddle && nb >= 0, "irfft_f64: bad arguments");
  return irfft_f64(X, out, twiddle, nb, N, (hipStream_t)stream);
}
int gn_rfft_f64(const double* x, double* X, const double* twiddle, int nb, int N, void* stream) {
  GN_REQUIRE(x && X && twiddle && nb >= 0, "rfft_f64: bad arguments");
  return rfft_f64(x, X, twiddle, nb, N, (hipStream_t)stream);
}
int gn_align_crop(const double* hp, const double* hc, const int32_t* idx, double* out, int32_t* ref_out, int nb, int N, int roll, int crop0, int crop_len,
                  int peak_off, double Fp, double Fc, double g, void* stream) {
  GN_REQUIRE(hp && hc && idx && out && nb >= 0 && N > 0, "align_crop: bad arguments");
  return align_crop(hp, hc, idx, out, ref_out, nb, N, roll, crop0, crop_len, peak_off, Fp, Fc, g, (hipStream_t)stream);
}
int gn_noise_fd(const double* amp, double* X, int nb, int Nf, uint64_t seed, uint64_t offset, void* stream) {
  GN_REQUIRE(amp && X && nb >= 0 && Nf > 1, "noise_fd: bad arguments");
  return noise_fd(amp, X, nb, Nf, seed, offset, (hipStream_t)stream);
}
int gn_scale_f64(double* x, double s, size_t n, void* stream) {
  GN_REQUIRE(x, "scale_f64: null pointer");
  return scale_f64(x, s, n, (hipStream_t)stream);
}
int gn_mul_f64(double* x, const double* w, size_t n, size_t period, int complex_x, void* stream) {
  GN_REQUIRE(x && w && period > 0, "mul_f64: bad arguments");
  return mul_f64(x, w, n, period, complex_x, (hipStream_t)stream);
}
int gn_f64_to_f32(const double* x, float* y, double s, size_t n, void* stream) {
  GN_REQUIRE(x && y, "f64_to_f32: null pointer");
  return f64_to_f32(x, y, s, n, (hipStream_t)stream);
}

}  // extern "C"
