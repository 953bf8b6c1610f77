// PSD-coloured Gaussian noise, generated AND whitened inside one workgroup without touching HBM in between (BASELINE configs[4]):
//     gen_noise      gw_template_maker.py:161-193   amp = sqrt(0.25 T psd) (0 where psd == 0); re, im = amp * N(0,1) (Nf each); DC = 0;
//                                                   x = N * irfft(re + i im) * df
//     whiten_data    gw_template_maker.py:243-286   'td': xf = rfft(tukey(N, 1/8) * x); xf *= sqrt(2 invpsd / fs); xf[0] = 0; irfft(xf)
// followed by the caller's crop.  The unfused surface (templates.gen_noise / whiten_data: gn_noise_fd -> gn_irfft_f64 -> 2x gn_scale ->
// gn_mul_f64 -> gn_rfft_f64 -> gn_mul_f64 -> gn_irfft_f64) moves the N-sample fp64 series through HBM nine times; here HBM sees three
// L2-resident tables (amp, whitening scale, window), the twiddles and the output crop.
//
// Each N-point real transform runs as ONE M = N/2-point complex transform in LDS (fft_lds.h):
//   irfft:  E[k] = X[k] + conj(X[M-k]),  O[k] = (X[k] - conj(X[M-k])) W^k,  z = IFFT_M(E + i O)  ->  N irfft(X)[2n] = Re z[n], [2n+1] = Im z[n]
//           (W = exp(+2 pi i / N); numpy's irfft takes the REAL part of X[0] and X[M] only -- so does this packing)
//   rfft:   c[n] = y[2n] + i y[2n+1],  C = FFT_M(c) = conj(IFFT_M(conj c)),
//           Y[k] = ((C[k] + conj(C[M-k])) - i conj(W^k) (C[k] - conj(C[M-k]))) / 2,   Y[0] = Re C[0] + Im C[0],  Y[M] = Re C[0] - Im C[0]
// The transform leaves its output in natural order and wants its input in digit-reversed order; every pointwise pass between two
// transforms (window, spectral whitening) therefore reads its values into registers, and after a barrier writes them to the digit-reversed
// slots: the permutation costs no pass of its own.
#pragma once
#include "fft_lds.h"

namespace gn {

struct NoiseArgs {
  const double* amp;         // (M+1)  sqrt(0.25 T psd), 0 where psd == 0            (:184-186)
  const double* wscale;      // (M+1)  sqrt(2 invpsd / fs), invpsd = 0 where psd <= 0 (:273-276); bin 0 is zeroed here (:279)
  const double* win;         // (N)    tukey(N, alpha = 1/8)                          (:267)
  const double2* W;          // exp(+2 pi i k / N), k < M
  const double* normals_in;  // (nb, 2 (M+1)) [re block | im block], numpy's draw order (:187-188), or NULL: Philox
  double* normals_out;       // same layout, or NULL: the normals this launch used (so a test can feed them to the unfused chain)
  uint64_t seed, counter;    // Philox: bin k of row b takes counter + b (M+1) + k; one call -> Box-Muller pair (re, im)
  double df;                 // 1 / T_obs                                             (:182, :190)
};

// (re, im) standard normals of bin k of row `row`
template <int LOGM>
__device__ __forceinline__ double2 noise_normals(const NoiseArgs& a, int row, int k) {
  constexpr int Nf = (1 << LOGM) + 1;
  double2 z;
  if (a.normals_in) {
    const double* p = a.normals_in + (size_t)row * 2 * Nf;
    z = make_double2(p[k], p[Nf + k]);
  } else {
    const Philox4 r = philox4x32_10(a.counter + (uint64_t)row * Nf + (uint64_t)k, a.seed);
    const double u1 = u01_53(r.v[0], r.v[1]), u2 = u01_53(r.v[2], r.v[3]);
    const double rad = sqrt(-2.0 * log(u1));
    double sn, cs;
    sincos(2.0 * 3.141592653589793238462643383279502884 * u2, &sn, &cs);
    z = make_double2(rad * cs, rad * sn);
  }
  if (a.normals_out) {
    double* p = a.normals_out + (size_t)row * 2 * Nf;
    p[k] = z.x; p[Nf + k] = z.y;
  }
  return z;
}

// pack of the Hermitian half-spectrum pair (X[k], X[M-k]), 0 < k <= M/2, for the M-point inverse transform: Z[k] and Z[M-k]
__device__ __forceinline__ void irfft_pack(double2 xk, double2 xmk, double2 wk, double2* zk, double2* zmk) {
  const double2 e = make_double2(xk.x + xmk.x, xk.y - xmk.y);                    // X[k] + conj(X[M-k])
  const double2 o = cmulf(make_double2(xk.x - xmk.x, xk.y + xmk.y), wk);         // (X[k] - conj(X[M-k])) W^k
  *zk = make_double2(e.x - o.y, e.y + o.x);                                      // E + i O
  *zmk = make_double2(e.x + o.y, o.x - e.y);                                     // conj(E) + i conj(O)
}

// Runs the whole chain for row `row` in the LDS image d (PH(M) complex slots).  On return (after a barrier) d[PH(n)] holds
// N * whitened[2n] (real part) and N * whitened[2n+1] (imaginary part); noise_sample() reads one sample.
template <int LOGM, int NT>
__device__ void noise_chain(double2* d, const NoiseArgs& a, int row) {
  constexpr int M = 1 << LOGM, H = M / 2, PPT = (H + NT - 1) / NT;     // pairs (k, M-k), 1 <= k <= H, per thread (k = tid + 1 + j NT)
  static_assert(H % NT == 0, "M/2 must be a multiple of the block size");
  const int tid = threadIdx.x;

  // ---- gen_noise spectrum, packed straight into the digit-reversed slots of the first inverse transform
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int k = tid + 1 + j * NT;                                               // 1 .. H
    const double2 nk = noise_normals<LOGM>(a, row, k);
    const double ak = a.amp[k];
    const double2 xk = make_double2(ak * nk.x, ak * nk.y);
    double2 zk, zmk;
    if (k < H) {
      const double2 nm = noise_normals<LOGM>(a, row, M - k);
      const double am = a.amp[M - k];
      irfft_pack(xk, make_double2(am * nm.x, am * nm.y), a.W[k], &zk, &zmk);
      d[PH(digitrev<LOGM>(M - k))] = zmk;
    } else {                                                                       // k == M/2 pairs with itself
      irfft_pack(xk, xk, a.W[k], &zk, &zmk);
    }
    d[PH(digitrev<LOGM>(k))] = zk;
  }
  if (tid == 0) {                                                                  // bins 0 and M: X[0] = 0 (:189-190), Re X[M] only (irfft)
    (void)noise_normals<LOGM>(a, row, 0);                                          // drawn and discarded, like numpy's re[0], im[0]
    const double xm = a.amp[M] * noise_normals<LOGM>(a, row, M).x;
    d[PH(digitrev<LOGM>(0))] = make_double2(xm, -xm);                              // E = X[M], O = -X[M]
  }
  __syncthreads();
  ifft_lds<LOGM, NT>(d, a.W);

  // ---- x = N * irfft * df (N * irfft is the un-normalised sum, exactly, N being a power of two); y = tukey * x; forward transform of
  // c[n] = y[2n] + i y[2n+1] as conj(IFFT(conj c)): conj(c) goes to the digit-reversed slots
  {
    constexpr int KPT = M / NT;
    double2 c[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const int n = tid + j * NT;
      const double2 z = d[PH(n)];
      const double2 w = *reinterpret_cast<const double2*>(a.win + 2 * n);
      c[j] = make_double2(w.x * (z.x * a.df), -(w.y * (z.y * a.df)));
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < KPT; ++j) d[PH(digitrev<LOGM>(tid + j * NT))] = c[j];
    __syncthreads();
  }
  ifft_lds<LOGM, NT>(d, a.W);                                                      // d[PH(k)] = conj(C[k])

  // ---- rfft unpack -> whitening scale, DC = 0 (:276-279) -> irfft pack, pair by pair in registers
  {
    double2 zk[PPT], zmk[PPT];
    double z0 = 0.0;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int k = tid + 1 + j * NT;
      const double2 ck = d[PH(k)], cm = d[PH(M - k)];                              // conj(C[k]), conj(C[M-k])
      const double2 wk = a.W[k];
      // A = C[k] + conj(C[M-k]),  B = C[k] - conj(C[M-k]);  Y[k] = (A - i conj(W) B) / 2,  Y[M-k] = conj((A + i conj(W) B) / 2)
      const double2 A = make_double2(ck.x + cm.x, -ck.y + cm.y), B = make_double2(ck.x - cm.x, -ck.y - cm.y);
      const double2 t = cmulf(make_double2(wk.x, -wk.y), B);                       // conj(W) B
      const double2 it = make_double2(-t.y, t.x);                                  // i conj(W) B
      const double sk = a.wscale[k], sm = a.wscale[M - k];
      const double2 yk = make_double2(0.5 * (A.x - it.x) * sk, 0.5 * (A.y - it.y) * sk);
      const double2 ym = make_double2(0.5 * (A.x + it.x) * sm, -0.5 * (A.y + it.y) * sm);
      irfft_pack(yk, k < H ? ym : yk, wk, &zk[j], &zmk[j]);
    }
    if (tid == 0) {
      const double2 c0 = d[PH(0)];                                                 // conj(C[0]):  Y[M] = Re C[0] - Im C[0] = c0.x + c0.y
      const double ym = (c0.x + c0.y) * a.wscale[M];
      z0 = ym;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int k = tid + 1 + j * NT;
      d[PH(digitrev<LOGM>(k))] = zk[j];
      if (k < H) d[PH(digitrev<LOGM>(M - k))] = zmk[j];
    }
    if (tid == 0) d[PH(digitrev<LOGM>(0))] = make_double2(z0, -z0);                 // Y[0] := 0, Re Y[M] only
    __syncthreads();
  }
  ifft_lds<LOGM, NT>(d, a.W);
}

// sample s (0 <= s < N) of the whitened noise series after noise_chain
template <int LOGM>
__device__ __forceinline__ double noise_sample(const double2* d, int s) {
  const double2 z = d[PH(s >> 1)];
  return ((s & 1) ? z.y : z.x) * (1.0 / (double)(2 << LOGM));
}

}  // namespace gn
