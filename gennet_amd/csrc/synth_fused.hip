// Fused template synthesiser: ONE workgroup per template goes from (m1, m2, idx) to the cropped detector strain without touching HBM
// in between (gw_template_maker.py:507-565 + the crop of sim_data :695).  The unfused entry points of synth.hip (chirp spectrum ->
// irFFT x2 -> align/crop) stay for whiten_data / gen_noise / the single-template gen_bbh surface; they move ~0.5 MB per template
// through HBM for 8 KB of output.  Here HBM sees the whitening scale (L2-resident, shared by every block), the twiddle table and
// the output row.
//
// Math.  Both polarisations come from the same complex spectrum S[k] = h~(f_k) * whiten(f_k) (k = 1..M-1, M = N/2; S[0] = 0):
//     h+~ = fp * S,  hx~ = -i * ci * S       (fp = (1 + cos^2 iota)/2, ci = cos iota)
// so with the one-sided complex series a[n] = sum_{k<M} S[k] exp(+2 pi i k n / N):
//     irfft(h+~)[n] = fp/N * (2 Re a[n] + Re S[M] (-1)^n),   irfft(hx~)[n] = ci/N * (2 Im a[n] + Im S[M] (-1)^n)
// and a[] splits by sample parity into two M-point complex inverse FFTs: a[2n'] = IFFT_M(S)[n'], a[2n'+1] = IFFT_M(S * W_N^k)[n'].
// The block keeps its M spectrum bins in registers (the cbrt / pow / sincos are evaluated once) and runs the transforms in ONE LDS buffer
// (in-place decimation in time, radix-8 stages plus one radix-2/4 stage: 4-5 barriers per transform instead of 12-13), in one of two forms
// (round 4; three M-point passes until then):
//   M >= 4096 (fs >= 2048): TWO M-point passes -- even samples (arg-max; every thread keeps the detector strain (hp Fp + hc Fc) g of its M / NT even
//     samples in registers, ONE double per sample, in the registers the spectrum bins free once the odd pass has loaded them), then odd samples
//     (arg-max, the block's ref_idx, the odd half of the crop straight from LDS, the even half from the registers of whichever thread owns the sample:
//     thread np mod NT owns even sample 2 np, the crop position follows from np).  10.14 -> 12.9 M templates/s at fs 2048 against three passes.
//   M <= 2048 (fs <= 1024): FOUR M/2-point passes in a half-size image after one radix-2 step in registers (see the kernel): four workgroups per CU
//     instead of two, 16.2 -> 21.0 M templates/s at fs 1024; at the larger sizes the registers it needs do not fit four waves per SIMD and it loses.
// Measured and NOT kept: the stage twiddles from a two-level table in LDS (W^i = Tc[i >> 6] Tf[i & 63], 2 KiB) instead of three L2-resident global loads
// per butterfly -- no change: the transform does not wait for its twiddles.
// ref_idx = argmax(h+^2 + hx^2) over the rolled series (first maximum), slide = ref_idx - idx - peak_off with python slice
// semantics, zero fill past the end, exactly as align_crop_kernel (synth.hip) does.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "fft_lds.h"
#include "noise_chain.h"

namespace gn {

struct ChirpCoeffsF {
  double piM, f_merg, f_ring, sigma, f_cut, amp0, t0, wnorm;
  double psi[6];
  double nyq_re, nyq_im;
  double iv_merg;         // 1 / cbrt(piM f_merg)
  double m1, m2;          // the template's masses (given, or drawn from the prior by the block itself)
  int idx, pad_;
};

struct SynthArgs {
  const double* m1;
  const double* m2;
  const int32_t* idx;
  const double* scale;      // whitening scale per bin (Nf = M + 1)
  const double2* W;         // exp(+2 pi i k / N), k < M
  double* out64;            // (nb, crop_len) or NULL
  float* out32;             // (nb, crop_len) or NULL
  int32_t* ref_out;         // (nb,) or NULL
  int nb, N, roll, crop0, crop_len, peak_off;
  double df, f_low, dist_mpc, iota, phi0, Fp, Fc, g;
  // prior mode (m1 == NULL): every block draws its own (m1, m2, idx) from the hunt_constrain prior (gw_template_maker.py:327-339, :422-426)
  // with a counter-based Philox stream, so an on-line batch needs no host random numbers and no host -> device parameter copy
  uint64_t seed, counter;
  int idx_lo, idx_hi;
  double m_min, M_max;
  float* labels;           // (nb, 2) [mc, m2/m1] or NULL
  double* m_out;           // (nb, 2) [m1, m2] or NULL
  int32_t* idx_out;        // (nb,) or NULL
  // noise mode (nz.amp != NULL; BASELINE configs[4]): after the template's crop is formed the SAME workgroup runs gen_noise ->
  // whiten_data('td') (noise_chain.h) in the same LDS image and writes template * g + whitened noise; the template crop waits in registers
  NoiseArgs nz;
};

constexpr int kPriorTrials = 1024;   // Philox counters reserved per template (acceptance of the prior box is ~4.5 % per trial)

__constant__ double kFf[4][3] = {{2.9740e-1, 4.4810e-2, 9.5560e-2}, {5.9411e-1, 8.9794e-2, 1.9111e-1}, {5.0801e-1, 7.7515e-2, 2.2369e-2}, {8.4845e-1, 1.2848e-1, 2.7299e-1}};
__constant__ double kPsif[6][3] = {{1.7516e-1, 7.9483e-2, -7.2390e-2}, {-5.1571e1, -1.7595e1, 1.3253e1}, {6.5866e2, 1.7803e2, -1.5972e2},
                                   {-3.9031e3, -7.7493e2, 8.8195e2},   {-2.4874e4, -1.4892e3, 4.4588e3}, {2.5196e4, 3.3970e2, -3.9573e3}};
__constant__ int kOrdf[6] = {0, 2, 3, 4, 6, 7};

static constexpr double kPiF = 3.141592653589793238462643383279502884;
static constexpr double kMtsunF = 4.925491025543576e-06;
static constexpr double kMpcSecF = 3.085677581491367e22 / 299792458.0;

// same closed form as chirp_coeffs / chirp_fd_kernel of synth.hip (this project's own PhenomA-form model, oracle/synth_ref.chirp_fd)
__device__ void chirp_coeffs_f(double m1, double m2, double dist_mpc, ChirpCoeffsF* c) {
  const double Mt = m1 + m2;
  const double eta = m1 * m2 / (Mt * Mt);
  c->piM = kPiF * Mt * kMtsunF;
  double fk[4];
  for (int i = 0; i < 4; ++i) fk[i] = (kFf[i][0] * eta * eta + kFf[i][1] * eta + kFf[i][2]) / c->piM;
  c->f_merg = fk[0]; c->f_ring = fk[1]; c->sigma = fk[2]; c->f_cut = fk[3];
  for (int i = 0; i < 6; ++i) c->psi[i] = (kPsif[i][0] * eta * eta + kPsif[i][1] * eta + kPsif[i][2]) / eta;
  c->amp0 = pow(Mt * kMtsunF, 5.0 / 6.0) / (dist_mpc * kMpcSecF * pow(kPiF, 2.0 / 3.0)) * sqrt(5.0 * eta / 24.0) * pow(c->f_merg, -7.0 / 6.0);
  const double v = pow(c->piM * c->f_ring, 1.0 / 3.0);
  double dsum = 0.0;
  for (int i = 0; i < 6; ++i) dsum += c->psi[i] * ((kOrdf[i] - 5) / 3.0) * pow(v, (double)(kOrdf[i] - 5)) / c->f_ring;
  c->t0 = -dsum / (2.0 * kPiF);
  c->wnorm = (kPiF * c->sigma / 2.0) * pow(c->f_ring / c->f_merg, -2.0 / 3.0);
  c->iv_merg = 1.0 / cbrt(c->piM * c->f_merg);
}

// S[k] = h~(k df) * scale[k]  (complex; zero outside [f_low, f_cut) and at k = 0)
__device__ __forceinline__ double2 chirp_bin(const ChirpCoeffsF& c, int k, double df, double f_low, double phi0, const double* __restrict__ scale) {
  const double f = k * df;
  if (!(f >= f_low && f > 0.0 && f < c.f_cut && k > 0)) return make_double2(0.0, 0.0);
  const double v = cbrt(c.piM * f);
  const double v2 = v * v, iv = 1.0 / v;
  const double iv2 = iv * iv;
  const double pw[6] = {iv2 * iv2 * iv, iv2 * iv, iv2, iv, v, v2};
  double phase = 2.0 * kPiF * f * c.t0 + 2.0 * phi0;
#pragma unroll
  for (int i = 0; i < 6; ++i) phase = phase + c.psi[i] * pw[i];
  const double r = f / c.f_merg;
  double shape;
  // r^(-7/6) and r^(-2/3) through cr = cbrt(r) = v * iv_merg (v is there for the phase): no pow per bin (same expressions as chirp_fd_kernel)
  const double cr = v * c.iv_merg;
  if (f < c.f_merg) shape = 1.0 / (r * sqrt(cr));
  else if (f < c.f_ring) shape = 1.0 / (cr * cr);
  else shape = c.wnorm * ((1.0 / (2.0 * kPiF)) * c.sigma / ((f - c.f_ring) * (f - c.f_ring) + 0.25 * c.sigma * c.sigma));
  const double amp = c.amp0 * shape;
  double sn, cs;
  sincos(phase, &sn, &cs);
  const double w = scale[k];
  return make_double2(amp * cs * w, -amp * sn * w);       // h = amp * exp(-i phase)
}

// OCC: waves per SIMD the register allocation must allow (0: 1 for 1024 threads, 2 with the noise chain, 4 without)
template <int LOGM, int NT, bool NOISE, int OCC = 0>
__global__ __launch_bounds__(NT, (OCC ? OCC : (NT >= 1024 ? 1 : ((NOISE || LOGM > 11) ? 2 : 4)))) void synth_fused_kernel(SynthArgs a) {
  constexpr int M = 1 << LOGM, N = 2 * M, KPT = M / NT, KC = (KPT + 1) / 2;      // KC crop samples per thread in noise mode (crop_len <= M/2 = N/4)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double2* d = reinterpret_cast<double2*>(smem_raw);                                           // PH(M) padded complex values
  // the image: M / 2 complex slots (+ padding) for the template's quarter transforms; the noise chain (NOISE) runs M-point transforms in the same buffer
  constexpr bool QUARTER = LOGM <= 11;
  constexpr int IMG = (NOISE || !QUARTER) ? (M + M / 8) : (M / 2 + M / 16);
  ChirpCoeffsF* cf = reinterpret_cast<ChirpCoeffsF*>(smem_raw + (size_t)IMG * sizeof(double2));
  double* rv = reinterpret_cast<double*>(cf + 1);                                              // per-wave arg-max partials
  int* ri = reinterpret_cast<int*>(rv + 16);
  const int tid = threadIdx.x, b = blockIdx.x;

  if (a.m1 == nullptr) {
    // rejection sampling of the mass prior by wave 0: lane l tries trial 64*round + l, the LOWEST accepted trial wins (deterministic in
    // (seed, counter, b)); two log-uniform component masses in [m_min, M_max - m_min] s.t. m1 + m2 < M_max, m1 >= m2, q >= 0.5,
    // 20 <= mc <= 35 (the reference's flag expression), then idx uniform in [idx_lo, idx_hi)
    if (tid < 64) {
      const double lmin = log(a.m_min), lspan = log(a.M_max - a.m_min) - lmin;
      for (int round = 0; round < kPriorTrials / 64; ++round) {
        const Philox4 r = philox4x32_10(a.counter + (uint64_t)b * kPriorTrials + (uint64_t)(round * 64 + tid), a.seed);
        const double x1 = exp(lmin + ((double)r.v[0] + 0.5) * (1.0 / 4294967296.0) * lspan), x2 = exp(lmin + ((double)r.v[1] + 0.5) * (1.0 / 4294967296.0) * lspan);
        const double eta = x1 * x2 / ((x1 + x2) * (x1 + x2)), mc = (x1 + x2) * pow(eta, 0.6);
        const bool ok = (x1 + x2 < a.M_max) && (x1 > a.m_min) && (x2 > a.m_min) && (x1 >= x2) && (x2 / x1 >= 0.5) && (mc >= 20.0) && (mc <= 35.0);
        const unsigned long long hit = __ballot(ok);
        if (hit) {
          if (tid == (int)__ffsll((long long)hit) - 1) {
            cf->m1 = x1; cf->m2 = x2;
            cf->idx = a.idx_hi > a.idx_lo ? a.idx_lo + (int)(((uint64_t)r.v[2] * (uint64_t)(a.idx_hi - a.idx_lo)) >> 32) : a.idx_lo;
            if (a.labels) { a.labels[2 * (size_t)b] = (float)mc; a.labels[2 * (size_t)b + 1] = (float)(x2 / x1); }
            if (a.m_out) { a.m_out[2 * (size_t)b] = x1; a.m_out[2 * (size_t)b + 1] = x2; }
          }
          break;
        }
        if (round == kPriorTrials / 64 - 1 && tid == 0) { cf->m1 = 36.0; cf->m2 = 29.0; cf->idx = a.idx_lo; }   // (probability ~1e-21)
      }
    }
    __syncthreads();
    if (tid == 0 && a.idx_out) a.idx_out[b] = cf->idx;
  } else if (tid == 0) {
    cf->m1 = a.m1[b]; cf->m2 = a.m2[b]; cf->idx = a.idx[b];
  }
  if (tid == 0) {
    chirp_coeffs_f(cf->m1, cf->m2, a.dist_mpc, cf);
    const double2 nq = chirp_bin(*cf, M, a.df, a.f_low, a.phi0, a.scale);                      // Nyquist bin: irfft uses its real part only
    cf->nyq_re = nq.x; cf->nyq_im = nq.y;
  }
  __syncthreads();
  const ChirpCoeffsF c = *cf;                  // private copy (reading the coefficients from LDS per bin measured 6 % slower)

  double2 S[KPT];
#pragma unroll
  for (int j = 0; j < KPT; ++j) S[j] = chirp_bin(c, tid + j * NT, a.df, a.f_low, a.phi0, a.scale);

  const double ci = cos(a.iota);
  const double fpn = 0.5 * (1.0 + ci * ci) / (double)N, cin = ci / (double)N;
  double best = -1.0;
  int bi = 0x7fffffff;

  long start = 0;
  double keep[KC];                                            // noise mode: the template crop, n = tid + j NT
  // Two forms of the template transform (round 4).  QUARTER (M <= 2048, i.e. fs <= 1024): four M/2-point transforms in a half-size image -- the
  // spectrum is 8 bins per thread there, everything stays in registers at four waves per SIMD, and the kernel gains 30 % (16.2 -> 21.0 M templates/s
  // at fs 1024).  From M = 4096 on a thread holds 16 bins AND 16 kept samples per class pair: at the 128 registers four waves per SIMD allow the
  // quarter form spills 600-760 bytes per lane and LOSES (fs 2048: 9.7 M/s at four workgroups per CU, 10.6 M/s at two, against 11.6 M/s for the
  // two-transform form in the full image on the same box; fs 4096: 4.6 against 5.2), so the large sizes keep two M-point transforms.
  if constexpr (QUARTER) {
    // Four QUARTER transforms instead of two half ones (round 4): the series a[n], n < N, splits by n mod 4.  With X_p[k] = S[k] W_N^(p k)
    // (p = sample parity, as before) one radix-2 decimation-in-frequency step done in registers -- a thread holds bin k and its partner k + M/2 --
    //     a[4u + 2r + p] = IFFT_{M/2}(z_{p,r})[u],   z_{p,0}[k] = X_p[k] + X_p[k + M/2],   z_{p,1}[k] = (X_p[k] - X_p[k + M/2]) W_N^(2k),   k < M/2
    // leaves transforms of M/2 points: the LDS image halves (36 KiB at fs 2048: four workgroups per CU instead of two; measured with the image padded
    // back up, this kernel runs 1.57x faster at two workgroups per CU than at one).  Nothing of a sub-pass survives in LDS, so every thread keeps
    // the detector strain of its M / NT samples per class -- one double per sample, 2 M / NT in all, the registers the spectrum bins free after the
    // last load -- and the crop is written by whichever thread owns the sample.
    constexpr int LOGH = LOGM - 1, MH = M / 2, HB = KPT / 2;    // bins k = tid + j NT, j < HB, and their partners k + M/2 (index j + HB)
    static_assert(KPT % 2 == 0, "a thread must hold both halves of the spectrum");
    double qv[4][HB];                                           // qv[2r + p][j] = (hp Fp + hc Fc) g of series sample 4 (tid + j NT) + 2r + p
    auto sub_pass = [&](auto pc, auto rc) {
      constexpr int p = decltype(pc)::value, r = decltype(rc)::value;
  #pragma unroll
      for (int j = 0; j < HB; ++j) {
        const int k = tid + j * NT;
        double2 x0 = S[j], x1 = S[j + HB];
        if constexpr (p == 1) { x0 = cmulf(x0, a.W[k]); x1 = cmulf(x1, a.W[k + MH]); }
        const double2 z = r ? cmulf(csub(x0, x1), a.W[2 * k]) : cadd(x0, x1);
        d[PH(digitrev<LOGH>(k))] = z;
      }
      __syncthreads();
      ifft_lds<LOGH, NT, 2>(d, a.W);
      const double sg = p ? -1.0 : 1.0;                         // (-1)^n of the Nyquist term: n = 4u + 2r + p
  #pragma unroll
      for (int j = 0; j < HB; ++j) {
        const int u = tid + j * NT;
        const double2 z = d[PH(u)];
        const double hp = fpn * (2.0 * z.x + c.nyq_re * sg), hc = cin * (2.0 * z.y + c.nyq_im * sg);
        const double pw = hp * hp + hc * hc;
        const double t1 = hp * a.Fp, t2 = hc * a.Fc;
        qv[2 * r + p][j] = (t1 + t2) * a.g;
        int n = 4 * u + 2 * r + p - a.roll;                     // rolled index: rolled[n] = series[(n + roll) mod N]
        if (n < 0) n += N;
        if (pw > best || (pw == best && n < bi)) { best = pw; bi = n; }
      }
      __syncthreads();                                          // the image is free for the next sub-pass (or the exchange / the noise chain)
    };
    // the crop from the owners' registers: series sample s sits at rolled index r_ = (s - roll) mod N, i.e. at position r_ - start of the slid series
    // (python slice ht[start:]) and at crop position n = r_ - start - crop0; positions whose slid index start + crop0 + n is past the end are zeros
    auto emit = [&]() {
      double* xch = reinterpret_cast<double*>(d);               // noise mode: crop-indexed exchange through the (free) LDS image
      if constexpr (NOISE) {
  #pragma unroll
        for (int j = 0; j < KC; ++j) keep[j] = 0.0;
      } else {
        for (int n = tid; n < a.crop_len; n += NT) {
          if (start + a.crop0 + n < N) continue;
          const size_t o = (size_t)b * a.crop_len + n;
          if (a.out64) a.out64[o] = 0.0;
          if (a.out32) a.out32[o] = 0.f;
        }
      }
  #pragma unroll
      for (int cls = 0; cls < 4; ++cls)
  #pragma unroll
        for (int j = 0; j < HB; ++j) {
          int r_ = 4 * (tid + j * NT) + cls - a.roll;
          if (r_ < 0) r_ += N;
          const long n = (long)r_ - start - a.crop0;
          if (n < 0 || n >= a.crop_len) continue;
          if constexpr (NOISE) xch[n] = qv[cls][j];
          else {
            const size_t o = (size_t)b * a.crop_len + n;
            if (a.out64) a.out64[o] = qv[cls][j];
            if (a.out32) a.out32[o] = (float)qv[cls][j];
          }
        }
      if constexpr (NOISE) {
        __syncthreads();
  #pragma unroll
        for (int j = 0; j < KC; ++j) {
          const int n = tid + j * NT;
          if (n < a.crop_len && start + a.crop0 + n < N) keep[j] = xch[n];
        }
      }
    };

    sub_pass(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    sub_pass(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
    sub_pass(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
    sub_pass(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
    // block arg-max (first maximum): wave shuffle, then one wave over the per-wave partials
  #pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const double ov = __shfl_down(best, off, 64);
      const int oi = __shfl_down(bi, off, 64);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if ((tid & 63) == 0) { rv[tid >> 6] = best; ri[tid >> 6] = bi; }
    __syncthreads();
    if (tid < 64) {
      best = tid < NT / 64 ? rv[tid] : -2.0;
      bi = tid < NT / 64 ? ri[tid] : 0x7fffffff;
  #pragma unroll
      for (int off = 8; off >= 1; off >>= 1) {
        const double ov = __shfl_down(best, off, 64);
        const int oi = __shfl_down(bi, off, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
      }
      if (tid == 0) ri[0] = bi;
    }
    __syncthreads();
    const int ref = ri[0];
    if (tid == 0 && a.ref_out) a.ref_out[b] = ref;
    start = (long)ref - c.idx - a.peak_off;                  // python slice ht[start:]: negative counts from the end, clamped at 0
    if (start < 0) { start += N; if (start < 0) start = 0; }
    emit();
  } else {
    auto load_pass = [&](int parity) {
  #pragma unroll
      for (int j = 0; j < KPT; ++j) {
        const int k = tid + j * NT;
        double2 v = S[j];
        if (parity) v = cmulf(v, a.W[k]);
        d[PH(digitrev<LOGM>(k))] = v;
      }
      __syncthreads();
    };
    double hte[KPT];                                            // even pass: (hp Fp + hc Fc) g of even sample 2 (tid + j NT), crop_value's expression
    auto argmax_pass = [&](auto par) {
      constexpr int parity = decltype(par)::value;
      const double sg = parity ? -1.0 : 1.0;
  #pragma unroll
      for (int j = 0; j < KPT; ++j) {
        const int np = tid + j * NT;
        const double2 z = d[PH(np)];
        const double hp = fpn * (2.0 * z.x + c.nyq_re * sg), hc = cin * (2.0 * z.y + c.nyq_im * sg);
        const double pw = hp * hp + hc * hc;
        if constexpr (parity == 0) {
          const double t1 = hp * a.Fp, t2 = hc * a.Fc;
          hte[j] = (t1 + t2) * a.g;
        }
        int n = 2 * np + parity - a.roll;                       // rolled index: rolled[n] = series[(n + roll) mod N]
        if (n < 0) n += N;
        if (pw > best || (pw == best && n < bi)) { best = pw; bi = n; }
      }
    };
    // sample n of the crop, if this parity pass owns it (the zero fill past the end of the slid series belongs to the odd pass)
    auto crop_value = [&](int n, int parity, double* v) -> bool {
      const long sidx = start + a.crop0 + n;
      if (sidx >= N) { *v = 0.0; return parity == 1; }
      const int s = (int)((sidx + a.roll) % N);
      if ((s & 1) != parity) return false;
      const double sg = parity ? -1.0 : 1.0;
      const double2 z = d[PH(s >> 1)];
      const double hp = fpn * (2.0 * z.x + c.nyq_re * sg), hc = cin * (2.0 * z.y + c.nyq_im * sg);
      const double t1 = hp * a.Fp, t2 = hc * a.Fc;
      *v = (t1 + t2) * a.g;
      return true;
    };
    auto emit_odd = [&]() {                                     // odd samples from the LDS image + the zero fill past the end of the slid series
      if constexpr (NOISE) {
  #pragma unroll
        for (int j = 0; j < KC; ++j) {
          const int n = tid + j * NT;
          double v;
          if (n < a.crop_len && crop_value(n, 1, &v)) keep[j] = v;
        }
      } else {
        for (int n = tid; n < a.crop_len; n += NT) {
          double v;
          if (!crop_value(n, 1, &v)) continue;
          const size_t o = (size_t)b * a.crop_len + n;
          if (a.out64) a.out64[o] = v;
          if (a.out32) a.out32[o] = (float)v;
        }
      }
    };
    // even samples from the owners' registers: even sample 2 np sits at rolled index r = (2 np - roll) mod N, i.e. at position r - start of the slid
    // series and at crop position n = r - start - crop0 (crop_value's map read backwards: sidx = r < N always, so no zero fill here)
    auto emit_even = [&]() {
      double* xch = reinterpret_cast<double*>(d);               // noise mode: crop-indexed exchange through the (now free) LDS image
  #pragma unroll
      for (int j = 0; j < KPT; ++j) {
        int r = 2 * (tid + j * NT) - a.roll;
        if (r < 0) r += N;
        const long n = (long)r - start - a.crop0;
        if (n < 0 || n >= a.crop_len) continue;
        if constexpr (NOISE) xch[n] = hte[j];
        else {
          const size_t o = (size_t)b * a.crop_len + n;
          if (a.out64) a.out64[o] = hte[j];
          if (a.out32) a.out32[o] = (float)hte[j];
        }
      }
      if constexpr (NOISE) {
        __syncthreads();
  #pragma unroll
        for (int j = 0; j < KC; ++j) {
          const int n = tid + j * NT;
          const long sidx = start + a.crop0 + n;
          if (n < a.crop_len && sidx < N && (((sidx + a.roll) % N) & 1) == 0) keep[j] = xch[n];
        }
      }
    };

    // pass 1: even samples: arg-max, strain values into registers
    load_pass(0);
    ifft_lds<LOGM, NT>(d, a.W);
    argmax_pass(std::integral_constant<int, 0>{});
    __syncthreads();
    // pass 2: odd samples
    load_pass(1);
    ifft_lds<LOGM, NT>(d, a.W);
    argmax_pass(std::integral_constant<int, 1>{});
    // block arg-max (first maximum): wave shuffle, then one wave over the per-wave partials
  #pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const double ov = __shfl_down(best, off, 64);
      const int oi = __shfl_down(bi, off, 64);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if ((tid & 63) == 0) { rv[tid >> 6] = best; ri[tid >> 6] = bi; }
    __syncthreads();
    if (tid < 64) {
      best = tid < NT / 64 ? rv[tid] : -2.0;
      bi = tid < NT / 64 ? ri[tid] : 0x7fffffff;
  #pragma unroll
      for (int off = 8; off >= 1; off >>= 1) {
        const double ov = __shfl_down(best, off, 64);
        const int oi = __shfl_down(bi, off, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
      }
      if (tid == 0) ri[0] = bi;
    }
    __syncthreads();
    const int ref = ri[0];
    if (tid == 0 && a.ref_out) a.ref_out[b] = ref;
    start = (long)ref - c.idx - a.peak_off;                  // python slice ht[start:]: negative counts from the end, clamped at 0
    if (start < 0) { start += N; if (start < 0) start = 0; }
    emit_odd();
    if constexpr (NOISE) __syncthreads();                     // the odd samples have been read: the image becomes the exchange buffer
    emit_even();
  }
  if constexpr (NOISE) {
    __syncthreads();                                          // every read of the template image is done: the noise chain reuses it
    noise_chain<LOGM, NT>(d, a.nz, b);
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      const int n = tid + j * NT;
      if (n < a.crop_len) {
        const double v = keep[j] + noise_sample<LOGM>(d, a.crop0 + n);
        const size_t o = (size_t)b * a.crop_len + n;
        if (a.out64) a.out64[o] = v;
        if (a.out32) a.out32[o] = (float)v;
      }
    }
  }
}

// fp64 operations per template (the figure roofline_synth prices the kernel with): TWO M-point transforms (5 M log2 M each; three until round 4 --
// the third was the implementation's, not the algorithm's, and is no longer run or counted), the spectrum (cbrt, pow, sincos and the phase
// polynomial: ~150 flop per live bin, M bins), the twiddle multiply of the odd pass (6 M), two arg-max passes and the emit (~12 per sample, N samples)
static double synth_flops_per_template(int M) {
  double l2 = 0;
  for (int m = M; m > 1; m >>= 1) l2 += 1;
  return 2.0 * 5.0 * M * l2 + 150.0 * M + 6.0 * M + 12.0 * 2.0 * M;      // (four M/2-point transforms + one radix-2 step in registers = two M-point transforms)
}
double noise_flops_per_row(int M);

template <int LOGM, int NT, int OCC = 0>
static int launch_synth(const SynthArgs& a, hipStream_t s) {
  constexpr int M = 1 << LOGM;
  const bool noise = a.nz.amp != nullptr;
  const size_t lds = (size_t)((noise || LOGM > 11) ? M + M / 8 : M / 2 + M / 16) * sizeof(double2) + sizeof(ChirpCoeffsF) + 16 * sizeof(double) + 16 * sizeof(int);
  if (noise && a.crop_len > M / 2) {
    set_error("synth_templates: noise mode needs crop_len <= N/4 (crop %d, N %d)", a.crop_len, 2 * M);
    return GN_EINVAL;
  }
  static unsigned long long lds_done = 0, lds_done_nz = 0;
  prof_begin(s);
  if (noise) {
    allow_big_lds((const void*)synth_fused_kernel<LOGM, NT, true>, &lds_done_nz);
    hipLaunchKernelGGL((synth_fused_kernel<LOGM, NT, true>), dim3(a.nb), dim3(NT), lds, s, a);
  } else {
    allow_big_lds((const void*)synth_fused_kernel<LOGM, NT, false, OCC>, &lds_done);
    hipLaunchKernelGGL((synth_fused_kernel<LOGM, NT, false, OCC>), dim3(a.nb), dim3(NT), lds, s, a);
  }
  // algorithmic bytes: per template 2 spectra of Nf complex128 + the PSD read, the cropped row written; noise mode adds the
  // noise spectrum (Nf complex128) and the window (N doubles) the unfused chain reads
  prof_end(s, (double)a.nb * (synth_flops_per_template(M) + (noise ? noise_flops_per_row(M) : 0.0)), 3,
           (double)a.nb * (2.0 * (M + 1) * 16.0 + (M + 1) * 8.0 + (double)a.crop_len * (a.out64 ? 8.0 : 4.0) + (noise ? (M + 1) * 16.0 + 2.0 * M * 8.0 : 0.0)));
  return check_launch("synth_fused");
}

int synth_templates(const SynthArgs& a, hipStream_t s) {
  if (a.nb == 0) return GN_OK;
  switch (a.N) {
    case 1024: return launch_synth<9, 128>(a, s);
    case 2048: return launch_synth<10, 256>(a, s);
    case 4096: return launch_synth<11, 256>(a, s);
    // threads per block measured in round 3 (ms per 16 384 templates): N 8192: 256
    // threads 1.61, 512 threads 2.31 (two blocks per CU either way, the barriers of 8 waves cost more than their latency hiding buys);
    // N 16384 (one block per CU): 512 threads 8.26 (template + noise), 1024 threads 7.33
    case 8192: return launch_synth<12, 256>(a, s);
    case 16384: return launch_synth<13, 1024>(a, s);
    default:
      set_error("synth_templates: N %d unsupported (1024, 2048, 4096, 8192, 16384)", a.N);
      return GN_EINVAL;
  }
}

}  // namespace gn

using namespace gn;

extern "C" int gn_synth_templates(const double* m1, const double* m2, const int32_t* idx, const double* scale, const double* twiddle, double* out_f64,
                                  float* out_f32, int32_t* ref_idx, int nb, int N, int roll, int crop0, int crop_len, int peak_off, double df, double f_low,
                                  double dist_mpc, double iota, double phi0, double Fp, double Fc, double g, void* stream) {
  GN_REQUIRE(m1 && m2 && idx && scale && twiddle && (out_f64 || out_f32) && nb >= 0, "synth_templates: bad arguments");
  GN_REQUIRE(roll >= 0 && roll < N && crop0 >= 0 && crop_len > 0 && crop0 + crop_len <= N && df > 0, "synth_templates: bad window (N %d roll %d crop %d+%d)", N,
             roll, crop0, crop_len);
  SynthArgs a;
  a.m1 = m1; a.m2 = m2; a.idx = idx; a.scale = scale; a.W = (const double2*)twiddle; a.out64 = out_f64; a.out32 = out_f32; a.ref_out = ref_idx;
  a.nb = nb; a.N = N; a.roll = roll; a.crop0 = crop0; a.crop_len = crop_len; a.peak_off = peak_off;
  a.df = df; a.f_low = f_low; a.dist_mpc = dist_mpc; a.iota = iota; a.phi0 = phi0; a.Fp = Fp; a.Fc = Fc; a.g = g;
  a.seed = 0; a.counter = 0; a.idx_lo = a.idx_hi = 0; a.m_min = 5.0; a.M_max = 100.0; a.labels = nullptr; a.m_out = nullptr; a.idx_out = nullptr;
  a.nz = NoiseArgs{};
  return synth_templates(a, (hipStream_t)stream);
}

extern "C" int gn_synth_templates_prior(const double* scale, const double* twiddle, double* out_f64, float* out_f32, float* labels, double* m_out, int32_t* idx_out,
                                        int32_t* ref_idx, int nb, int N, int roll, int crop0, int crop_len, int peak_off, double df, double f_low, double dist_mpc,
                                        double iota, double phi0, double Fp, double Fc, double g, uint64_t seed, uint64_t counter, int idx_lo, int idx_hi,
                                        double m_min, double M_max, void* stream) {
  GN_REQUIRE(scale && twiddle && (out_f64 || out_f32) && nb >= 0, "synth_templates_prior: bad arguments");
  GN_REQUIRE(roll >= 0 && roll < N && crop0 >= 0 && crop_len > 0 && crop0 + crop_len <= N && df > 0, "synth_templates_prior: bad window (N %d roll %d crop %d+%d)",
             N, roll, crop0, crop_len);
  GN_REQUIRE(idx_hi >= idx_lo && m_min > 0 && M_max > 2 * m_min, "synth_templates_prior: bad prior (idx [%d, %d), masses %g .. %g)", idx_lo, idx_hi, m_min, M_max);
  SynthArgs a;
  a.m1 = nullptr; a.m2 = nullptr; a.idx = nullptr; a.scale = scale; a.W = (const double2*)twiddle; a.out64 = out_f64; a.out32 = out_f32; a.ref_out = ref_idx;
  a.nb = nb; a.N = N; a.roll = roll; a.crop0 = crop0; a.crop_len = crop_len; a.peak_off = peak_off;
  a.df = df; a.f_low = f_low; a.dist_mpc = dist_mpc; a.iota = iota; a.phi0 = phi0; a.Fp = Fp; a.Fc = Fc; a.g = g;
  a.seed = seed; a.counter = counter; a.idx_lo = idx_lo; a.idx_hi = idx_hi; a.m_min = m_min; a.M_max = M_max; a.labels = labels; a.m_out = m_out; a.idx_out = idx_out;
  a.nz = NoiseArgs{};
  return synth_templates(a, (hipStream_t)stream);
}

// Templates (given parameters, or with m1 == NULL drawn from the prior in the kernel) PLUS PSD-coloured noise whitened with the same PSD, one
// launch, one workgroup per row: gw_template_maker.py:462-575 + :695, then :161-193 and :243-286 ('td'), crop, add (BASELINE configs[4]).
extern "C" int gn_synth_templates_noise(const double* m1, const double* m2, const int32_t* idx, const double* scale, const double* twiddle, const double* noise_amp,
                                        const double* window, double* out_f64, float* out_f32, float* labels, double* m_out, int32_t* idx_out, int32_t* ref_idx,
                                        int nb, int N, int roll, int crop0, int crop_len, int peak_off, double df, double f_low, double dist_mpc, double iota,
                                        double phi0, double Fp, double Fc, double g, uint64_t seed, uint64_t counter, int idx_lo, int idx_hi, double m_min,
                                        double M_max, uint64_t noise_seed, uint64_t noise_counter, double* normals_out, void* stream) {
  GN_REQUIRE(scale && twiddle && noise_amp && window && (out_f64 || out_f32) && nb >= 0, "synth_templates_noise: bad arguments");
  GN_REQUIRE((m1 && m2 && idx) || (!m1 && !m2 && !idx), "synth_templates_noise: m1, m2, idx are given together or not at all (prior mode)");
  GN_REQUIRE(roll >= 0 && roll < N && crop0 >= 0 && crop_len > 0 && crop0 + crop_len <= N && df > 0, "synth_templates_noise: bad window (N %d roll %d crop %d+%d)",
             N, roll, crop0, crop_len);
  GN_REQUIRE(m1 || (idx_hi >= idx_lo && m_min > 0 && M_max > 2 * m_min), "synth_templates_noise: bad prior (idx [%d, %d), masses %g .. %g)", idx_lo, idx_hi, m_min, M_max);
  SynthArgs a;
  a.m1 = m1; a.m2 = m2; a.idx = idx; a.scale = scale; a.W = (const double2*)twiddle; a.out64 = out_f64; a.out32 = out_f32; a.ref_out = ref_idx;
  a.nb = nb; a.N = N; a.roll = roll; a.crop0 = crop0; a.crop_len = crop_len; a.peak_off = peak_off;
  a.df = df; a.f_low = f_low; a.dist_mpc = dist_mpc; a.iota = iota; a.phi0 = phi0; a.Fp = Fp; a.Fc = Fc; a.g = g;
  a.seed = seed; a.counter = counter; a.idx_lo = idx_lo; a.idx_hi = idx_hi; a.m_min = m_min; a.M_max = M_max; a.labels = labels; a.m_out = m_out; a.idx_out = idx_out;
  a.nz.amp = noise_amp; a.nz.wscale = scale; a.nz.win = window; a.nz.W = (const double2*)twiddle; a.nz.normals_in = nullptr; a.nz.normals_out = normals_out;
  a.nz.seed = noise_seed; a.nz.counter = noise_counter; a.nz.df = df;
  return synth_templates(a, (hipStream_t)stream);
}
