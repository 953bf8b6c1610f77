// gn_noise_whitened: one workgroup per row runs gen_noise -> whiten_data('td') -> crop (noise_chain.h) and writes the crop, optionally
// added to a template row, as fp64 and / or fp32.  gw_template_maker.py:161-193, :243-286; BASELINE configs[4] ("coloured-Gaussian PSD").
#include <stdlib.h>
#include "common.h"
#include "noise_chain.h"

namespace gn {

struct NoiseLaunch {
  NoiseArgs n;
  const double* add64;     // (nb, crop_len) rows the noise is added to (template crops), or NULL
  double* out64;           // (nb, crop_len) or NULL
  float* out32;            // (nb, crop_len) or NULL
  int nb, crop0, crop_len;
};

template <int LOGM, int NT>
__global__ __launch_bounds__(NT, (NT >= 1024 ? 1 : 2)) void noise_whitened_kernel(NoiseLaunch a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double2* d = reinterpret_cast<double2*>(smem_raw);
  const int b = blockIdx.x;
  noise_chain<LOGM, NT>(d, a.n, b);
  for (int n = threadIdx.x; n < a.crop_len; n += NT) {
    const size_t o = (size_t)b * a.crop_len + n;
    double v = noise_sample<LOGM>(d, a.crop0 + n);
    if (a.add64) v = a.add64[o] + v;
    if (a.out64) a.out64[o] = v;
    if (a.out32) a.out32[o] = (float)v;
  }
}

// fp64 operations per row (the figure roofline_synth prices the kernel with): three M-point transforms of 5 N log2 N / 2 each
// (N = 2 M real points -> M complex points: 5 M log2 M flop), the two pack / unpack passes (~40 flop per bin pair), window and
// scale (4 per sample), Box-Muller (log, sqrt, sincos ~ 120 flop per bin)
double noise_flops_per_row(int M) {
  double l2 = 0;
  for (int m = M; m > 1; m >>= 1) l2 += 1;
  return 3.0 * 5.0 * M * l2 + 2.0 * 40.0 * (M / 2) + 4.0 * 2.0 * M + 120.0 * (M + 1);
}

template <int LOGM, int NT>
static int launch_noise(const NoiseLaunch& a, hipStream_t s) {
  constexpr int M = 1 << LOGM;
  const size_t lds = (size_t)(M + M / 8) * sizeof(double2);
  static unsigned long long lds_done = 0;
  allow_big_lds((const void*)noise_whitened_kernel<LOGM, NT>, &lds_done);
  prof_begin(s);
  hipLaunchKernelGGL((noise_whitened_kernel<LOGM, NT>), dim3(a.nb), dim3(NT), lds, s, a);
  // algorithmic bytes: what the unfused chain's first and last stage touch per row -- the two spectra tables and the window read, the crop written
  prof_end(s, (double)a.nb * noise_flops_per_row(M), 4, (double)a.nb * ((M + 1) * 16.0 + 2.0 * M * 8.0 + (double)a.crop_len * (a.out64 ? 8.0 : 4.0)));
  return check_launch("noise_whitened");
}

int noise_whitened(const NoiseLaunch& a, int N, hipStream_t s) {
  if (a.nb == 0) return GN_OK;
  switch (N) {
    case 1024: return launch_noise<9, 128>(a, s);
    case 2048: return launch_noise<10, 256>(a, s);
    case 4096: return launch_noise<11, 256>(a, s);
    case 8192: return launch_noise<12, 256>(a, s);
    case 16384: return launch_noise<13, 1024>(a, s);
    default:
      set_error("noise_whitened: N %d unsupported (1024, 2048, 4096, 8192, 16384)", N);
      return GN_EINVAL;
  }
}

}  // namespace gn

using namespace gn;

extern "C" int gn_noise_whitened(const double* amp, const double* wscale, const double* window, const double* twiddle, const double* normals_in,
                                 double* normals_out, const double* add_f64, double* out_f64, float* out_f32, int nb, int N, int crop0, int crop_len,
                                 double df, uint64_t seed, uint64_t counter, void* stream) {
  GN_REQUIRE(amp && wscale && window && twiddle && (out_f64 || out_f32) && nb >= 0, "noise_whitened: bad arguments");
  GN_REQUIRE(crop0 >= 0 && crop_len > 0 && crop0 + crop_len <= N && df > 0, "noise_whitened: bad window (N %d crop %d+%d)", N, crop0, crop_len);
  NoiseLaunch a;
  a.n.amp = amp; a.n.wscale = wscale; a.n.win = window; a.n.W = (const double2*)twiddle; a.n.normals_in = normals_in; a.n.normals_out = normals_out;
  a.n.seed = seed; a.n.counter = counter; a.n.df = df;
  a.add64 = add_f64; a.out64 = out_f64; a.out32 = out_f32; a.nb = nb; a.crop0 = crop0; a.crop_len = crop_len;
  return noise_whitened(a, N, (hipStream_t)stream);
}
