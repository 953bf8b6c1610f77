// Posterior read-out scoring (bbhMahoGANy.py:811-873): 2-D Gaussian kernel-density estimate evaluated on a grid.
//   pdf(p) = norm * sum_i exp(-0.5 * (p - x_i)^T Sinv (p - x_i)),  the formula of scipy.stats.gaussian_kde.pdf that
//   overlap_tests calls twice on a 100 x 100 grid (:861, :866) for 4000 + 3907 samples.  fp64, one thread per grid point,
//   the sample set streamed through LDS in tiles.
#include "common.h"

namespace gn {

__global__ __launch_bounds__(256) void kde2d_kernel(const double* __restrict__ data, int n, const double* __restrict__ pts, int m, double i00, double i01,
                                                    double i11, double norm, double* __restrict__ out) {
  __shared__ double sx[256], sy[256];
  const int p = blockIdx.x * 256 + threadIdx.x;
  const double px = p < m ? pts[p] : 0.0, py = p < m ? pts[m + p] : 0.0;
  double acc = 0.0;
  for (int base = 0; base < n; base += 256) {
    const int i = base + threadIdx.x;
    sx[threadIdx.x] = i < n ? data[i] : 0.0;
    sy[threadIdx.x] = i < n ? data[n + i] : 0.0;
    __syncthreads();
    const int cnt = min(256, n - base);
    for (int k = 0; k < cnt; ++k) {
      const double dx = px - sx[k], dy = py - sy[k];
      const double e = dx * dx * i00 + 2.0 * dx * dy * i01 + dy * dy * i11;
      acc += exp(-0.5 * e);
    }
    __syncthreads();
  }
  if (p < m) out[p] = acc * norm;
}

}  // namespace gn

extern "C" int gn_kde2d_pdf(const double* data, int n, const double* pts, int m, double inv00, double inv01, double inv11, double norm, double* out,
                            void* stream) {
  GN_REQUIRE(data && pts && out && n > 0 && m >= 0, "kde2d_pdf: bad arguments");
  if (m == 0) return GN_OK;
  hipLaunchKernelGGL(gn::kde2d_kernel, dim3(gn::cdiv(m, 256)), dim3(256), 0, (hipStream_t)stream, data, n, pts, m, inv00, inv01, inv11, norm, out);
  return gn::check_launch("kde2d_pdf");
}
