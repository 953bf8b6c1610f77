// M-point complex inverse FFT resident in LDS (fp64), shared by the fused template synthesiser (synth_fused.hip) and the fused coloured-noise
// chain (noise_chain.h).  In-place decimation in time: the caller stores input bin k at PH(digitrev<LOGM>(k)), the transform leaves output
// sample n at PH(n).  Radix-8 stages (span 1, 8, 64, ...) plus one radix-2 / radix-4 stage: 4-5 barriers per transform instead of 12-13.
#pragma once
#include "common.h"

namespace gn {

__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
// complex product with two fused multiply-adds (the library is built with -ffp-contract=off, so the fusion is spelled out): 4 instead of 6 fp64
// instructions per product, and the transforms are mostly products -- each component has one rounding less than the unfused form
__device__ __forceinline__ double2 cmulf(double2 a, double2 b) { return make_double2(fma(a.x, b.x, -(a.y * b.y)), fma(a.x, b.y, a.y * b.x)); }
__device__ __forceinline__ double2 muli(double2 a) { return make_double2(-a.y, a.x); }         // i * a

// inverse (exp(+i ..)) 8-point DFT, natural order in and out
__device__ __forceinline__ void dft8_inv(double2 (&a)[8]) {
  const double r = 0.70710678118654752440;
  const double2 t0 = cadd(a[0], a[4]), t1 = csub(a[0], a[4]), t2 = cadd(a[2], a[6]), t3 = csub(a[2], a[6]);
  const double2 t4 = cadd(a[1], a[5]), t5 = csub(a[1], a[5]), t6 = cadd(a[3], a[7]), t7 = csub(a[3], a[7]);
  const double2 e0 = cadd(t0, t2), e1 = csub(t0, t2), e2 = cadd(t4, t6), e3 = muli(csub(t4, t6));
  a[0] = cadd(e0, e2); a[4] = csub(e0, e2); a[2] = cadd(e1, e3); a[6] = csub(e1, e3);
  const double2 v1 = make_double2((t5.x - t5.y) * r, (t5.x + t5.y) * r), v2 = muli(t3), v3 = make_double2((-t7.x - t7.y) * r, (t7.x - t7.y) * r);
  const double2 f0 = cadd(t1, v2), f1 = csub(t1, v2), f2 = cadd(v1, v3), f3 = muli(csub(v1, v3));
  a[1] = cadd(f0, f2); a[5] = csub(f0, f2); a[3] = cadd(f1, f3); a[7] = csub(f1, f3);
}

// LDS image: one complex slot of padding after every 8 (the span-1 stage reads 8 consecutive values per thread: 144-byte lane stride
// instead of 128 keeps ds_read_b128 conflict-free; later stages read consecutive values across lanes)
__device__ __forceinline__ int PH(int i) { return i + (i >> 3); }

// Stage radices: LOGM/3 radix-8 stages (span 1, 8, 64, ...) then one radix-2 (LOGM % 3 == 1) or radix-4 (== 2) stage.  Storage position
// of input bin k for the in-place decimation-in-time transform = mixed-radix digit reversal: the LAST stage's digit is the least
// significant digit of k and selects the outermost block.
template <int LOGM>
__device__ __forceinline__ int digitrev(int k) {
  constexpr int NR8 = LOGM / 3, REM = LOGM % 3;
  int p = 0;
  if (REM) {
    p = (k & ((1 << REM) - 1)) << (3 * NR8);
    k >>= REM;
  }
#pragma unroll
  for (int j = NR8 - 1; j >= 0; --j) {
    p += (k & 7) << (3 * j);
    k >>= 3;
  }
  return p;
}

// WS: stride of the twiddle table.  W holds exp(+2 pi i k / N) for the caller's N; a transform of M points wants the table of N' = 2 M, which is
// every (N / N')-th entry: WS = N / (2 M)  (1 for the M = N / 2 transforms, 2 for the quarter transforms of synth_fused.hip).
template <int LOGM, int NT, int WS = 1>
__device__ void ifft_lds(double2* d, const double2* __restrict__ W) {
  constexpr int M = 1 << LOGM, NR8 = LOGM / 3, REM = LOGM % 3;
  const int tid = threadIdx.x;
#pragma unroll
  for (int s = 0; s < NR8; ++s) {
    const int lspan = 3 * s, span = 1 << lspan;
    const int lstep = LOGM + 1 - lspan - 3;                  // table step N / (8 span), N = 2 M
    for (int bf = tid; bf < M / 8; bf += NT) {
      const int g = bf >> lspan, pos = bf & (span - 1);
      const int base = (g << (lspan + 3)) + pos;
      double2 a[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) a[q] = d[PH(base + (q << lspan))];
      if (s > 0) {
        const int i1 = pos << lstep;
        const double2 w1 = W[WS * i1], w2 = W[WS * 2 * i1], w4 = W[WS * 4 * i1];
        const double2 w3 = cmulf(w1, w2);
        a[1] = cmulf(a[1], w1); a[2] = cmulf(a[2], w2); a[3] = cmulf(a[3], w3); a[4] = cmulf(a[4], w4);
        a[5] = cmulf(a[5], cmulf(w4, w1)); a[6] = cmulf(a[6], cmulf(w4, w2)); a[7] = cmulf(a[7], cmulf(w4, w3));
      }
      dft8_inv(a);
#pragma unroll
      for (int q = 0; q < 8; ++q) d[PH(base + (q << lspan))] = a[q];
    }
    __syncthreads();
  }
  if (REM == 1) {
    constexpr int lspan = 3 * NR8, span = 1 << lspan;        // last stage: span = M / 2, table step N / (2 span) = 2
    for (int bf = tid; bf < M / 2; bf += NT) {
      const int pos = bf & (span - 1);
      const double2 x0 = d[PH(pos)], x1 = cmulf(d[PH(pos + span)], W[WS * (pos << 1)]);
      d[PH(pos)] = cadd(x0, x1);
      d[PH(pos + span)] = csub(x0, x1);
    }
    __syncthreads();
  } else if (REM == 2) {
    constexpr int lspan = 3 * NR8, span = 1 << lspan;        // last stage: span = M / 4, table step N / (4 span) = 2
    for (int bf = tid; bf < M / 4; bf += NT) {
      const int pos = bf & (span - 1);
      const int i1 = pos << 1;
      const double2 w1 = W[WS * i1], w2 = W[WS * 2 * i1];
      const double2 u0 = d[PH(pos)], u1 = cmulf(d[PH(pos + span)], w1), u2 = cmulf(d[PH(pos + 2 * span)], w2), u3 = cmulf(d[PH(pos + 3 * span)], cmulf(w1, w2));
      const double2 s02 = cadd(u0, u2), d02 = csub(u0, u2), s13 = cadd(u1, u3), d13 = muli(csub(u1, u3));
      d[PH(pos)] = cadd(s02, s13);
      d[PH(pos + span)] = cadd(d02, d13);
      d[PH(pos + 2 * span)] = csub(s02, s13);
      d[PH(pos + 3 * span)] = csub(d02, d13);
    }
    __syncthreads();
  }
}

}  // namespace gn
