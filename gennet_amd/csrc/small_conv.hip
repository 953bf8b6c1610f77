// HBM-bound convolution edge cases that do not belong on the matrix cores:
//   * small-Cin (1..4 input channels): first layers of the point-estimator (bbhMahoGANy.py:362,:382), the folded first
//     discriminator Conv2D (:439, Cin' = 2), and the data gradient of the generator's Cout=1 output conv (:292).
//     Write-bound: each thread keeps its weights in registers and streams float4 outputs.
//   * small-Cout (1..4 output channels): generator output conv (:292) and the data gradient of the small-Cin layers.
//     Read-bound: one wave per output row, lanes across channels, shuffle reduction.
//   * flatten -> Dense(1) heads (:377,:399,:494): one 1024-thread block per sample.
// All use the same generic tap description as conv_mfma.hip.
#include "common.h"

namespace gn {



// ---------------------------------------------------------------------------------------------
// small Cin: y[b, os*m+o0, n] = act(bias[n] + sum_j sum_c x[b, is*m+off_j, c] * w[widx_j, c, n]),  Cin <= 4, Cout % 4 == 0
// block = 256 threads: NQc = min(Cout/4, 256) float4 columns x (256/NQc) row lanes; MT rows per block.
// ---------------------------------------------------------------------------------------------
constexpr int SMALLCIN_WIN = (2 * 255 + 5) * 4 + 4;   // floats of LDS input window: 256 rows at stride 2, 5 taps, 4 channels

template <int CIN>
__global__ __launch_bounds__(256) void conv_smallcin_kernel(ConvArgs a, int m_tiles, int MT) {
  constexpr int MAXT = 5;
  const int NQ = a.Cout >> 2;
  const int NQc = NQ < 256 ? NQ : 256;
  const int RL = 256 / NQc;
  const int tid = threadIdx.x;
  const int q0 = tid % NQc, rl = tid / NQc;
  const int m_tile = blockIdx.x % m_tiles, b = blockIdx.x / m_tiles;
  const int m_lo = m_tile * MT, m_hi = min(a.M, m_lo + MT);
  const float* xb = a.x + (size_t)b * a.Lin * CIN;
  float* yb = a.y + (size_t)b * a.Ly * a.Cout;
  // the input window of the block's rows goes through LDS once (zeros outside [0, Lin)): the row loop then issues one store per
  // row and nothing else on the vector-memory path (it issued ntaps * CIN broadcast loads per row before: 2.3 -> 3.9 TB/s written)
  __shared__ float xw[SMALLCIN_WIN];
  int minoff = a.t.off[0], maxoff = a.t.off[0];
  for (int j = 1; j < a.t.ntaps; ++j) {
    minoff = min(minoff, a.t.off[j]);
    maxoff = max(maxoff, a.t.off[j]);
  }
  const int t_lo = a.t.in_stride * m_lo + minoff;
  const int n_win = (a.t.in_stride * (m_hi - 1 - m_lo) + (maxoff - minoff) + 1) * CIN;
  for (int idx = tid; idx < n_win; idx += 256) {
    const int t = t_lo + idx / CIN;
    xw[idx] = (t >= 0 && t < a.Lin) ? xb[(size_t)t * CIN + idx % CIN] : 0.f;
  }
  __syncthreads();
  if (rl >= RL) return;
  for (int q = q0; q < NQ; q += NQc) {
    float4 wv[MAXT][CIN];
#pragma unroll
    for (int j = 0; j < MAXT; ++j)
#pragma unroll
      for (int c = 0; c < CIN; ++c)
        wv[j][c] = j < a.t.ntaps ? *reinterpret_cast<const float4*>(a.w + ((size_t)a.t.widx[j] * CIN + c) * a.Cout + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bias) bias = *reinterpret_cast<const float4*>(a.bias + 4 * q);
#pragma unroll 4
    for (int m = m_lo + rl; m < m_hi; m += RL) {
      float4 s = bias;
#pragma unroll
      for (int j = 0; j < MAXT; ++j) {
        if (j >= a.t.ntaps) break;
        const float* xr = xw + (a.t.in_stride * m + a.t.off[j] - t_lo) * CIN;
#pragma unroll
        for (int c = 0; c < CIN; ++c) {
          const float xv = xr[c];
          s.x = fmaf(xv, wv[j][c].x, s.x); s.y = fmaf(xv, wv[j][c].y, s.y);
          s.z = fmaf(xv, wv[j][c].z, s.z); s.w = fmaf(xv, wv[j][c].w, s.w);
        }
      }
      s.x = act_apply(s.x, a.act, a.act_param); s.y = act_apply(s.y, a.act, a.act_param);
      s.z = act_apply(s.z, a.act, a.act_param); s.w = act_apply(s.w, a.act, a.act_param);
      const size_t o = (size_t)(a.t.out_stride * m + a.t.out_off) * a.Cout + 4 * q;
      if (a.mask) {
        const uchar4 k = *reinterpret_cast<const uchar4*>(a.mask + (size_t)b * a.Ly * a.Cout + o);
        s.x = k.x ? s.x * a.keep_scale : 0.f; s.y = k.y ? s.y * a.keep_scale : 0.f;
        s.z = k.z ? s.z * a.keep_scale : 0.f; s.w = k.w ? s.w * a.keep_scale : 0.f;
      }
      *reinterpret_cast<float4*>(yb + o) = s;
    }
  }
}

int conv_smallcin_dispatch(const ConvArgs& a, hipStream_t s) {
  if (a.Cin < 1 || a.Cin > 4 || a.Cout % 4 || a.t.ntaps > 5) {
    set_error("conv_smallcin: Cin %d (1..4) / Cout %d (%%4) / ntaps %d (<=5) unsupported", a.Cin, a.Cout, a.t.ntaps);
    return GN_EINVAL;
  }
  const int NQ = a.Cout / 4, NQc = NQ < 256 ? NQ : 256, RL = 256 / NQc;
  int MT = RL * 16 < 256 ? RL * 16 : 256;  // 16 rows per thread amortise the register-resident weights; <= 256: the LDS input window
  // the block's input window (in_stride * (MT-1) + tap span + 1 rows of Cin floats) must fit the kernel's LDS array: strides > 2 and
  // dilated tap sets get fewer rows per block, and a tap span that does not fit even one row is refused
  int minoff = a.t.off[0], maxoff = a.t.off[0];
  for (int j = 1; j < a.t.ntaps; ++j) {
    minoff = a.t.off[j] < minoff ? a.t.off[j] : minoff;
    maxoff = a.t.off[j] > maxoff ? a.t.off[j] : maxoff;
  }
  if (a.t.in_stride < 1) {
    set_error("conv_smallcin: in_stride %d < 1", a.t.in_stride);
    return GN_EINVAL;
  }
  auto window = [&](int mt) { return ((long long)a.t.in_stride * (mt - 1) + (maxoff - minoff) + 1) * a.Cin; };
  while (MT > 1 && window(MT) > SMALLCIN_WIN) MT >>= 1;
  if (window(MT) > SMALLCIN_WIN) {
    set_error("conv_smallcin: tap span %d x Cin %d does not fit the %d-float input window", maxoff - minoff + 1, a.Cin, SMALLCIN_WIN);
    return GN_EINVAL;
  }
  const int m_tiles = cdiv(a.M, MT);
  const unsigned grid = (unsigned)m_tiles * a.B;
  if (grid == 0) return GN_OK;
  switch (a.Cin) {
    case 1: hipLaunchKernelGGL(conv_smallcin_kernel<1>, dim3(grid), dim3(256), 0, s, a, m_tiles, MT); break;
    case 2: hipLaunchKernelGGL(conv_smallcin_kernel<2>, dim3(grid), dim3(256), 0, s, a, m_tiles, MT); break;
    case 3: hipLaunchKernelGGL(conv_smallcin_kernel<3>, dim3(grid), dim3(256), 0, s, a, m_tiles, MT); break;
    default: hipLaunchKernelGGL(conv_smallcin_kernel<4>, dim3(grid), dim3(256), 0, s, a, m_tiles, MT); break;
  }
  return check_launch("conv_smallcin");
}

// ---------------------------------------------------------------------------------------------
// small Cout: one wave per output row (b, m); lanes across Cin in float4 steps.  Cin % 4 == 0, Cout <= 4.
// w layout [tap][Cin][Cout].
// ---------------------------------------------------------------------------------------------
template <int COUT>
__global__ __launch_bounds__(256) void conv_smallcout_kernel(ConvArgs a) {
  const int lane = threadIdx.x & 63;
  const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (size_t)a.B * a.M) return;
  const int b = (int)(row / a.M), m = (int)(row % a.M);
  const float* xb = a.x + (size_t)b * a.Lin * a.Cin;
  float acc[COUT];
#pragma unroll
  for (int o = 0; o < COUT; ++o) acc[o] = 0.f;
  for (int j = 0; j < a.t.ntaps; ++j) {
    const int t = a.t.in_stride * m + a.t.off[j];
    if (t < 0 || t >= a.Lin) continue;
    const float* xr = xb + (size_t)t * a.Cin;
    const float* wj = a.w + (size_t)a.t.widx[j] * a.Cin * COUT;
    for (int c = 4 * lane; c < a.Cin; c += 256) {
      const float4 xv = *reinterpret_cast<const float4*>(xr + c);
      if (COUT == 1) {
        const float4 wv = *reinterpret_cast<const float4*>(wj + c);
        acc[0] = fmaf(xv.x, wv.x, acc[0]); acc[0] = fmaf(xv.y, wv.y, acc[0]);
        acc[0] = fmaf(xv.z, wv.z, acc[0]); acc[0] = fmaf(xv.w, wv.w, acc[0]);
      } else {
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int o = 0; o < COUT; ++o) acc[o] = fmaf(xs[e], wj[(size_t)(c + e) * COUT + o], acc[o]);
      }
    }
  }
#pragma unroll
  for (int o = 0; o < COUT; ++o) {
    float v = acc[o];
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) v += __shfl_xor(v, sft, 64);
    acc[o] = v;
  }
  if (lane == 0) {
    float* yr = a.y + ((size_t)b * a.Ly + (size_t)(a.t.out_stride * m + a.t.out_off)) * COUT;
#pragma unroll
    for (int o = 0; o < COUT; ++o) yr[o] = act_apply(acc[o] + (a.bias ? a.bias[o] : 0.f), a.act, a.act_param);
  }
}

// Cout == 1, unit strides, contiguous taps (the generator's output conv, bbhMahoGANy.py:292): one wave owns a RUN of ROWS
// consecutive output rows, reads each of the ROWS + NTAPS - 1 input rows it needs ONCE (the kernel above re-reads every row
// NTAPS times through L1/L2) and scatters the per-tap partial dot products into ROWS lane-private accumulators.
template <int NTAPS, int ROWS>
__global__ __launch_bounds__(256) void conv_cout1_rows_kernel(ConvArgs a, int runs_per_b) {
  const int lane = threadIdx.x & 63;
  const size_t run = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (run >= (size_t)a.B * runs_per_b) return;
  const int b = (int)(run / runs_per_b), m0 = (int)(run % runs_per_b) * ROWS;
  const float* xb = a.x + (size_t)b * a.Lin * a.Cin;
  const int t0 = m0 + a.t.off[0];
  float acc[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) acc[r] = 0.f;
  for (int c = 4 * lane; c < a.Cin; c += 256) {
    float4 wv[NTAPS];
#pragma unroll
    for (int j = 0; j < NTAPS; ++j) wv[j] = *reinterpret_cast<const float4*>(a.w + (size_t)a.t.widx[j] * a.Cin + c);
#pragma unroll
    for (int tl = 0; tl < ROWS + NTAPS - 1; ++tl) {
      const int t = t0 + tl;
      float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (t >= 0 && t < a.Lin) xv = *reinterpret_cast<const float4*>(xb + (size_t)t * a.Cin + c);
#pragma unroll
      for (int j = 0; j < NTAPS; ++j) {
        const int r = tl - j;                      // output row (local) that tap j of input row tl feeds
        if (r >= 0 && r < ROWS) {
          acc[r] = fmaf(xv.x, wv[j].x, acc[r]); acc[r] = fmaf(xv.y, wv[j].y, acc[r]);
          acc[r] = fmaf(xv.z, wv[j].z, acc[r]); acc[r] = fmaf(xv.w, wv[j].w, acc[r]);
        }
      }
    }
  }
  float mine = 0.f;
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    float v = acc[r];
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) v += __shfl_xor(v, sft, 64);
    if (lane == r) mine = v;
  }
  if (lane < ROWS && m0 + lane < a.M)
    a.y[(size_t)b * a.Ly + (size_t)(m0 + lane + a.t.out_off)] = act_apply(mine + (a.bias ? a.bias[0] : 0.f), a.act, a.act_param);
}

int conv_smallcout_dispatch(const ConvArgs& a, hipStream_t s) {
  if (a.Cout < 1 || a.Cout > 4 || a.Cin % 4) {
    set_error("conv_smallcout: Cout %d (1..4) / Cin %d (%%4) unsupported", a.Cout, a.Cin);
    return GN_EINVAL;
  }
  const size_t rows = (size_t)a.B * a.M;
  if (rows == 0) return GN_OK;
  bool contiguous = a.t.in_stride == 1 && a.t.out_stride == 1 && a.t.ntaps == 5;
  for (int j = 1; j < a.t.ntaps; ++j) contiguous = contiguous && a.t.off[j] == a.t.off[0] + j;
  if (a.Cout == 1 && contiguous && a.Cin >= 256) {
    constexpr int ROWS = 16;
    const int runs_per_b = cdiv(a.M, ROWS);
    hipLaunchKernelGGL((conv_cout1_rows_kernel<5, ROWS>), dim3(cdiv((size_t)a.B * runs_per_b, 4)), dim3(256), 0, s, a, runs_per_b);
    return check_launch("conv_cout1_rows");
  }
  const unsigned grid = cdiv(rows, 4);
  switch (a.Cout) {
    case 1: hipLaunchKernelGGL(conv_smallcout_kernel<1>, dim3(grid), dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL(conv_smallcout_kernel<2>, dim3(grid), dim3(256), 0, s, a); break;
    case 3: hipLaunchKernelGGL(conv_smallcout_kernel<3>, dim3(grid), dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL(conv_smallcout_kernel<4>, dim3(grid), dim3(256), 0, s, a); break;
  }
  return check_launch("conv_smallcout");
}

// ---------------------------------------------------------------------------------------------
// weight gradient when Cin or Cout is tiny: dw[j, c, n] = sum_{b,m} x[b, is*m+off_j, c] * dy[b, m, n]
// (ntaps*CS) scalar-by-row products per row, CS = the small side.  Grid: (column blocks, row chunks); each thread owns
// one float4 column group of the LARGE side and accumulates ntaps*CS float4 sums over its rows; row lanes reduced
// through LDS; partials [chunk][ntaps*CS][Clarge] summed by the caller-visible reduce (fixed order).
//   SMALL_IS_IN = true : x has CS channels (small Cin), dy has Clarge = Cout channels -> dw[j][c][n]
//   SMALL_IS_IN = false: dy has CS channels (small Cout), x has Clarge = Cin channels -> dw[j][n][c]
// ---------------------------------------------------------------------------------------------

// Small Cin (SMALL_IS_IN = true of the kernel below) with the x values every (row, tap, channel) meets staged through an LDS
// table per 128-row tile (zeros outside [0, Lin)) and the dy rows loaded U at a time: the row loop then has one 16-byte load per
// row on the vector-memory path instead of 1 + ntaps * CS dependent broadcast loads (measured 1.8 -> 2.6 TB/s of dy read,
// reduce included).
template <int CS>
__global__ __launch_bounds__(256) void wgrad_smallcin_tab_kernel(WgradSmallArgs a) {
  constexpr int MAXT = 5, TR = 128, U = 4;
  const int NQ = a.Cout >> 2;
  const int NQc = NQ < 256 ? NQ : 256;
  const int RL = 256 / NQc;
  const int tid = threadIdx.x;
  const int ql = tid % NQc, rl = tid / NQc;
  const int q = blockIdx.x * NQc + ql;
  const bool active = (rl < RL) && (q < NQ);
  float4 acc[MAXT][CS];
#pragma unroll
  for (int j = 0; j < MAXT; ++j)
#pragma unroll
    for (int c = 0; c < CS; ++c) acc[j][c] = make_float4(0.f, 0.f, 0.f, 0.f);
  __shared__ float tab[TR][MAXT * CS];
  __shared__ float4 red[256];

  const size_t rows = (size_t)a.B * a.M;
  const size_t r_lo = (size_t)blockIdx.y * a.rows_per_chunk;
  const size_t r_hi = r_lo + a.rows_per_chunk < rows ? r_lo + a.rows_per_chunk : rows;
  const int per_row = a.ntaps * CS;
  for (size_t base = r_lo; base < r_hi; base += TR) {
    const int nr = (int)(r_hi - base < (size_t)TR ? r_hi - base : (size_t)TR);
    __syncthreads();                                   // the previous tile's table has been consumed
    for (int e = tid; e < nr * per_row; e += 256) {
      const int rr = e / per_row, jc = e % per_row;
      const int j = jc / CS, c = jc % CS;
      const size_t r = base + rr;
      const int b = (int)(r / a.M), m = (int)(r % a.M);
      const int t = a.in_stride * m + a.off[j];
      tab[rr][j * CS + c] = (t >= 0 && t < a.Lin) ? a.x[((size_t)b * a.Lin + t) * CS + c] : 0.f;
    }
    __syncthreads();
    if (active) {
      for (int r0 = rl; r0 < nr; r0 += RL * U) {
        float4 g[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int rr = r0 + u * RL;
          g[u] = rr < nr ? *reinterpret_cast<const float4*>(a.dy + (base + rr) * a.Cout + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int rr = r0 + u * RL;
          if (rr >= nr) break;
#pragma unroll
          for (int j = 0; j < MAXT; ++j) {
            if (j >= a.ntaps) break;
#pragma unroll
            for (int c = 0; c < CS; ++c) {
              const float xv = tab[rr][j * CS + c];
              acc[j][c].x = fmaf(xv, g[u].x, acc[j][c].x); acc[j][c].y = fmaf(xv, g[u].y, acc[j][c].y);
              acc[j][c].z = fmaf(xv, g[u].z, acc[j][c].z); acc[j][c].w = fmaf(xv, g[u].w, acc[j][c].w);
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < MAXT; ++j) {
    if (j >= a.ntaps) break;
#pragma unroll
    for (int c = 0; c < CS; ++c) {
      __syncthreads();
      red[tid] = acc[j][c];
      __syncthreads();
      if (rl == 0 && q < NQ) {
        float4 s = red[ql];
        for (int k = 1; k < RL; ++k) {
          const float4 v = red[k * NQc + ql];
          s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        float* pc = a.part + (size_t)blockIdx.y * a.ntaps * a.Cin * a.Cout;
        *reinterpret_cast<float4*>(pc + ((size_t)j * CS + c) * a.Cout + 4 * q) = s;      // dw[j][c][n]
      }
    }
  }
}

template <int CS, bool SMALL_IS_IN>
__global__ __launch_bounds__(256) void wgrad_small_kernel(WgradSmallArgs a) {
  constexpr int MAXT = 5;
  const int CL = SMALL_IS_IN ? a.Cout : a.Cin;
  const int NQ = CL >> 2;
  const int NQc = NQ < 256 ? NQ : 256;
  const int RL = 256 / NQc;
  const int tid = threadIdx.x;
  const int ql = tid % NQc, rl = tid / NQc;
  const int q = blockIdx.x * NQc + ql;
  const bool active = (rl < RL) && (q < NQ);
  float4 acc[MAXT][CS];
#pragma unroll
  for (int j = 0; j < MAXT; ++j)
#pragma unroll
    for (int c = 0; c < CS; ++c) acc[j][c] = make_float4(0.f, 0.f, 0.f, 0.f);

  const size_t rows = (size_t)a.B * a.M;
  const size_t r_lo = (size_t)blockIdx.y * a.rows_per_chunk;
  const size_t r_hi = r_lo + a.rows_per_chunk < rows ? r_lo + a.rows_per_chunk : rows;
  if (active) {
    for (size_t r = r_lo + rl; r < r_hi; r += RL) {
      const int b = (int)(r / a.M), m = (int)(r % a.M);
      if (SMALL_IS_IN) {
        const float4 g = *reinterpret_cast<const float4*>(a.dy + r * a.Cout + 4 * q);
#pragma unroll
        for (int j = 0; j < MAXT; ++j) {
          if (j >= a.ntaps) break;
          const int t = a.in_stride * m + a.off[j];
          if (t < 0 || t >= a.Lin) continue;
#pragma unroll
          for (int c = 0; c < CS; ++c) {
            const float xv = a.x[((size_t)b * a.Lin + t) * CS + c];
            acc[j][c].x = fmaf(xv, g.x, acc[j][c].x); acc[j][c].y = fmaf(xv, g.y, acc[j][c].y);
            acc[j][c].z = fmaf(xv, g.z, acc[j][c].z); acc[j][c].w = fmaf(xv, g.w, acc[j][c].w);
          }
        }
      } else {
        float gs[CS];
#pragma unroll
        for (int c = 0; c < CS; ++c) gs[c] = a.dy[r * CS + c];
#pragma unroll
        for (int j = 0; j < MAXT; ++j) {
          if (j >= a.ntaps) break;
          const int t = a.in_stride * m + a.off[j];
          if (t < 0 || t >= a.Lin) continue;
          const float4 xv = *reinterpret_cast<const float4*>(a.x + ((size_t)b * a.Lin + t) * a.Cin + 4 * q);
#pragma unroll
          for (int c = 0; c < CS; ++c) {
            acc[j][c].x = fmaf(xv.x, gs[c], acc[j][c].x); acc[j][c].y = fmaf(xv.y, gs[c], acc[j][c].y);
            acc[j][c].z = fmaf(xv.z, gs[c], acc[j][c].z); acc[j][c].w = fmaf(xv.w, gs[c], acc[j][c].w);
          }
        }
      }
    }
  }
  // reduce the RL row lanes through LDS (one (j,c) pair at a time keeps LDS at 4 KiB)
  __shared__ float4 red[256];
#pragma unroll
  for (int j = 0; j < MAXT; ++j) {
    if (j >= a.ntaps) break;
#pragma unroll
    for (int c = 0; c < CS; ++c) {
      red[tid] = acc[j][c];
      __syncthreads();
      if (rl == 0 && q < NQ) {
        float4 s = red[ql];
        for (int k = 1; k < RL; ++k) {
          const float4 v = red[k * NQc + ql];
          s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        float* pc = a.part + (size_t)blockIdx.y * a.ntaps * a.Cin * a.Cout;
        if (SMALL_IS_IN) {
          *reinterpret_cast<float4*>(pc + ((size_t)j * CS + c) * a.Cout + 4 * q) = s;      // dw[j][c][n]
        } else {
          float* d = pc + ((size_t)j * a.Cin + 4 * q) * CS + c;                            // dw[j][n=cin][c=cout]
          d[0] = s.x; d[CS] = s.y; d[2 * CS] = s.z; d[3 * CS] = s.w;
        }
      }
      __syncthreads();
    }
  }
}

// Small Cout, unit input stride: the same partial sums with the loop turned over INPUT rows, so every x row is loaded once
// (the loop above loads it ntaps times) and the ntaps dy scalars it meets are L2-resident broadcasts; U rows in flight per
// thread.  Chunks are ranges of flattened input rows (b, t); the partial-slab layout and the final reduce are unchanged.
template <int CS>
__global__ __launch_bounds__(256) void wgrad_smallcout_s1_kernel(WgradSmallArgs a) {
  constexpr int MAXT = 5, U = 4;
  const int NQ = a.Cin >> 2;
  const int NQc = NQ < 256 ? NQ : 256;
  const int RL = 256 / NQc;
  const int tid = threadIdx.x;
  const int ql = tid % NQc, rl = tid / NQc;
  const int q = blockIdx.x * NQc + ql;
  const bool active = (rl < RL) && (q < NQ);
  float4 acc[MAXT][CS];
#pragma unroll
  for (int j = 0; j < MAXT; ++j)
#pragma unroll
    for (int c = 0; c < CS; ++c) acc[j][c] = make_float4(0.f, 0.f, 0.f, 0.f);
  const size_t rows = (size_t)a.B * a.Lin;
  const size_t r_lo = (size_t)blockIdx.y * a.rows_per_chunk;
  const size_t r_hi = r_lo + a.rows_per_chunk < rows ? r_lo + a.rows_per_chunk : rows;
  // the dy values every (input row, tap, channel) meets go through an LDS table per TR-row tile (zeros where the tap falls outside
  // [0, M)), so the row loop has only its U batched x loads on the vector-memory path
  constexpr int TR = 128;
  __shared__ float tab[TR][MAXT * CS];
  const int per_row = a.ntaps * CS;
  for (size_t base = r_lo; base < r_hi; base += TR) {
    const int nr = (int)(r_hi - base < (size_t)TR ? r_hi - base : (size_t)TR);
    __syncthreads();
    for (int e = tid; e < nr * per_row; e += 256) {
      const int rr = e / per_row, jc = e % per_row;
      const int j = jc / CS, c = jc % CS;
      const size_t r = base + rr;
      const int b = (int)(r / a.Lin), t = (int)(r % a.Lin);
      const int m = t - a.off[j];
      tab[rr][j * CS + c] = (m >= 0 && m < a.M) ? a.dy[((size_t)b * a.M + m) * CS + c] : 0.f;
    }
    __syncthreads();
    if (active) {
      for (int r0 = rl; r0 < nr; r0 += RL * U) {
        float4 xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int rr = r0 + u * RL;
          xv[u] = rr < nr ? *reinterpret_cast<const float4*>(a.x + (base + rr) * a.Cin + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int rr = r0 + u * RL;
          if (rr >= nr) break;
#pragma unroll
          for (int j = 0; j < MAXT; ++j) {
            if (j >= a.ntaps) break;
#pragma unroll
            for (int c = 0; c < CS; ++c) {
              const float g = tab[rr][j * CS + c];
              acc[j][c].x = fmaf(xv[u].x, g, acc[j][c].x); acc[j][c].y = fmaf(xv[u].y, g, acc[j][c].y);
              acc[j][c].z = fmaf(xv[u].z, g, acc[j][c].z); acc[j][c].w = fmaf(xv[u].w, g, acc[j][c].w);
            }
          }
        }
      }
    }
  }
  __syncthreads();
  __shared__ float4 red[256];
#pragma unroll
  for (int j = 0; j < MAXT; ++j) {
    if (j >= a.ntaps) break;
#pragma unroll
    for (int c = 0; c < CS; ++c) {
      red[tid] = acc[j][c];
      __syncthreads();
      if (rl == 0 && q < NQ) {
        float4 s = red[ql];
        for (int k = 1; k < RL; ++k) {
          const float4 v = red[k * NQc + ql];
          s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        float* d = a.part + (size_t)blockIdx.y * a.ntaps * a.Cin * a.Cout + ((size_t)j * a.Cin + 4 * q) * CS + c;   // dw[j][cin][cout]
        d[0] = s.x; d[CS] = s.y; d[2 * CS] = s.z; d[3 * CS] = s.w;
      }
      __syncthreads();
    }
  }
}

static int wgrad_small_chunks(size_t rows, int CL) {
  const int NQ = CL / 4, NQc = NQ < 256 ? NQ : 256, RL = 256 / NQc;
  const int gx = (NQ + NQc - 1) / NQc;
  int chunks = (1024 + gx - 1) / gx;
  const size_t max_chunks = (rows + (size_t)RL * 8 - 1) / ((size_t)RL * 8);
  if ((size_t)chunks > max_chunks) chunks = (int)max_chunks;
  if (chunks < 1) chunks = 1;
  return chunks;
}

size_t wgrad_small_workspace_bytes(int B, int M, int Cin, int Cout, int ntaps) {
  const int CL = Cin <= 4 ? Cout : Cin;
  return (size_t)wgrad_small_chunks((size_t)B * M, CL) * ntaps * Cin * Cout * sizeof(float);
}

// out[i] = sum_k part[k][i] in a FIXED order (reproducible).  n is small here (ntaps * Cin * Cout with one side <= 4) and the chunk
// count large (up to 1024), so one thread per element would walk 1024 strided loads with 20 blocks on the chip: two levels instead.
// Level 1 (grid.y = G groups): group g folds the chunks k = g, g + G, ... into slab g IN PLACE (no other group touches slab g);
// level 2 sums the G slabs.
constexpr int SUM_GROUPS = 32;

__global__ void sum_partials_l1_kernel(float* __restrict__ part, size_t n, int chunks) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int g = blockIdx.y;
  if (i >= n || g >= chunks) return;
  float s = part[(size_t)g * n + i];
  for (int k = g + SUM_GROUPS; k < chunks; k += SUM_GROUPS) s += part[(size_t)k * n + i];
  part[(size_t)g * n + i] = s;
}

__global__ void sum_partials_kernel(const float* __restrict__ part, float* __restrict__ out, size_t n, int chunks) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = part[i];
  for (int k = 1; k < chunks; ++k) s += part[(size_t)k * n + i];
  out[i] = s;
}

static int sum_partials(float* part, float* out, size_t n, int chunks, hipStream_t s) {
  if (chunks > 2 * SUM_GROUPS) {
    hipLaunchKernelGGL(sum_partials_l1_kernel, dim3(cdiv(n, 256), SUM_GROUPS), dim3(256), 0, s, part, n, chunks);
    int rc = check_launch("sum_partials_l1");
    if (rc) return rc;
    chunks = SUM_GROUPS;
  }
  hipLaunchKernelGGL(sum_partials_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, part, out, n, chunks);
  return check_launch("wgrad_small_reduce");
}

int wgrad_small_dispatch(WgradSmallArgs a, float* dw, size_t ws_bytes, hipStream_t s) {
  const bool small_in = a.Cin <= 4;
  const int CS = small_in ? a.Cin : a.Cout, CL = small_in ? a.Cout : a.Cin;
  if (CS < 1 || CS > 4 || CL % 4 || a.ntaps > 5) {
    set_error("wgrad_small: Cin %d Cout %d ntaps %d unsupported", a.Cin, a.Cout, a.ntaps);
    return GN_EINVAL;
  }
  const size_t rows = (size_t)a.B * a.M;
  const int chunks = wgrad_small_chunks(rows, CL);
  if (ws_bytes < (size_t)chunks * a.ntaps * a.Cin * a.Cout * sizeof(float)) {
    set_error("wgrad_small: workspace too small");
    return GN_EWORKSPACE;
  }
  a.rows_per_chunk = (int)((rows + chunks - 1) / chunks);
  const int NQ = CL / 4, NQc = NQ < 256 ? NQ : 256;
  dim3 grid((NQ + NQc - 1) / NQc, chunks);
  if (!small_in && a.in_stride == 1) {   // every x row loaded once
    a.rows_per_chunk = (int)(((size_t)a.B * a.Lin + chunks - 1) / chunks);
    switch (CS) {
      case 1: hipLaunchKernelGGL(wgrad_smallcout_s1_kernel<1>, grid, dim3(256), 0, s, a); break;
      case 2: hipLaunchKernelGGL(wgrad_smallcout_s1_kernel<2>, grid, dim3(256), 0, s, a); break;
      case 3: hipLaunchKernelGGL(wgrad_smallcout_s1_kernel<3>, grid, dim3(256), 0, s, a); break;
      default: hipLaunchKernelGGL(wgrad_smallcout_s1_kernel<4>, grid, dim3(256), 0, s, a); break;
    }
    int rc1 = check_launch("wgrad_smallcout_s1");
    if (rc1) return rc1;
    const size_t n1 = (size_t)a.ntaps * a.Cin * a.Cout;
    return sum_partials(a.part, dw, n1, chunks, s);
  }
#define GN_WS(CSV)                                                                                            \
  if (small_in) hipLaunchKernelGGL((wgrad_smallcin_tab_kernel<CSV>), grid, dim3(256), 0, s, a);               \
  else hipLaunchKernelGGL((wgrad_small_kernel<CSV, false>), grid, dim3(256), 0, s, a);
  switch (CS) {
    case 1: GN_WS(1); break;
    case 2: GN_WS(2); break;
    case 3: GN_WS(3); break;
    default: GN_WS(4); break;
  }
#undef GN_WS
  int rc = check_launch("wgrad_small");
  if (rc) return rc;
  const size_t n = (size_t)a.ntaps * a.Cin * a.Cout;
  return sum_partials(a.part, dw, n, chunks, s);
}

// ---------------------------------------------------------------------------------------------
// Dense with a tiny output (flatten -> Dense(1)):  y[b,o] = act(bias[o] + sum_i x[b,i] w[i,o]),  out <= 4
// one 1024-thread block per sample; w is L2-resident across blocks.
// ---------------------------------------------------------------------------------------------
template <int OUT>
__global__ __launch_bounds__(1024) void dense_small_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                               float* __restrict__ y, int in, int act, float act_param) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* xb = x + (size_t)b * in;
  float acc[OUT];
#pragma unroll
  for (int o = 0; o < OUT; ++o) acc[o] = 0.f;
  for (int i = 4 * tid; i < in; i += 4096) {
    const float4 xv = *reinterpret_cast<const float4*>(xb + i);
    if (OUT == 1) {
      const float4 wv = *reinterpret_cast<const float4*>(w + i);
      acc[0] = fmaf(xv.x, wv.x, acc[0]); acc[0] = fmaf(xv.y, wv.y, acc[0]);
      acc[0] = fmaf(xv.z, wv.z, acc[0]); acc[0] = fmaf(xv.w, wv.w, acc[0]);
    } else {
      const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int o = 0; o < OUT; ++o) acc[o] = fmaf(xs[e], w[(size_t)(i + e) * OUT + o], acc[o]);
    }
  }
  __shared__ float red[16][OUT];
#pragma unroll
  for (int o = 0; o < OUT; ++o) {
    float v = acc[o];
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) v += __shfl_xor(v, sft, 64);
    if ((tid & 63) == 0) red[tid >> 6][o] = v;
  }
  __syncthreads();
  if (tid < OUT) {
    float v = 0.f;
    for (int k = 0; k < 16; ++k) v += red[k][tid];
    y[(size_t)b * OUT + tid] = act_apply(v + (bias ? bias[tid] : 0.f), act, act_param);
  }
}

int dense_small_fwd(const float* x, const float* w, const float* bias, float* y, int B, int in, int out, int act, float p, hipStream_t s) {
  if (out < 1 || out > 4 || in % 4) {
    set_error("dense_small_fwd: out %d (1..4) / in %d (%%4) unsupported", out, in);
    return GN_EINVAL;
  }
  if (B == 0) return GN_OK;
  switch (out) {
    case 1: hipLaunchKernelGGL(dense_small_fwd_kernel<1>, dim3(B), dim3(1024), 0, s, x, w, bias, y, in, act, p); break;
    case 2: hipLaunchKernelGGL(dense_small_fwd_kernel<2>, dim3(B), dim3(1024), 0, s, x, w, bias, y, in, act, p); break;
    case 3: hipLaunchKernelGGL(dense_small_fwd_kernel<3>, dim3(B), dim3(1024), 0, s, x, w, bias, y, in, act, p); break;
    default: hipLaunchKernelGGL(dense_small_fwd_kernel<4>, dim3(B), dim3(1024), 0, s, x, w, bias, y, in, act, p); break;
  }
  return check_launch("dense_small_fwd");
}

// dw[i,o] = sum_b x[b,i] dy[b,o];  dx[b,i] = sum_o dy[b,o] w[i,o];  one thread per 4 input features, loop over b.
template <int OUT>
__global__ __launch_bounds__(256) void dense_small_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ dy,
                                                              float* __restrict__ dx, float* __restrict__ dw, int B, int in, int gact, float gparam,
                                                              const uint8_t* __restrict__ gmask, float gscale) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= (size_t)in) return;
  float wv[4][OUT], acc[4][OUT];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int o = 0; o < OUT; ++o) {
      wv[e][o] = w[(i + e) * OUT + o];
      acc[e][o] = 0.f;
    }
  // U rows of x (and their masks) are loaded before any of them is used: with one load per iteration and 2 blocks per CU the
  // loop ran at the latency of a single 16-byte load per wave (1.7-2.5 TB/s); the sums stay in b order
  constexpr int U = 8;
  for (int b0 = 0; b0 < B; b0 += U) {
    float4 xq[U];
    uchar4 mq[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int bb = min(b0 + u, B - 1);
      xq[u] = *reinterpret_cast<const float4*>(x + (size_t)bb * in + i);
      mq[u] = (dx && gmask) ? *reinterpret_cast<const uchar4*>(gmask + (size_t)bb * in + i) : make_uchar4(1, 1, 1, 1);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
    const int b = b0 + u;
    if (b >= B) break;
    const float4 xv = xq[u];
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
    float g[OUT];
#pragma unroll
    for (int o = 0; o < OUT; ++o) g[o] = dy[(size_t)b * OUT + o];
    float d[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int o = 0; o < OUT; ++o) {
        acc[e][o] = fmaf(xs[e], g[o], acc[e][o]);
        d[e] = fmaf(g[o], wv[e][o], d[e]);
      }
    if (dx) {
      if (gact != GN_ACT_LINEAR || gmask) {       // x IS the producer's output: fuse its [activation -> dropout] backward here
        const uint8_t k[4] = {mq[u].x, mq[u].y, mq[u].z, mq[u].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = k[e] ? d[e] * gscale * act_grad_from_y(xs[e] / gscale, gact, gparam) : 0.f;
      }
      *reinterpret_cast<float4*>(dx + (size_t)b * in + i) = make_float4(d[0], d[1], d[2], d[3]);
    }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int o = 0; o < OUT; ++o) dw[(i + e) * OUT + o] = acc[e][o];
}

__global__ void colsum_small_kernel(const float* __restrict__ dy, float* __restrict__ db, int B, int out) {
  const int o = threadIdx.x;
  if (o >= out) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += dy[(size_t)b * out + o];
  db[o] = s;
}

int dense_small_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int B, int in, int out, hipStream_t s, int gact,
                    float gparam, const uint8_t* gmask, float gscale) {
  if (out < 1 || out > 4 || in % 4) {
    set_error("dense_small_bwd: out %d (1..4) / in %d (%%4) unsupported", out, in);
    return GN_EINVAL;
  }
  const unsigned grid = cdiv((size_t)in / 4, 256);
  switch (out) {
    case 1: hipLaunchKernelGGL(dense_small_bwd_kernel<1>, dim3(grid), dim3(256), 0, s, x, w, dy, dx, dw, B, in, gact, gparam, gmask, gscale); break;
    case 2: hipLaunchKernelGGL(dense_small_bwd_kernel<2>, dim3(grid), dim3(256), 0, s, x, w, dy, dx, dw, B, in, gact, gparam, gmask, gscale); break;
    case 3: hipLaunchKernelGGL(dense_small_bwd_kernel<3>, dim3(grid), dim3(256), 0, s, x, w, dy, dx, dw, B, in, gact, gparam, gmask, gscale); break;
    default: hipLaunchKernelGGL(dense_small_bwd_kernel<4>, dim3(grid), dim3(256), 0, s, x, w, dy, dx, dw, B, in, gact, gparam, gmask, gscale); break;
  }
  int rc = check_launch("dense_small_bwd");
  if (rc) return rc;
  if (db) {
    hipLaunchKernelGGL(colsum_small_kernel, dim3(1), dim3(64), 0, s, dy, db, B, out);
    rc = check_launch("colsum_small");
  }
  return rc;
}

}  // namespace gn
