// Implicit-GEMM 1-D convolution on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, = fmaf chain).
//
// One kernel serves Conv1D forward, its data gradient (per output phase for stride 2) and Dense-as-1-tap-conv:
//     y[b, os*m + o0, n] = act(bias[n] + sum_j sum_c x[b, is*m + off[j], c] * w[widx[j], c, n]),   rows outside [0,Lin) read 0
// GEMM view: M = (b, m), N = Cout, K = (tap, Cin).  Channels-last makes the im2col operand a set of SHIFTED VIEWS of
// one input slab: per Cin-chunk the block stages [is*(TM-1)+span+1 rows] x [KC channels] once into LDS and reads it
// ntaps times, so A-side global traffic is 1/ntaps of a materialised im2col.
//
// Replaces the TF kernels behind bbhMahoGANy.py:250-292 (generator Conv1D), :362-394 (point-estimator Conv1D) and
// :439,:447 (discriminator Conv2D after the width-2 fold).
//
// Tile: block = WAVES_M x WAVES_N waves, each wave WM x WN MFMA tiles of 32x32 -> TM x TN outputs per block.
// LDS: slab rows padded to KC+1 words (stride-17 -> conflict-free ds_read_b32 across the 32 M-lanes); for in_stride 2
// the slab is de-interleaved by row parity so the lane stride stays KC+1.  Weight tile [tap][KC][TN], N contiguous.
#include <stdlib.h>
#include "common.h"

namespace gn {

typedef float f32x16 __attribute__((ext_vector_type(16)));



// WGLDS: the weight tile goes global -> LDS directly (global_load_lds_dwordx4, no VGPR round trip, no ds_write); legal when the
// tile has no ragged edge (Cout % TN == 0, Cin % KC == 0) because the LDS image [tap][KC][TN] is exactly lane-linear per wave.
template <int WM, int WN, int WAVES_M, int WAVES_N, int KC, int NTAPS, bool WGLDS>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, 2) void conv_mfma_kernel(ConvArgs a, int m_tiles, int n_tiles) {
  constexpr int TM = WAVES_M * WM * 32;
  constexpr int TN = WAVES_N * WN * 32;
  constexpr int NT = 64 * WAVES_M * WAVES_N;
  constexpr int RS = KC + 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int i32 = lane & 31, h = lane >> 5;

  // n fastest: blocks b and b+8 share an XCD (round-robin dispatch), so with n_tiles | 8 or 8 | n_tiles each XCD keeps
  // re-reading the same weight panel from its own L2.  Pure speed; nothing depends on placement.
  const int bid = blockIdx.x;
  const int n_tile = bid % n_tiles;
  const int rest = bid / n_tiles;
  const int m_tile = rest % m_tiles;
  const int b = rest / m_tiles;
  const int m0 = m_tile * TM, n0 = n_tile * TN;

  const int is = a.t.in_stride;
  constexpr int ntaps = NTAPS;   // compile-time: the tap loop is fully unrolled so LDS reads pipeline across taps
  int minoff = a.t.off[0], maxoff = a.t.off[0];
#pragma unroll
  for (int j = 1; j < ntaps; ++j) {
    minoff = min(minoff, a.t.off[j]);
    maxoff = max(maxoff, a.t.off[j]);
  }
  const int R = is * (TM - 1) + (maxoff - minoff) + 1;
  const int Rper = (R + is - 1) / is;
  const int slab_floats = (is * Rper * RS + 3) & ~3;
  const int buf_floats = slab_floats + ntaps * KC * TN;       // one stage: [slab | weights]; two stages in LDS

  f32x16 acc[WM][WN];
#pragma unroll
  for (int mt = 0; mt < WM; ++mt)
#pragma unroll
    for (int nt = 0; nt < WN; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  const int t_base = is * m0 + minoff;
  const float* xb = a.x + (size_t)b * a.Lin * a.Cin;

  // Software pipeline, ONE barrier per K-chunk: registers hold chunk c+1 (its global loads were issued a whole MFMA block
  // earlier); at the top of chunk c they are written to the OTHER LDS stage, the loads of chunk c+2 are issued into the same
  // registers, then the MFMA block of chunk c runs from the current stage.  LDS writes and global loads execute beside the
  // matrix pipe (separate issue ports, 64-cycle MFMAs leave the slots free).  S_ITEMS / W_ITEMS float4 per thread, static
  // indexing so they stay in VGPRs.
  constexpr int S_ITEMS = ((2 * (TM - 1) + 5) * (KC / 4) + NT - 1) / NT;
  constexpr int W_ITEMS = (NTAPS * KC * (TN / 4) + NT - 1) / NT;
  float4 sreg[S_ITEMS], wreg[W_ITEMS];
  const int s_count = R * (KC / 4);
  constexpr int w_count = NTAPS * KC * (TN / 4);

  // Per-thread staging addresses are chunk-invariant: computed once here, so the per-chunk staging code is one 64-bit add,
  // one select and one load (or LDS store) per item.  Items that are never valid point at the tensor base with limit 0;
  // an item is live for chunk c0 iff c0 < lim (only the last, partial chunk of a Cin that is not a multiple of KC differs).
  const float* sp[S_ITEMS];
  int slim[S_ITEMS], sl[S_ITEMS];
#pragma unroll
  for (int it = 0; it < S_ITEMS; ++it) {
    const int id = tid + it * NT;
    const int r = id / (KC / 4), c4 = id % (KC / 4);
    const int t = t_base + r;
    const bool ok = id < s_count && t >= 0 && t < a.Lin && 4 * c4 < a.Cin;
    sp[it] = ok ? xb + (size_t)t * a.Cin + 4 * c4 : a.x;
    slim[it] = ok ? a.Cin - 4 * c4 : 0;
    const int lr = (is == 1) ? r : ((r & 1) * Rper + (r >> 1));
    sl[it] = id < s_count ? lr * RS + 4 * c4 : -1;
  }
  const float* wp[W_ITEMS];
  int wlim[W_ITEMS];
#pragma unroll
  for (int it = 0; it < W_ITEMS; ++it) {
    const int id = tid + it * NT;
    const int n4 = id % (TN / 4);
    const int kk = (id / (TN / 4)) % KC;
    const int j = id / ((TN / 4) * KC);
    const int n = n0 + 4 * n4;
    const bool ok = id < w_count && kk < a.Cin && n < a.Cout;
    wp[it] = ok ? a.w + ((size_t)a.t.widx[j < NTAPS ? j : 0] * a.Cin + kk) * a.Cout + n : a.w;
    wlim[it] = ok ? a.Cin - kk : 0;
  }

  // Validity is applied when the registers are WRITTEN TO LDS, not when they are loaded: a select on freshly loaded data would
  // make the compiler wait for the prefetch immediately.  smask / wmask remember which items of the in-flight chunk are live.
  unsigned smask = 0, wmask = 0;
  auto load_s = [&](int it, int c0) {
    const bool ok = c0 < slim[it];
    smask = (smask & ~(1u << it)) | ((ok ? 1u : 0u) << it);
    sreg[it] = *reinterpret_cast<const float4*>(sp[it] + (ok ? c0 : 0));   // always an in-bounds global address; zeroed at STORE time
  };
  auto load_w = [&](int it, int c0) {
    const bool ok = c0 < wlim[it];
    wmask = (wmask & ~(1u << it)) | ((ok ? 1u : 0u) << it);
    wreg[it] = *reinterpret_cast<const float4*>(wp[it] + (ok ? (size_t)c0 * a.Cout : (size_t)0));
  };
  auto store_s = [&](int it, float* stage) {
    if (sl[it] >= 0) {
      float* d = stage + sl[it];
      const bool ok = (smask >> it) & 1u;
      d[0] = ok ? sreg[it].x : 0.f; d[1] = ok ? sreg[it].y : 0.f; d[2] = ok ? sreg[it].z : 0.f; d[3] = ok ? sreg[it].w : 0.f;
    }
  };
  auto store_w = [&](int it, float* stage) {
    const int id = tid + it * NT;
    if (id < w_count) {                                                        // [tap][KC][TN] is exactly id order
      const bool ok = (wmask >> it) & 1u;
      *reinterpret_cast<float4*>(stage + slab_floats + id * 4) = make_float4(ok ? wreg[it].x : 0.f, ok ? wreg[it].y : 0.f, ok ? wreg[it].z : 0.f, ok ? wreg[it].w : 0.f);
    }
  };

  // LDS-DMA of one weight item: 64 lanes x 16 B land contiguously at the wave-uniform LDS address of the wave's first lane
  auto glds_w = [&](int it, int c0, float* stage) {
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const float* src = wp[it] + (size_t)c0 * a.Cout;
    float* dst = stage + slab_floats + (it * NT + (tid & ~63)) * 4;
    gn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
  };

  const int n_chunks = (a.Cin + KC - 1) / KC;
#pragma unroll
  for (int it = 0; it < S_ITEMS; ++it) { load_s(it, 0); }
  if (WGLDS) {
#pragma unroll
    for (int it = 0; it < W_ITEMS; ++it) { glds_w(it, 0, smem); }
  } else {
#pragma unroll
    for (int it = 0; it < W_ITEMS; ++it) { load_w(it, 0); }
  }
#pragma unroll
  for (int it = 0; it < S_ITEMS; ++it) { store_s(it, smem); }
  if (!WGLDS) {
#pragma unroll
    for (int it = 0; it < W_ITEMS; ++it) { store_w(it, smem); }
  }
  {
    const int c1 = (n_chunks > 1 ? 1 : 0) * KC;
#pragma unroll
    for (int it = 0; it < S_ITEMS; ++it) { load_s(it, c1); }
    if (!WGLDS) {
#pragma unroll
      for (int it = 0; it < W_ITEMS; ++it) { load_w(it, c1); }
    }
  }
  __syncthreads();       // with WGLDS the compiler drains the LDS-DMA (vmcnt(0)) in front of this barrier

  // Main loop, ONE barrier per K-chunk and NO separate staging phase: while the MFMAs of chunk ch run from stage `cur`, each
  // staging item (a float4 of chunk ch+1 held in registers since the previous iteration) is written to stage `nxt` and its
  // register is immediately re-loaded with chunk ch+2, one item per slot, spread evenly between the MFMA groups of the unrolled
  // tap loop -- so co-resident blocks that run in lockstep never all sit in a staging phase at the same time.  The code is
  // branch-free: past the end the loads are clamped to the last chunk and the stores land in the stage nobody reads again.
  constexpr int SLOTS = NTAPS * (KC / 2);
  constexpr int ITEMS = S_ITEMS + (WGLDS ? 0 : W_ITEMS);
  for (int ch = 0; ch < n_chunks; ++ch) {
    float* cur = smem + (ch & 1) * buf_floats;
    float* nxt = smem + ((ch + 1) & 1) * buf_floats;
    const int c2 = min(ch + 2, n_chunks - 1) * KC;
    if (WGLDS) {          // weights of chunk ch+1 fly global -> LDS stage `nxt` during this chunk's MFMAs; retired by the barrier below
      const int c1 = min(ch + 1, n_chunks - 1) * KC;
#pragma unroll
      for (int it = 0; it < W_ITEMS; ++it) { glds_w(it, c1, nxt); }
    }
    const float* slab = cur;
    const float* wl = cur + slab_floats;

    // operand registers are double-buffered: the ds_reads of MFMA group g+1 are issued before the MFMAs of group g, so a
    // wave's LDS latency hides under its own 256 cycles of matrix work (groups run across the tap boundary)
    auto read_ops = [&](int g, float (&av)[WM], float (&bv)[WN]) {
      const int j = g / (KC / 2), q = g % (KC / 2);
      const int d = a.t.off[j] - minoff;
      const int rowbase = (is == 1) ? d : ((d & 1) * Rper + (d >> 1));
      const float* ap = slab + (rowbase + wm * WM * 32 + i32) * RS + h;
      const float* bp = wl + (j * KC + h) * TN + wn * WN * 32 + i32;
#pragma unroll
      for (int mt = 0; mt < WM; ++mt) av[mt] = ap[mt * 32 * RS + 2 * q];
#pragma unroll
      for (int nt = 0; nt < WN; ++nt) bv[nt] = bp[2 * q * TN + nt * 32];
    };
    float av0[WM], bv0[WN], av1[WM], bv1[WN];
    read_ops(0, av0, bv0);
#pragma unroll
    for (int g = 0; g < SLOTS; ++g) {
#pragma unroll
      for (int i = 0; i < ITEMS; ++i) {
        if ((i * SLOTS) / ITEMS == g) {             // compile-time after unrolling: item i is staged in this slot
          if (i < S_ITEMS) { store_s(i, nxt); load_s(i, c2); }
          else { store_w(i - S_ITEMS, nxt); load_w(i - S_ITEMS, c2); }
        }
      }
      if (g % 2 == 0) {
        if (g + 1 < SLOTS) read_ops(g + 1, av1, bv1);
#pragma unroll
        for (int mt = 0; mt < WM; ++mt)
#pragma unroll
          for (int nt = 0; nt < WN; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[mt], bv0[nt], acc[mt][nt], 0, 0, 0);
      } else {
        if (g + 1 < SLOTS) read_ops(g + 1, av0, bv0);
#pragma unroll
        for (int mt = 0; mt < WM; ++mt)
#pragma unroll
          for (int nt = 0; nt < WN; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[mt], bv1[nt], acc[mt][nt], 0, 0, 0);
      }
    }
    __syncthreads();   // (a) stage `cur` may be overwritten during ch+1; (b) the stores into `nxt` are visible to every wave
  }

  // ---- epilogue: C/D layout of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* yb = a.y + (size_t)b * a.Ly * a.Cout;
  const uint8_t* mb = a.mask ? a.mask + (size_t)b * a.Ly * a.Cout : nullptr;
  const float* gyb = a.gy ? a.gy + (size_t)b * a.Ly * a.Cout : nullptr;
  const uint8_t* gmb = a.gmask ? a.gmask + (size_t)b * a.Ly * a.Cout : nullptr;
#pragma unroll
  for (int nt = 0; nt < WN; ++nt) {
    const int n = n0 + wn * WN * 32 + nt * 32 + i32;
    if (n >= a.Cout) continue;
    const float bias = a.bias ? a.bias[n] : 0.f;
#pragma unroll
    for (int mt = 0; mt < WM; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int m = m0 + wm * WM * 32 + mt * 32 + row;
        if (m < a.M) {
          const size_t o = (size_t)(a.t.out_stride * m + a.t.out_off) * a.Cout + n;
          float v = act_apply(acc[mt][nt][r] + bias, a.act, a.act_param);
          if (mb) v = mb[o] ? v * a.keep_scale : 0.f;
          if (gyb) {                                    // fused backward of the producer's [activation -> dropout]
            const float gv = gyb[o];
            if (gmb) v = gmb[o] ? v * a.gscale * act_grad_from_y(gv / a.gscale, a.gact, a.gparam) : 0.f;
            else v *= act_grad_from_y(gv, a.gact, a.gparam);
          }
          yb[o] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// All-DMA variant for tiles without ragged channel edges (Cin % KC == 0, Cout % TN == 0): BOTH operands go global -> LDS by
// LDS-DMA, no staging registers, no ds_write, no per-item validity logic.
//   * weights: global_load_lds_dwordx4, LDS image [tap][KC][TN] is lane-linear per wave;
//   * input slab: buffer_load_dwordx4 ... lds through a per-batch-element buffer descriptor whose range check returns 0 for rows
//     outside [0, Lin) -- the zero padding of the convolution comes from the hardware bounds check.  Rows are UNPADDED (KC floats):
//     the A-operand ds_read_b32 is then 4-way bank-conflicted (stride 8 words), 16 LDS cycles per MFMA group per wave, which the
//     64-cycle fp32 MFMAs hide (LDS < 35 % busy at 3 blocks/CU).
// Same tile, same one-barrier double-stage pipeline, same epilogue as conv_mfma_kernel.
// ---------------------------------------------------------------------------------------------
template <int WM, int WN, int WAVES_M, int WAVES_N, int KC, int NTAPS>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, 2) void conv_mfma_dma_kernel(ConvArgs a, int m_tiles, int n_tiles) {
#if defined(__HIP_DEVICE_COMPILE__)   // the buffer-descriptor type does not exist in the host pass (the stub needs no body)
  constexpr int TM = WAVES_M * WM * 32;
  constexpr int TN = WAVES_N * WN * 32;
  constexpr int NT = 64 * WAVES_M * WAVES_N;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int i32 = lane & 31, h = lane >> 5;
  const int bid = blockIdx.x;
  const int n_tile = bid % n_tiles;
  const int rest = bid / n_tiles;
  const int m_tile = rest % m_tiles;
  const int b = rest / m_tiles;
  const int m0 = m_tile * TM, n0 = n_tile * TN;

  const int is = a.t.in_stride;
  int minoff = a.t.off[0], maxoff = a.t.off[0];
#pragma unroll
  for (int j = 1; j < NTAPS; ++j) {
    minoff = min(minoff, a.t.off[j]);
    maxoff = max(maxoff, a.t.off[j]);
  }
  const int R = is * (TM - 1) + (maxoff - minoff) + 1;
  const int Rper = (R + is - 1) / is;
  const int slab_floats = is * Rper * KC;
  const int buf_floats = slab_floats + NTAPS * KC * TN;

  f32x16 acc[WM][WN];
#pragma unroll
  for (int mt = 0; mt < WM; ++mt)
#pragma unroll
    for (int nt = 0; nt < WN; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  const int t_base = is * m0 + minoff;
  // descriptor inputs through readfirstlane: provably wave-uniform, so no waterfall loop around the buffer_load ... lds (see the
  // pipelined kernel below)
  const uintptr_t xbp = (uintptr_t)(a.x + (size_t)b * a.Lin * a.Cin);
  const unsigned xb_lo = __builtin_amdgcn_readfirstlane((unsigned)xbp), xb_hi = __builtin_amdgcn_readfirstlane((unsigned)(xbp >> 32));
  const int xbytes = __builtin_amdgcn_readfirstlane(a.Lin * a.Cin * 4);
  const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)xb_hi << 32) | xb_lo), 0, xbytes, 0x00020000);

  constexpr int S_ITEMS = ((2 * (TM - 1) + 6) * (KC / 4) + NT - 1) / NT;
  constexpr int W_TOTAL = NTAPS * KC * (TN / 4);         // weight granules of one stage (a multiple of 64: whole waves)
  constexpr int W_ITEMS = (W_TOTAL + NT - 1) / NT;
  const int s_count = is * Rper * (KC / 4);              // granules of one slab stage, in LDS order
  int soff[S_ITEMS];                                      // byte offset of the item's source inside the batch element (chunk 0)
#pragma unroll
  for (int it = 0; it < S_ITEMS; ++it) {
    const int id = tid + it * NT;
    const int lr = id / (KC / 4), c4 = id % (KC / 4);
    const int r = (is == 1) ? lr : (lr < Rper ? 2 * lr : 2 * (lr - Rper) + 1);
    soff[it] = (id < s_count && r < R) ? ((t_base + r) * a.Cin + 4 * c4) * 4 : 0x40000000;   // out of range -> the descriptor returns 0
  }
  const float* wp[W_ITEMS];
#pragma unroll
  for (int it = 0; it < W_ITEMS; ++it) {
    const int id = min(tid + it * NT, W_TOTAL - 1);      // (the last item may cover only the first waves; the rest never issue it)
    const int n4 = id % (TN / 4);
    const int kk = (id / (TN / 4)) % KC;
    const int j = id / ((TN / 4) * KC);
    wp[it] = a.w + ((size_t)a.t.widx[j] * a.Cin + kk) * a.Cout + n0 + 4 * n4;
  }
  auto dma_chunk = [&](int c0, float* stage) {
#pragma unroll
    for (int it = 0; it < S_ITEMS; ++it) {
      if (tid + it * NT < s_count)                        // lanes past the slab end stay masked off (EXEC): they would land in the weight tile
        gn_buffer_load_lds(xsrd, (lptr_t)(stage + (it * NT + (tid & ~63)) * 4), 16, soff[it] + c0 * 4, 0, 0, 0);
    }
#pragma unroll
    for (int it = 0; it < W_ITEMS; ++it)
      if ((it + 1) * NT <= W_TOTAL || (tid & ~63) + it * NT < W_TOTAL)      // wave-uniform: whole waves issue or skip
        gn_global_load_lds((gptr_t)(wp[it] + (size_t)c0 * a.Cout), (lptr_t)(stage + slab_floats + (it * NT + (tid & ~63)) * 4), 16, 0, 0);
  };

  const int n_chunks = a.Cin / KC;
  dma_chunk(0, smem);
  __syncthreads();                                         // drains the LDS-DMA (vmcnt(0)) in front of the barrier

  constexpr int SLOTS = NTAPS * (KC / 2);
  for (int ch = 0; ch < n_chunks; ++ch) {
    const float* slab = smem + (ch & 1) * buf_floats;
    const float* wl = slab + slab_floats;
    dma_chunk(min(ch + 1, n_chunks - 1) * KC, smem + ((ch + 1) & 1) * buf_floats);   // chunk ch+1 flies during this chunk's MFMAs

    auto read_ops = [&](int g, float (&av)[WM], float (&bv)[WN]) {
      const int j = g / (KC / 2), q = g % (KC / 2);
      const int d = a.t.off[j] - minoff;
      const int rowbase = (is == 1) ? d : ((d & 1) * Rper + (d >> 1));
      const float* ap = slab + (rowbase + wm * WM * 32 + i32) * KC + h;
      const float* bp = wl + (j * KC + h) * TN + wn * WN * 32 + i32;
#pragma unroll
      for (int mt = 0; mt < WM; ++mt) av[mt] = ap[mt * 32 * KC + 2 * q];
#pragma unroll
      for (int nt = 0; nt < WN; ++nt) bv[nt] = bp[2 * q * TN + nt * 32];
    };
    float av0[WM], bv0[WN], av1[WM], bv1[WN];
    read_ops(0, av0, bv0);
#pragma unroll
    for (int g = 0; g < SLOTS; ++g) {
      if (g % 2 == 0) {
        if (g + 1 < SLOTS) read_ops(g + 1, av1, bv1);
#pragma unroll
        for (int mt = 0; mt < WM; ++mt)
#pragma unroll
          for (int nt = 0; nt < WN; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[mt], bv0[nt], acc[mt][nt], 0, 0, 0);
      } else {
        if (g + 1 < SLOTS) read_ops(g + 1, av0, bv0);
#pragma unroll
        for (int mt = 0; mt < WM; ++mt)
#pragma unroll
          for (int nt = 0; nt < WN; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[mt], bv1[nt], acc[mt][nt], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  float* yb = a.y + (size_t)b * a.Ly * a.Cout;
  const uint8_t* mb = a.mask ? a.mask + (size_t)b * a.Ly * a.Cout : nullptr;
  const float* gyb = a.gy ? a.gy + (size_t)b * a.Ly * a.Cout : nullptr;
  const uint8_t* gmb = a.gmask ? a.gmask + (size_t)b * a.Ly * a.Cout : nullptr;
#pragma unroll
  for (int nt = 0; nt < WN; ++nt) {
    const int n = n0 + wn * WN * 32 + nt * 32 + i32;
    const float bias = a.bias ? a.bias[n] : 0.f;
#pragma unroll
    for (int mt = 0; mt < WM; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int m = m0 + wm * WM * 32 + mt * 32 + row;
        if (m < a.M) {
          const size_t o = (size_t)(a.t.out_stride * m + a.t.out_off) * a.Cout + n;
          float v = act_apply(acc[mt][nt][r] + bias, a.act, a.act_param);
          if (mb) v = mb[o] ? v * a.keep_scale : 0.f;
          if (gyb) {
            const float gv = gyb[o];
            if (gmb) v = gmb[o] ? v * a.gscale * act_grad_from_y(gv / a.gscale, a.gact, a.gparam) : 0.f;
            else v *= act_grad_from_y(gv, a.gact, a.gparam);
          }
          yb[o] = v;
        }
      }
    }
  }
#endif
}

template <int WM, int WN, int WAVES_M, int WAVES_N, int KC, int NTAPS>
static int launch_conv_dma(const ConvArgs& a, hipStream_t s) {
  constexpr int TM = WAVES_M * WM * 32, TN = WAVES_N * WN * 32;
  const int is = a.t.in_stride;
  int minoff = a.t.off[0], maxoff = a.t.off[0];
  for (int j = 1; j < a.t.ntaps; ++j) {
    minoff = std::min(minoff, a.t.off[j]);
    maxoff = std::max(maxoff, a.t.off[j]);
  }
  const int R = is * (TM - 1) + (maxoff - minoff) + 1;
  const int Rper = (R + is - 1) / is;
  const size_t lds = 2 * sizeof(float) * ((size_t)is * Rper * KC + (size_t)NTAPS * KC * TN);
  if (lds > 64 * 1024) {
    static unsigned long long lds_done = 0;
    allow_big_lds((const void*)conv_mfma_dma_kernel<WM, WN, WAVES_M, WAVES_N, KC, NTAPS>, &lds_done);
  }
  const int m_tiles = (a.M + TM - 1) / TM, n_tiles = a.Cout / TN;
  const size_t blocks = (size_t)m_tiles * n_tiles * a.B;
  if (blocks == 0 || blocks > 0x7fffffffull) {
    set_error("conv_mfma_dma: bad grid %zu", blocks);
    return GN_EINVAL;
  }
  prof_begin(s);
  hipLaunchKernelGGL((conv_mfma_dma_kernel<WM, WN, WAVES_M, WAVES_N, KC, NTAPS>), dim3((unsigned)blocks), dim3(64 * WAVES_M * WAVES_N), lds, s, a, m_tiles, n_tiles);
  prof_end(s, 2.0 * a.B * (double)a.M * a.t.ntaps * a.Cin * a.Cout, 0, 4.0 * ((double)a.B * a.Lin * a.Cin + (double)a.t.ntaps * a.Cin * a.Cout + (double)a.B * a.M * a.Cout));
  return check_launch("conv_mfma_dma");
}

template <int WM, int WN, int WAVES_M, int WAVES_N, int KC, int NTAPS, bool WGLDS>
static int launch_conv_impl(const ConvArgs& a, hipStream_t s) {
  constexpr int TM = WAVES_M * WM * 32, TN = WAVES_N * WN * 32;
  const int is = a.t.in_stride;
  int minoff = a.t.off[0], maxoff = a.t.off[0];
  for (int j = 1; j < a.t.ntaps; ++j) {
    minoff = std::min(minoff, a.t.off[j]);
    maxoff = std::max(maxoff, a.t.off[j]);
  }
  const int R = is * (TM - 1) + (maxoff - minoff) + 1;
  const int Rper = (R + is - 1) / is;
  const int slab_floats = (is * Rper * (KC + 1) + 3) & ~3;
  const size_t lds = 2 * sizeof(float) * ((size_t)slab_floats + (size_t)a.t.ntaps * KC * TN);   // two pipeline stages
  if (lds > 160 * 1024) {
    set_error("conv_mfma: LDS tile %zu B exceeds 160 KiB (ntaps=%d, in_stride=%d)", lds, a.t.ntaps, is);
    return GN_EINVAL;
  }
  if (lds > 64 * 1024) {   // opt in to more than the default 64 KiB of dynamic LDS (once per instantiation)
    static unsigned long long lds_done = 0;
    allow_big_lds((const void*)conv_mfma_kernel<WM, WN, WAVES_M, WAVES_N, KC, NTAPS, WGLDS>, &lds_done);
  }
  const int m_tiles = (a.M + TM - 1) / TM, n_tiles = (a.Cout + TN - 1) / TN;
  const size_t blocks = (size_t)m_tiles * n_tiles * a.B;
  if (blocks == 0 || blocks > 0x7fffffffull) {
    set_error("conv_mfma: bad grid %zu", blocks);
    return GN_EINVAL;
  }
  prof_begin(s);
  hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, WAVES_M, WAVES_N, KC, NTAPS, WGLDS>), dim3((unsigned)blocks), dim3(64 * WAVES_M * WAVES_N), lds, s, a, m_tiles, n_tiles);
  prof_end(s, 2.0 * a.B * (double)a.M * a.t.ntaps * a.Cin * a.Cout, 0, 4.0 * ((double)a.B * a.Lin * a.Cin + (double)a.t.ntaps * a.Cin * a.Cout + (double)a.B * a.M * a.Cout));
  return check_launch("conv_mfma");
}

template <int WM, int WN, int WAVES_M, int WAVES_N, int KC, int NTAPS>
static int launch_conv(const ConvArgs& a, hipStream_t s) {
  constexpr int TN = WAVES_N * WN * 32;
  static const bool no_dma = getenv("GN_CONV_NODMA") != nullptr;       // A/B switch: the register-staged kernel everywhere (tests/test_switches_gpu.py)
  const bool even = (a.Cout % TN == 0) && (a.Cin % KC == 0);                                   // no ragged channel edges
  const bool full = even && ((NTAPS * KC * (TN / 4)) % (64 * WAVES_M * WAVES_N) == 0);       // ... and whole weight items per thread
  if (even && (NTAPS * KC * (TN / 4)) % 64 == 0 && !no_dma && (size_t)a.Lin * a.Cin * 4 < 0x40000000ull) {
    if constexpr (KC == 8 && WM == 2 && WN == 2 && NTAPS >= 2) {
      bool launched = false;
      const int rc = conv_pipe_try(a, WAVES_M == 4, s, &launched);     // conv_pipe.hip: hand-scheduled MFMA block, lean epilogue
      if (launched || rc) return rc;
    }
    return launch_conv_dma<WM, WN, WAVES_M, WAVES_N, KC, NTAPS>(a, s);
  }
  return full ? launch_conv_impl<WM, WN, WAVES_M, WAVES_N, KC, NTAPS, true>(a, s) : launch_conv_impl<WM, WN, WAVES_M, WAVES_N, KC, NTAPS, false>(a, s);
}

// Entry used by the C-ABI wrappers in capi.hip.
int conv_mfma_dispatch(const ConvArgs& a, hipStream_t s) {
  if (a.Cin % 4 || a.Cout % 4) {
    set_error("conv_mfma: Cin (%d) and Cout (%d) must be multiples of 4", a.Cin, a.Cout);
    return GN_EINVAL;
  }
  if (a.t.in_stride != 1 && a.t.in_stride != 2) {
    set_error("conv_mfma: in_stride %d unsupported", a.t.in_stride);
    return GN_EINVAL;
  }
  if (a.t.ntaps < 1 || a.t.ntaps > 5) {
    set_error("conv_mfma: ntaps %d unsupported (1..5)", a.t.ntaps);
    return GN_EINVAL;
  }
  // Tile: 256 x 64 (4 waves stacked in M) or 128 x 128 (2 x 2 waves).  The tall tile stages FEWER bytes per flop (a 10 KiB weight
  // stage, 37 KiB of LDS per block -> 4 blocks/CU against 3; for stride 2, 3 against 2) and wins on the 4-5-tap launches whenever
  // M fills it: measured 142 against 138 TFLOP/s on the dominant layer, 135.5 against 129.5 on the stride-2 forward; the 2-3-tap
  // data-gradient phases (130 against 136) and short sequences (M = 125: half a tile idle) keep the square tile.
  auto fill = [&](int T) { return (double)a.M / ((double)((a.M + T - 1) / T) * T); };
  const bool tall_ok = a.Cout % 64 == 0 && a.t.ntaps >= 4 && fill(256) >= 0.97 * fill(128);
  const bool narrow = a.Cout <= 64 || tall_ok;
  // K-chunk: 8 channels for 2-5 taps (stages of 8-25 KiB -> 3-4 blocks/CU; measured on the stride-2 data-gradient phases of 3 and
  // 2 taps: 134 TFLOP/s against 122 with 16-channel chunks at 2 blocks/CU -- occupancy beats MFMAs-per-barrier); 16 for the
  // single-tap Dense so that a barrier still covers 32 MFMAs per wave
#define GN_CONV(NT_, KC_)                                                   \
  return narrow ? launch_conv<2, 2, 4, 1, KC_, NT_>(a, s) : launch_conv<2, 2, 2, 2, KC_, NT_>(a, s)
  switch (a.t.ntaps) {
    case 1: GN_CONV(1, 16);
    case 2: GN_CONV(2, 8);
    case 3: GN_CONV(3, 8);
    case 4: GN_CONV(4, 8);
    default: GN_CONV(5, 8);
  }
#undef GN_CONV
}

// ---------------------------------------------------------------------------------------------
// Weight gradient: dw[j, c, n] = sum_{b,m} x[b, is*m + off[j], c] * dy[b, m, n]   (GEMM: M = Cin, N = Cout, K = (b, m))
// Block = one (Cin-tile, Cout-tile, K-split); every wave keeps ntaps accumulator tiles (the x slab again serves all
// taps as shifted views).  Partial slabs [split][tap][Cin][Cout] are summed by wgrad_reduce_kernel in a fixed order.
// ---------------------------------------------------------------------------------------------

template <int WAVES_C, int WAVES_N, int WNT, int NTAPS, int KT>
// (220 registers with 5 taps = 2 blocks/CU; forcing 3 with __launch_bounds__(.., 3) spills into the loop: 131 -> 58 TFLOP/s)
__global__ __launch_bounds__(64 * WAVES_C * WAVES_N) void wgrad_mfma_kernel(WgradArgs a) {
  constexpr int TC = WAVES_C * 32, TN = WAVES_N * WNT * 32;   // each wave: 32 input channels x (WNT x 32) output channels x NTAPS taps
  constexpr int NT = 64 * WAVES_C * WAVES_N;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wc = wave / WAVES_N, wn = wave % WAVES_N;
  const int i32 = lane & 31, h = lane >> 5;
  const int c0 = blockIdx.x * TC, n0 = blockIdx.y * TN, split = blockIdx.z;
  const int is = a.in_stride;

  int minoff = a.off[0], maxoff = a.off[0];
#pragma unroll
  for (int j = 1; j < NTAPS; ++j) {
    minoff = min(minoff, a.off[j]);
    maxoff = max(maxoff, a.off[j]);
  }
  const int R = is * (KT - 1) + (maxoff - minoff) + 1;
  float* slab = smem;                                    // [R][TC]
  float* dyl = smem + ((R * TC + 3) & ~3);               // [KT][TN]

  f32x16 acc[NTAPS][WNT];
#pragma unroll
  for (int j = 0; j < NTAPS; ++j)
#pragma unroll
    for (int u = 0; u < WNT; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][u][r] = 0.f;

  const int cpb = (a.M + KT - 1) / KT;                     // K-chunks per batch element
  const int c_lo = split * a.chunks_per_split, c_hi = min(a.B * cpb, c_lo + a.chunks_per_split);
  const int n_chunks = max(c_hi - c_lo, 0);

  // register-staged pipeline over the K-chunks (b, m0): loads of chunk i+1 fly during the MFMA block of chunk i
  constexpr int S_ITEMS = ((2 * (KT - 1) + 5) * (TC / 4) + NT - 1) / NT;
  constexpr int D_ITEMS = (KT * (TN / 4) + NT - 1) / NT;
  float4 sreg[S_ITEMS], dreg[D_ITEMS];
  const int s_count = R * (TC / 4);

  auto load_chunk = [&](int ch) {
    const int b = (c_lo + ch) / cpb, m0 = ((c_lo + ch) % cpb) * KT;
    const float* xb = a.x + (size_t)b * a.Lin * a.Cin;
    const float* dyb = a.dy + (size_t)b * a.M * a.Cout;
    const int t_base = is * m0 + minoff;
#pragma unroll
    for (int it = 0; it < S_ITEMS; ++it) {
      const int id = tid + it * NT;
      const int r = id / (TC / 4), c4 = id % (TC / 4);
      const int t = t_base + r, c = c0 + 4 * c4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (id < s_count && t >= 0 && t < a.Lin && c < a.Cin) v = *reinterpret_cast<const float4*>(xb + (size_t)t * a.Cin + c);
      sreg[it] = v;
    }
#pragma unroll
    for (int it = 0; it < D_ITEMS; ++it) {
      const int id = tid + it * NT;
      const int r = id / (TN / 4), n4 = id % (TN / 4);
      const int m = m0 + r, n = n0 + 4 * n4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (id < KT * (TN / 4) && m < a.M && n < a.Cout) v = *reinterpret_cast<const float4*>(dyb + (size_t)m * a.Cout + n);
      dreg[it] = v;
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int it = 0; it < S_ITEMS; ++it) {
      const int id = tid + it * NT;
      if (id < s_count) *reinterpret_cast<float4*>(slab + id * 4) = sreg[it];          // [R][TC] is id order
    }
#pragma unroll
    for (int it = 0; it < D_ITEMS; ++it) {
      const int id = tid + it * NT;
      if (id < KT * (TN / 4)) *reinterpret_cast<float4*>(dyl + id * 4) = dreg[it];     // [KT][TN] is id order
    }
  };

  if (n_chunks > 0) {
    load_chunk(0);
    store_chunk();
  }
  __syncthreads();
  for (int ch = 0; ch < n_chunks; ++ch) {
    const bool has_next = ch + 1 < n_chunks;
    if (has_next) load_chunk(ch + 1);
    const float* bp = dyl + h * TN + wn * WNT * 32 + i32;
#pragma unroll
    for (int q = 0; q < KT / 2; ++q) {
      float bv[WNT];
#pragma unroll
      for (int u = 0; u < WNT; ++u) bv[u] = bp[2 * q * TN + u * 32];
#pragma unroll
      for (int j = 0; j < NTAPS; ++j) {
        const float av = slab[(is * (2 * q + h) + (a.off[j] - minoff)) * TC + wc * 32 + i32];
#pragma unroll
        for (int u = 0; u < WNT; ++u) acc[j][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[u], acc[j][u], 0, 0, 0);
      }
    }
    if (has_next) {
      __syncthreads();
      store_chunk();
      __syncthreads();
    }
  }

#pragma unroll
  for (int u = 0; u < WNT; ++u) {
    const int n = n0 + (wn * WNT + u) * 32 + i32;
    if (n >= a.Cout) continue;
#pragma unroll
    for (int j = 0; j < NTAPS; ++j) {
      float* pj = a.part + ((size_t)split * NTAPS + j) * a.Cin * a.Cout;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = c0 + wc * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (c < a.Cin) pj[(size_t)c * a.Cout + n] = acc[j][u][r];
      }
    }
  }
}

// dw[e] = sum_s part[s][e]  (fixed order: reproducible); float4 over e
__global__ void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, size_t n4, int splits, size_t stride4) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float4* p = reinterpret_cast<const float4*>(part);
  float4 s = p[i];
#pragma unroll 4
  for (int k = 1; k < splits; ++k) {                 // (several splits' loads in flight: the pass is latency-bound; the order of the sum is unchanged)
    const float4 v = p[(size_t)k * stride4 + i];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  reinterpret_cast<float4*>(dw)[i] = s;
}

// K-splits of the weight gradient: about 2048 blocks in all.  A split is a range of K-chunks (32 rows of one batch element, KT in the kernels);
// normally whole batch elements (chunks_per_split a multiple of the chunks per element: the partition of every earlier round).  Only when
// even one split per batch element leaves the chip idle (tiles * B < 256: the small layers at the script's own batch 8) are the elements cut
// into ranges of at least 4 chunks, for about 512 blocks.  (Measured at batch 8, us per launch: G 128 -> 256 90 -> 54, PE q 128 -> 256 87 -> 51;
// cutting layers that already had 256+ blocks was slower -- G 256 -> 512 99 -> 134 -- the extra partial slabs cost more than they buy.)
void wgrad_split_plan(int B, int M, int Cin, int Cout, int TC, int TN, int* splits, int* chunks_per_split) {
  const int cpb = (M + 31) / 32;
  const int tiles = cdiv(Cin, TC) * cdiv(Cout, TN);
  // blocks per launch the K-splits aim at: two full rounds of the chip's 1024 block slots.  Measured (round 3, step in waveforms/s): 2048: 1437,
  // 1024 (one round, half the partial slabs to write and reduce): 1436 -- the kernel loses what the reduce gains; 1536 / 3072: 1429 (ragged rounds)
  constexpr int target_blocks = 2048;
  int s = (target_blocks + tiles - 1) / tiles;
  if (s < 1) s = 1;
  int cps;
  if (s <= B || (long)tiles * B >= 256) {
    if (s > B) s = B;
    const int bps = (B + s - 1) / s;
    cps = bps * cpb;
  } else {
    s = (512 + tiles - 1) / tiles;
    const int per_b = std::min((s + B - 1) / B, std::max(cpb / 4, 1));      // ranges per batch element, each at least 4 chunks
    cps = (cpb + per_b - 1) / per_b;
  }
  const long total = (long)B * cpb;
  *chunks_per_split = cps;
  *splits = (int)((total + cps - 1) / cps);
}

static bool wgrad_square(int Cin, int Cout, int ntaps) {
  // 64 x 64 block tile (2 x 2 waves) wherever it divides: 34 KiB of LDS per block -> 4 blocks per CU against 3 for 32 x 128, and fewer
  // staged bytes per flop: 145.8 against 143.9 TFLOP/s on G 512->1024, equal on the stride-2 layers.
  return Cout <= 64 || (ntaps == 5 && Cin % 64 == 0 && Cout % 64 == 0);
}
static void wgrad_tile(int Cin, int Cout, int ntaps, int* TC, int* TN) {
  if (wgrad_square(Cin, Cout, ntaps)) { *TC = 64; *TN = 64; } else { *TC = 32; *TN = 128; }
}

size_t wgrad_workspace_bytes(int B, int M, int Cin, int Cout, int ntaps) {
  int TC, TN;
  wgrad_tile(Cin, Cout, ntaps, &TC, &TN);
  int s, cps;
  wgrad_split_plan(B, M, Cin, Cout, TC, TN, &s, &cps);
  return (size_t)s * ntaps * Cin * Cout * sizeof(float) + (size_t)s * Cout * sizeof(double);      // dw slabs + per-split bias partials
}


template <int WAVES_C, int WAVES_N, int WNT, int NTAPS>
static int launch_wgrad(WgradArgs& a, float* dw, hipStream_t s) {
  constexpr int KT = 32, TC = WAVES_C * 32, TN = WAVES_N * WNT * 32;
  int splits;
  wgrad_split_plan(a.B, a.M, a.Cin, a.Cout, TC, TN, &splits, &a.chunks_per_split);
  int minoff = a.off[0], maxoff = a.off[0];
  for (int j = 1; j < NTAPS; ++j) {
    minoff = std::min(minoff, a.off[j]);
    maxoff = std::max(maxoff, a.off[j]);
  }
  const int R = a.in_stride * (KT - 1) + (maxoff - minoff) + 1;
  const size_t lds = sizeof(float) * (((size_t)R * TC + 3 & ~(size_t)3) + (size_t)KT * TN);
  dim3 grid(cdiv(a.Cin, TC), cdiv(a.Cout, TN), splits);
  static const bool no_pipe = getenv("GN_WGRAD_NOPIPE") != nullptr;      // A/B switch: the register-staged weight-gradient kernel everywhere (tests/test_switches_gpu.py)
  bool piped = false;
  if constexpr (NTAPS == 5 && WNT == 1) {
    // opt-in split math: six bf16 products per fp32 product (wgrad_bf16x3.hip); same K-split plan, partial slabs and reduce pass as the exact kernel
    if (a.split_ws && wgrad_bf16x3_supported(a)) {
      if (wgrad_bf16x3_workspace_bytes(a.B, a.M, a.Cin, a.Cout, a.in_stride) > a.split_ws_bytes) {      // never a silent change of arithmetic
        set_error("weight gradient (bf16x3 math): the split operands need %zu bytes, the workspace has %zu -- raise GENNET_CONV_WS_GB",
                  wgrad_bf16x3_workspace_bytes(a.B, a.M, a.Cin, a.Cout, a.in_stride), a.split_ws_bytes);
        return GN_EWORKSPACE;
      }
      int rc = wgrad_bf16x3_run(a, splits, a.split_ws, a.split_ws_bytes, s);
      if (rc) return rc;
      const size_t n = (size_t)NTAPS * a.Cin * a.Cout;
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(n / 4, 256)), dim3(256), 0, s, a.part, dw, n / 4, splits, n / 4);
      return check_launch("wgrad_reduce");
    }
  }
  prof_begin(s);
  if constexpr (NTAPS == 5 && WNT == 1) {
    if (!no_pipe && a.Cin % TC == 0 && a.Cout % TN == 0 && maxoff - minoff + 1 == NTAPS && (size_t)a.Lin * a.Cin * 4 < 0x40000000ull &&
        (size_t)a.M * a.Cout * 4 < 0x40000000ull) {
      // the bias gradient rides along: the blocks of Cin-tile 0 sum the columns of the dy tiles they stage anyway (no separate pass over dy)
      a.db_part = a.db ? reinterpret_cast<double*>(reinterpret_cast<char*>(a.part) + (size_t)splits * NTAPS * a.Cin * a.Cout * sizeof(float)) : nullptr;
      wgrad_pipe_launch(a, grid, WAVES_C == 2, s);                       // wgrad_pipe.hip
      piped = true;
    }
  }
  if (!piped)
    hipLaunchKernelGGL((wgrad_mfma_kernel<WAVES_C, WAVES_N, WNT, NTAPS, KT>), grid, dim3(64 * WAVES_C * WAVES_N), lds, s, a);
  prof_end(s, 2.0 * a.B * (double)a.M * NTAPS * a.Cin * a.Cout, 1, 4.0 * ((double)a.B * a.Lin * a.Cin + (double)a.B * a.M * a.Cout + (double)NTAPS * a.Cin * a.Cout));
  int rc = check_launch("wgrad_mfma");
  if (rc) return rc;
  const size_t n = (size_t)NTAPS * a.Cin * a.Cout;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(n / 4, 256)), dim3(256), 0, s, a.part, dw, n / 4, splits, n / 4);
  if (piped && a.db_part) {
    int rc2 = colred_finalize_f32(a.db_part, a.db, (size_t)a.Cout, splits, s);      // db[n] = sum over splits, fp64, fixed order
    if (rc2) return rc2;
    a.db_done = 1;
  }
  return check_launch("wgrad_reduce");
}

int wgrad_mfma_dispatch(WgradArgs& a, float* dw, size_t ws_bytes, hipStream_t s) {
  if (a.Cin % 4 || a.Cout % 4) {
    set_error("wgrad_mfma: Cin (%d) and Cout (%d) must be multiples of 4", a.Cin, a.Cout);
    return GN_EINVAL;
  }
  if (a.in_stride != 1 && a.in_stride != 2) {
    set_error("wgrad_mfma: in_stride %d unsupported", a.in_stride);
    return GN_EINVAL;
  }
  if (ws_bytes < wgrad_workspace_bytes(a.B, a.M, a.Cin, a.Cout, a.ntaps)) {
    set_error("wgrad_mfma: workspace too small");
    return GN_EWORKSPACE;
  }
  const bool narrow = wgrad_square(a.Cin, a.Cout, a.ntaps);
  switch (a.ntaps) {
    // per-wave tile 32 ci x 32 co x taps (WNT = 1): 32 x 64 (WNT = 2) needs 160 accumulator registers, drops to one wave per SIMD
    // and measured 115 vs 128 TFLOP/s on MI355X
    case 1: return narrow ? launch_wgrad<2, 2, 1, 1>(a, dw, s) : launch_wgrad<1, 4, 1, 1>(a, dw, s);
    case 5: return narrow ? launch_wgrad<2, 2, 1, 5>(a, dw, s) : launch_wgrad<1, 4, 1, 5>(a, dw, s);
    // 2..4 taps (the 3-tap folded UpSampling1D -> Conv1D layers, user graphs): the register-staged kernel
    case 2: return narrow ? launch_wgrad<2, 2, 1, 2>(a, dw, s) : launch_wgrad<1, 4, 1, 2>(a, dw, s);
    case 3: return narrow ? launch_wgrad<2, 2, 1, 3>(a, dw, s) : launch_wgrad<1, 4, 1, 3>(a, dw, s);
    case 4: return narrow ? launch_wgrad<2, 2, 1, 4>(a, dw, s) : launch_wgrad<1, 4, 1, 4>(a, dw, s);
    default:
      set_error("wgrad_mfma: ntaps %d unsupported (1..5)", a.ntaps);
      return GN_EINVAL;
  }
}

}  // namespace gn
