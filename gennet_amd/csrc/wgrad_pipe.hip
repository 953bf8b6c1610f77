// Hand-scheduled, LDS-DMA weight-gradient kernel for gfx950; selected by wgrad_mfma_dispatch (conv_mfma.hip).
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include "common.h"

#ifndef GN_DMA_SPAN_NUM
#define GN_DMA_SPAN_NUM 3      // the staging pieces of a chunk are issued over the first NUM / DEN of its MFMA groups
#define GN_DMA_SPAN_DEN 4
#endif

namespace gn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------
// Weight gradient, pipelined: the design of conv_mfma_pipe_kernel applied to wgrad_mfma_kernel (same tile, same split-K slabs, same
// fixed-order reduce).  Both operands go global -> LDS by LDS-DMA through per-batch-element buffer descriptors whose range check
// supplies the zeros (x rows outside [0, Lin): the convolution's padding; dy rows >= M: the ragged last K-chunk), two LDS stages,
// ONE barrier per K-chunk (the register-staged kernel needs two and a ds_write pass), and the MFMA block is hand-scheduled: per
// k-pair the six ds_read_b32 of the NEXT pair (five taps of x as shifted rows of the slab + one dy value) are issued first, a counted
// s_waitcnt lgkmcnt(6) retires the current pair's operands, then the five v_mfma_f32_32x32x2_f32 (one per tap) issue back to back.
// No staging registers: ~110 VGPRs against 220, three blocks per CU against two.  Needs Cin % TC == 0, Cout % TN == 0, 5 consecutive taps.
// ---------------------------------------------------------------------------------------------
template <int OA0, int OA1, int OA2, int OA3, int OA4, int OB>
__device__ __forceinline__ void wg_group(f32x16& c0, f32x16& c1, f32x16& c2, f32x16& c3, f32x16& c4, float a0, float a1, float a2, float a3, float a4, float b0,
                                         float& na0, float& na1, float& na2, float& na3, float& na4, float& nb0, unsigned addr_a, unsigned addr_b) {
  asm volatile(
      "ds_read_b32 %5, %17 offset:%19\n\t"
      "ds_read_b32 %6, %17 offset:%20\n\t"
      "ds_read_b32 %7, %17 offset:%21\n\t"
      "ds_read_b32 %8, %17 offset:%22\n\t"
      "ds_read_b32 %9, %17 offset:%23\n\t"
      "ds_read_b32 %10, %18 offset:%24\n\t"
      "s_waitcnt lgkmcnt(6)\n\t"
      "v_mfma_f32_32x32x2_f32 %0, %11, %16, %0\n\t"
      "v_mfma_f32_32x32x2_f32 %1, %12, %16, %1\n\t"
      "v_mfma_f32_32x32x2_f32 %2, %13, %16, %2\n\t"
      "v_mfma_f32_32x32x2_f32 %3, %14, %16, %3\n\t"
      "v_mfma_f32_32x32x2_f32 %4, %15, %16, %4"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "=&v"(na0), "=&v"(na1), "=&v"(na2), "=&v"(na3), "=&v"(na4), "=&v"(nb0)
      : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(b0), "v"(addr_a), "v"(addr_b), "i"(OA0), "i"(OA1), "i"(OA2), "i"(OA3), "i"(OA4), "i"(OB)
      : "memory");
}
__device__ __forceinline__ void wg_last(f32x16& c0, f32x16& c1, f32x16& c2, f32x16& c3, f32x16& c4, float a0, float a1, float a2, float a3, float a4, float b0) {
  asm volatile(
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_mfma_f32_32x32x2_f32 %0, %5, %10, %0\n\t"
      "v_mfma_f32_32x32x2_f32 %1, %6, %10, %1\n\t"
      "v_mfma_f32_32x32x2_f32 %2, %7, %10, %2\n\t"
      "v_mfma_f32_32x32x2_f32 %3, %8, %10, %3\n\t"
      "v_mfma_f32_32x32x2_f32 %4, %9, %10, %4"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4)
      : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(b0)
      : "memory");
}
template <int OA0, int OA1, int OA2, int OA3, int OA4, int OB>
__device__ __forceinline__ void wg_first(float& na0, float& na1, float& na2, float& na3, float& na4, float& nb0, unsigned addr_a, unsigned addr_b) {
  asm volatile(
      "ds_read_b32 %0, %6 offset:%8\n\t"
      "ds_read_b32 %1, %6 offset:%9\n\t"
      "ds_read_b32 %2, %6 offset:%10\n\t"
      "ds_read_b32 %3, %6 offset:%11\n\t"
      "ds_read_b32 %4, %6 offset:%12\n\t"
      "ds_read_b32 %5, %7 offset:%13"
      : "=&v"(na0), "=&v"(na1), "=&v"(na2), "=&v"(na3), "=&v"(na4), "=&v"(nb0)
      : "v"(addr_a), "v"(addr_b), "i"(OA0), "i"(OA1), "i"(OA2), "i"(OA3), "i"(OA4), "i"(OB)
      : "memory");
}

template <int TC, int TN, int KT, int IS, int STAGE_BYTES, int STAGE>
struct WgChunk {
  static constexpr int QS = KT / 2;
  static constexpr int oa(int q, int j) { return STAGE * STAGE_BYTES + (IS * 2 * q + j) * TC * 4; }       // + lane part (h * IS rows, channel)
  static constexpr int ob(int q) { return STAGE * STAGE_BYTES + 2 * q * TN * 4; }
  // staging of the NEXT chunk spread over this chunk's groups, as in conv_pipe.hip (PipeChunk::issue): piece K of NP in front of group
  // K * GSPAN / NP, GSPAN = the first three quarters of the chunk's QS groups
  template <int G, int NP, int K = 0, class D>
  static __device__ __forceinline__ void issue(D& dma) {
    if constexpr (K < NP) {
      constexpr int GSPAN = (QS * GN_DMA_SPAN_NUM + GN_DMA_SPAN_DEN - 1) / GN_DMA_SPAN_DEN;
      if constexpr ((K * GSPAN) / NP == G) dma(std::integral_constant<int, K>{}, std::integral_constant<int, 1 - STAGE>{});
      issue<G, NP, K + 1>(dma);
    }
  }
  template <int NP, int K = 0, class D>
  static __device__ __forceinline__ void issue_all(D& dma) {
    if constexpr (K < NP) {
      dma(std::integral_constant<int, K>{}, std::integral_constant<int, 1 - STAGE>{});
      issue_all<NP, K + 1>(dma);
    }
  }
  template <int NP, int Q, class D>
  static __device__ __forceinline__ void run(f32x16 (&acc)[5], float (&s0)[6], float (&s1)[6], unsigned addr_a, unsigned addr_b, D& dma) {
    issue<Q, NP>(dma);
    float(&cur)[6] = (Q & 1) ? s1 : s0;
    float(&nxt)[6] = (Q & 1) ? s0 : s1;
    if constexpr (Q + 1 < QS) {
      wg_group<oa(Q + 1, 0), oa(Q + 1, 1), oa(Q + 1, 2), oa(Q + 1, 3), oa(Q + 1, 4), ob(Q + 1)>(acc[0], acc[1], acc[2], acc[3], acc[4], cur[0], cur[1], cur[2], cur[3], cur[4],
                                                                                               cur[5], nxt[0], nxt[1], nxt[2], nxt[3], nxt[4], nxt[5], addr_a, addr_b);
      run<NP, Q + 1>(acc, s0, s1, addr_a, addr_b, dma);
    } else {
      wg_last(acc[0], acc[1], acc[2], acc[3], acc[4], cur[0], cur[1], cur[2], cur[3], cur[4], cur[5]);
    }
  }
  template <int NP, class D>
  static __device__ __forceinline__ void chunk(f32x16 (&acc)[5], unsigned addr_a, unsigned addr_b, D& dma) {
    float s0[6], s1[6];
    wg_first<oa(0, 0), oa(0, 1), oa(0, 2), oa(0, 3), oa(0, 4), ob(0)>(s0[0], s0[1], s0[2], s0[3], s0[4], s0[5], addr_a, addr_b);
    run<NP, 0>(acc, s0, s1, addr_a, addr_b, dma);
  }
};

template <int WAVES_C, int WAVES_N, int IS>
__global__ __launch_bounds__(64 * WAVES_C * WAVES_N, 2) void wgrad_pipe_kernel(WgradArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NTAPS = 5, KT = 32;
  constexpr int TC = WAVES_C * 32, TN = WAVES_N * 32;
  constexpr int NT = 64 * WAVES_C * WAVES_N;
  constexpr int R = IS * (KT - 1) + NTAPS;                 // x rows of one K-chunk (taps consecutive: checked by the launcher)
  constexpr int SLAB = ((R * TC + 255) / 256) * 256;       // floats, rounded up to whole 1-KiB DMA pieces
  constexpr int BUF = SLAB + KT * TN;
  constexpr int STAGE_BYTES = BUF * 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wc = wave / WAVES_N, wn = wave % WAVES_N;
  const int i32 = lane & 31, h = lane >> 5;
  // block -> (Cin tile, Cout tile, K-split), grid order.  (Whole K-splits per XCD and a patch order of the tile plane were measured in rounds 2-3:
  // a third fewer fabric fetches, 1 % slower -- the Infinity Cache already serves the 8 XCDs that read the same dy tiles at the same time; removed.)
  int ct = blockIdx.x, nt_ = blockIdx.y, split = blockIdx.z;
  ct = __builtin_amdgcn_readfirstlane(ct); nt_ = __builtin_amdgcn_readfirstlane(nt_); split = __builtin_amdgcn_readfirstlane(split);
  const int c0 = ct * TC, n0 = nt_ * TN;

  int minoff = a.off[0];
#pragma unroll
  for (int j = 1; j < NTAPS; ++j) minoff = min(minoff, a.off[j]);

  f32x16 acc[NTAPS];
#pragma unroll
  for (int j = 0; j < NTAPS; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int cpb = (a.M + KT - 1) / KT;
  const int c_lo = split * a.chunks_per_split, c_hi = min(a.B * cpb, c_lo + a.chunks_per_split);
  const int n_chunks = max(c_hi - c_lo, 0);

  // per-lane byte offsets of the DMA granules inside one batch element, for m0 = 0 (chunk-invariant); the chunk adds m0 rows
  constexpr int S_COUNT = R * (TC / 4);
  constexpr int S_ITEMS = (S_COUNT + NT - 1) / NT;
  constexpr int D_COUNT = KT * (TN / 4);
  constexpr int D_ITEMS = D_COUNT / NT;
  static_assert(D_COUNT % NT == 0, "dy tile must be whole DMA pieces per thread");
  int soff[S_ITEMS], doff[D_ITEMS];
#pragma unroll
  for (int it = 0; it < S_ITEMS; ++it) {
    const int id = tid + it * NT;
    const int r = id / (TC / 4), c4 = id % (TC / 4);
    soff[it] = ((minoff + r) * a.Cin + c0 + 4 * c4) * 4;
  }
#pragma unroll
  for (int it = 0; it < D_ITEMS; ++it) {
    const int id = tid + it * NT;
    const int r = id / (TN / 4), n4 = id % (TN / 4);
    doff[it] = (r * a.Cout + n0 + 4 * n4) * 4;
  }
  const int xbytes = __builtin_amdgcn_readfirstlane(a.Lin * a.Cin * 4), dybytes = __builtin_amdgcn_readfirstlane(a.M * a.Cout * 4);
  // staging of one K-chunk: dma_setup builds the chunk's descriptors and row offsets (scalar work), dma_piece issues ONE wave-wide 16-byte
  // LDS-DMA of every wave (pieces 0 .. S_ITEMS-1 the x slab, the rest the dy tile) into stage STG.  The chunk loop spreads the pieces over the
  // MFMA groups instead of issuing them in one burst after the barrier (scripts/mfma_peak.hip: the burst costs the matrix pipe 0.9 %, the
  // spread issue 0.1 %).  The row offsets stay VALU adds: in an SGPR offset they would escape the descriptor's range check, which is what
  // turns rows outside [0, Lin) and dy rows >= M into zeros.
  constexpr int NPIECES = S_ITEMS + D_ITEMS;
  __amdgpu_buffer_rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, 0, 0x00020000), ds = xs;
  int xrow = 0, drow = 0;
  auto dma_setup = [&](int ch) {
    const int b = __builtin_amdgcn_readfirstlane((c_lo + ch) / cpb), m0 = __builtin_amdgcn_readfirstlane(((c_lo + ch) % cpb) * KT);
    const uintptr_t xp = (uintptr_t)(a.x + (size_t)b * a.Lin * a.Cin), dp = (uintptr_t)(a.dy + (size_t)b * a.M * a.Cout);
    // (unsigned halves: readfirstlane returns int, and a sign-extended low half would corrupt the high one)
    const unsigned xlo = __builtin_amdgcn_readfirstlane((unsigned)xp), xhi = __builtin_amdgcn_readfirstlane((unsigned)(xp >> 32));
    const unsigned dlo = __builtin_amdgcn_readfirstlane((unsigned)dp), dhi = __builtin_amdgcn_readfirstlane((unsigned)(dp >> 32));
    xs = __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)xhi << 32) | xlo), 0, xbytes, 0x00020000);
    ds = __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)dhi << 32) | dlo), 0, dybytes, 0x00020000);
    xrow = IS * m0 * a.Cin * 4;
    drow = m0 * a.Cout * 4;
  };
  const int wv64 = __builtin_amdgcn_readfirstlane(tid & ~63);       // the wave's first thread as a scalar: the LDS-DMA destination (M0) needs no per-piece v_readfirstlane
  auto dma_piece = [&](auto kc, auto stg) {
    constexpr int k = decltype(kc)::value;
    float* stage = smem + decltype(stg)::value * BUF;
    if constexpr (k < S_ITEMS) {
      if ((k + 1) * NT <= S_COUNT || tid + k * NT < S_COUNT)            // x rows before 0 give a negative (= huge unsigned) offset: out of range -> 0
        gn_buffer_load_lds(xs, (lptr_t)(stage + (k * NT + wv64) * 4), 16, soff[k] + xrow, 0, 0, 0);
    } else {
      constexpr int it = k - S_ITEMS;
      gn_buffer_load_lds(ds, (lptr_t)(stage + SLAB + (it * NT + wv64) * 4), 16, doff[it] + drow, 0, 0, 0);
    }
  };

  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const unsigned addr_a = lds0 + ((h * IS) * TC + wc * 32 + i32) * 4;
  const unsigned addr_b = lds0 + (SLAB + h * TN + wn * 32 + i32) * 4;

  // bias gradient on the way (blocks of Cin-tile 0 only): thread (column tid % TN, row group tid / TN) adds its rows of every staged
  // dy tile in fp64 (rows >= M were staged as zeros); 8 LDS reads per chunk against the 96 of the MFMA operands
  const bool do_bias = a.db_part != nullptr && ct == 0;
  constexpr int RG = NT / TN, RPG = KT / RG;
  double bsum = 0.0;
  auto bias_chunk = [&](const float* stage) {
    const float* col = stage + SLAB + (tid / TN) * RPG * TN + (tid % TN);
#pragma unroll
    for (int r = 0; r < RPG; ++r) bsum += (double)col[r * TN];
  };

  if (n_chunks > 0) {
    dma_setup(0);
    WgChunk<TC, TN, KT, IS, STAGE_BYTES, 1>::template issue_all<NPIECES>(dma_piece);      // chunk 0 into stage 0, in one burst
  }
  __syncthreads();
  for (int ch = 0; ch < n_chunks; ch += 2) {
    dma_setup(min(ch + 1, n_chunks - 1));
    if (do_bias) bias_chunk(smem);
    WgChunk<TC, TN, KT, IS, STAGE_BYTES, 0>::template chunk<NPIECES>(acc, addr_a, addr_b, dma_piece);
    __syncthreads();
    if (ch + 1 < n_chunks) {
      dma_setup(min(ch + 2, n_chunks - 1));
      if (do_bias) bias_chunk(smem + BUF);
      WgChunk<TC, TN, KT, IS, STAGE_BYTES, 1>::template chunk<NPIECES>(acc, addr_a, addr_b, dma_piece);
      __syncthreads();
    }
  }
  if (do_bias) {                                           // every wave is past the last barrier: the stages are free
    double* red = reinterpret_cast<double*>(smem);
    red[tid] = bsum;
    __syncthreads();
    if (tid < TN) {
      double t = 0.0;
#pragma unroll
      for (int g = 0; g < RG; ++g) t += red[g * TN + tid];
      a.db_part[(size_t)split * a.Cout + n0 + tid] = t;
    }
  }
  asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]));

  const int n = n0 + wn * 32 + i32;
#pragma unroll
  for (int j = 0; j < NTAPS; ++j) {
    // accumulator j belongs to the tap at slab row offset j, i.e. the tap whose off equals minoff + j
    int tap = 0;
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
      if (a.off[t] - minoff == j) tap = t;
    float* pj = a.part + ((size_t)split * NTAPS + tap) * a.Cin * a.Cout;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = c0 + wc * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      pj[(size_t)c * a.Cout + n] = acc[j][r];
    }
  }
  // (The split-K reduction folded into this kernel -- last-arriving block per tile -- was measured in round 4: -2.3 % on the step; the device-scope
  // release / acquire is an L2 write-back + invalidate of an XCD-private L2 under the blocks still in their main loop.  Removed; profiles/r04_ab_wgrad_fold.json.)
#endif
}

template <int WAVES_C, int WAVES_N, int IS>
static void launch_wgrad_pipe(const WgradArgs& a, dim3 grid, hipStream_t s) {
  constexpr int KT = 32, TC = WAVES_C * 32, TN = WAVES_N * 32;
  constexpr int R = IS * (KT - 1) + 5, SLAB = ((R * TC + 255) / 256) * 256;
  constexpr size_t lds = 2 * sizeof(float) * ((size_t)SLAB + (size_t)KT * TN);
  static_assert(lds / 2 + (size_t)(R * TC + KT * TN) * 4 < 65536, "ds_read offsets must fit 16 bits");
  if (lds > 64 * 1024) {
    static unsigned long long lds_done = 0;
    allow_big_lds((const void*)wgrad_pipe_kernel<WAVES_C, WAVES_N, IS>, &lds_done);
  }
  hipLaunchKernelGGL((wgrad_pipe_kernel<WAVES_C, WAVES_N, IS>), grid, dim3(64 * WAVES_C * WAVES_N), lds, s, a);
}


void wgrad_pipe_launch(const WgradArgs& a, dim3 grid, bool narrow, hipStream_t s) {
  if (narrow) {
    if (a.in_stride == 1) launch_wgrad_pipe<2, 2, 1>(a, grid, s);
    else launch_wgrad_pipe<2, 2, 2>(a, grid, s);
  } else {
    if (a.in_stride == 1) launch_wgrad_pipe<1, 4, 1>(a, grid, s);
    else launch_wgrad_pipe<1, 4, 2>(a, grid, s);
  }
}

}  // namespace gn
