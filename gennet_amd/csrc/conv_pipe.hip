// Hand-scheduled, software-pipelined implicit-GEMM Conv1D (forward + data gradient) for gfx950; selected by conv_mfma_dispatch.
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include "common.h"
#include "conv_epilogue.h"

#ifndef GN_DMA_SPAN_NUM
#define GN_DMA_SPAN_NUM 3      // the staging pieces of a chunk are issued over the first NUM / DEN of its MFMA groups
#define GN_DMA_SPAN_DEN 4
#endif

namespace gn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------
// Software-pipelined variant of the all-DMA kernel: same tile, same LDS-DMA staging, same one-barrier double-stage chunk loop and
// the same epilogue, but the MFMA block of a chunk is HAND-SCHEDULED.  hipcc turns the source-level operand double buffering of
// conv_mfma_dma_kernel into "ds_read, s_waitcnt lgkmcnt(0), 2-4 MFMAs" (reads sunk next to their use, every wait a full drain), so a
// wave exposes one LDS latency per 128-256 cycles of matrix work and the pipe only stays busy through the other waves of the SIMD.
// Here each group of four MFMAs is one asm statement: the operand reads of the NEXT group are issued first, a counted s_waitcnt
// lgkmcnt retires exactly the CURRENT group's operands (issued one group = 256 matrix cycles earlier), then the four
// v_mfma_f32_32x32x2_f32 issue back to back.  All LDS addresses are chunk-invariant VGPRs (one per tap for the input slab, one for
// the weight tile) plus compile-time 16-bit offsets (stage, tap, k-step, tile), so the block has no address arithmetic at all.
// The layout is compile-time: IS (input stride) is a template parameter and the slab holds IS*(TM-1)+NTAPS rows (consecutive taps).
// Descriptor and tile indices go through readfirstlane so that the buffer_load ... lds of the slab are not wrapped in waterfall loops.
// ---------------------------------------------------------------------------------------------
// LDS layout and operand reads (round 3, after `--pmc SQ_LDS_BANK_CONFLICT`: 0.78 of the kernel's LDS cycles were conflict replays).  The slab
// holds rows of KC = 8 channels (32 B).  Read with ds_read_b32 at one dword per lane, the 32 lanes of a half-wave sit 32 B apart: 4 banks of
// the 32 that instruction sees, 8 addresses each -- 16 LDS cycles per read instead of 2, the LDS array about half busy under 16 waves.  Now
// every lane fetches FOUR channels of its row with one ds_read_b128 (lane half h takes channels 4h..4h+3: k-step s of a tap multiplies
// channel s in the lower half-wave and channel 4+s in the upper one; the weight rows follow suit) and the two 16-byte granules of a row are
// swapped in rows 8..15 mod 16, so that the 16 lanes the hardware serves per LDS cycle (rows distinct mod 16) fall on 16 distinct 16-byte
// slots of the 256-byte bank row: conflict-free, 4 cycles per instruction, 10 input reads per chunk instead of 40.  The swap is applied by the
// DMA (which granule a lane fetches from global memory -- the two lanes of a row still read its 32 contiguous bytes) and by the per-tap read
// address.  Schedule per tap: the group of k-step 0 issues the next group's weight reads and then the NEXT tap's two input reads; lgkmcnt
// retires in order, so the counted waits over a tap's four groups are 4, 4, 2, 2 (2 throughout on the last tap, which issues no input read).
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int OB0, int OB1, int WAIT>
__device__ __forceinline__ void pipe_group(f32x16& c00, f32x16& c01, f32x16& c10, f32x16& c11, float a0, float a1, float b0, float b1, float& nb0, float& nb1,
                                           unsigned addr_b) {
  asm volatile(
      "ds_read_b32 %4, %10 offset:%11\n\t"
      "ds_read_b32 %5, %10 offset:%12\n\t"
      "s_waitcnt lgkmcnt(%13)\n\t"
      "v_mfma_f32_32x32x2_f32 %0, %6, %8, %0\n\t"
      "v_mfma_f32_32x32x2_f32 %1, %6, %9, %1\n\t"
      "v_mfma_f32_32x32x2_f32 %2, %7, %8, %2\n\t"
      "v_mfma_f32_32x32x2_f32 %3, %7, %9, %3"
      : "+v"(c00), "+v"(c01), "+v"(c10), "+v"(c11), "=&v"(nb0), "=&v"(nb1)
      : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(addr_b), "i"(OB0), "i"(OB1), "i"(WAIT)
      : "memory");
}
// ... the same, and the next tap's input rows for both row tiles (issued after the weight reads)
template <int OA0, int OA1, int OB0, int OB1, int WAIT>
__device__ __forceinline__ void pipe_group_a(f32x16& c00, f32x16& c01, f32x16& c10, f32x16& c11, float a0, float a1, float b0, float b1, float& nb0, float& nb1,
                                             f32x4& na0, f32x4& na1, unsigned addr_a, unsigned addr_b) {
  asm volatile(
      "ds_read_b32 %4, %13 offset:%16\n\t"
      "ds_read_b32 %5, %13 offset:%17\n\t"
      "ds_read_b128 %6, %12 offset:%14\n\t"
      "ds_read_b128 %7, %12 offset:%15\n\t"
      "s_waitcnt lgkmcnt(%18)\n\t"
      "v_mfma_f32_32x32x2_f32 %0, %8, %10, %0\n\t"
      "v_mfma_f32_32x32x2_f32 %1, %8, %11, %1\n\t"
      "v_mfma_f32_32x32x2_f32 %2, %9, %10, %2\n\t"
      "v_mfma_f32_32x32x2_f32 %3, %9, %11, %3"
      : "+v"(c00), "+v"(c01), "+v"(c10), "+v"(c11), "=&v"(nb0), "=&v"(nb1), "=&v"(na0), "=&v"(na1)
      : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(addr_a), "v"(addr_b), "i"(OA0), "i"(OA1), "i"(OB0), "i"(OB1), "i"(WAIT)
      : "memory");
}
__device__ __forceinline__ void pipe_last(f32x16& c00, f32x16& c01, f32x16& c10, f32x16& c11, float a0, float a1, float b0, float b1) {
  asm volatile(
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_mfma_f32_32x32x2_f32 %0, %4, %6, %0\n\t"
      "v_mfma_f32_32x32x2_f32 %1, %4, %7, %1\n\t"
      "v_mfma_f32_32x32x2_f32 %2, %5, %6, %2\n\t"
      "v_mfma_f32_32x32x2_f32 %3, %5, %7, %3"
      : "+v"(c00), "+v"(c01), "+v"(c10), "+v"(c11)
      : "v"(a0), "v"(a1), "v"(b0), "v"(b1)
      : "memory");
}
template <int OA0, int OA1, int OB0, int OB1>
__device__ __forceinline__ void pipe_first(f32x4& na0, f32x4& na1, float& nb0, float& nb1, unsigned addr_a, unsigned addr_b) {
  asm volatile(
      "ds_read_b128 %0, %4 offset:%6\n\t"
      "ds_read_b128 %1, %4 offset:%7\n\t"
      "ds_read_b32 %2, %5 offset:%8\n\t"
      "ds_read_b32 %3, %5 offset:%9"
      : "=&v"(na0), "=&v"(na1), "=&v"(nb0), "=&v"(nb1)
      : "v"(addr_a), "v"(addr_b), "i"(OA0), "i"(OA1), "i"(OB0), "i"(OB1)
      : "memory");
}

// Narrow wave tile (64 rows x 32 columns per wave: two accumulator tiles, one weight read per two MFMAs).  Twice the waves for the same
// block tile -- for launches whose 64 x 64 wave tiles do not fill the chip's 1024 SIMDs (the script's own batch 8, bbhMahoGANy.py:84-89).
template <int OB0, int WAIT>
__device__ __forceinline__ void pipe_group_n(f32x16& c00, f32x16& c10, float a0, float a1, float b0, float& nb0, unsigned addr_b) {
  asm volatile(
      "ds_read_b32 %2, %6 offset:%7\n\t"
      "s_waitcnt lgkmcnt(%8)\n\t"
      "v_mfma_f32_32x32x2_f32 %0, %3, %5, %0\n\t"
      "v_mfma_f32_32x32x2_f32 %1, %4, %5, %1"
      : "+v"(c00), "+v"(c10), "=&v"(nb0)
      : "v"(a0), "v"(a1), "v"(b0), "v"(addr_b), "i"(OB0), "i"(WAIT)
      : "memory");
}
template <int OA0, int OA1, int OB0, int WAIT>
__device__ __forceinline__ void pipe_group_na(f32x16& c00, f32x16& c10, float a0, float a1, float b0, float& nb0, f32x4& na0, f32x4& na1, unsigned addr_a,
                                              unsigned addr_b) {
  asm volatile(
      "ds_read_b32 %2, %9 offset:%12\n\t"
      "ds_read_b128 %3, %8 offset:%10\n\t"
      "ds_read_b128 %4, %8 offset:%11\n\t"
      "s_waitcnt lgkmcnt(%13)\n\t"
      "v_mfma_f32_32x32x2_f32 %0, %5, %7, %0\n\t"
      "v_mfma_f32_32x32x2_f32 %1, %6, %7, %1"
      : "+v"(c00), "+v"(c10), "=&v"(nb0), "=&v"(na0), "=&v"(na1)
      : "v"(a0), "v"(a1), "v"(b0), "v"(addr_a), "v"(addr_b), "i"(OA0), "i"(OA1), "i"(OB0), "i"(WAIT)
      : "memory");
}
__device__ __forceinline__ void pipe_last_n(f32x16& c00, f32x16& c10, float a0, float a1, float b0) {
  asm volatile(
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_mfma_f32_32x32x2_f32 %0, %2, %4, %0\n\t"
      "v_mfma_f32_32x32x2_f32 %1, %3, %4, %1"
      : "+v"(c00), "+v"(c10)
      : "v"(a0), "v"(a1), "v"(b0)
      : "memory");
}
template <int OA0, int OA1, int OB0>
__device__ __forceinline__ void pipe_first_n(f32x4& na0, f32x4& na1, float& nb0, unsigned addr_a, unsigned addr_b) {
  asm volatile(
      "ds_read_b128 %0, %3 offset:%5\n\t"
      "ds_read_b128 %1, %3 offset:%6\n\t"
      "ds_read_b32 %2, %4 offset:%7"
      : "=&v"(na0), "=&v"(na1), "=&v"(nb0)
      : "v"(addr_a), "v"(addr_b), "i"(OA0), "i"(OA1), "i"(OB0)
      : "memory");
}

// One K-chunk (KC = 8 channels x NTAPS taps) of a wave, on stage STAGE.  A group = one k-step (two channels: s and 4+s of the chunk) of one
// tap: 4 MFMAs on the 64 x 64 wave tile, 2 on the narrow one.  Input operands are double-buffered per TAP (A[tap parity][row tile], four
// k-steps per register quad), weight operands per GROUP (B[step parity][column tile]).
template <int TN, int KC, int NTAPS, int STAGE_BYTES, int STAGE>
struct PipeChunk {
  static_assert(KC == 8, "two 16-byte granules per slab row: lane half h reads channels 4h..4h+3");
  static constexpr int SB = STAGE * STAGE_BYTES;
  static constexpr int OA1 = 32 * KC * 4;                                                  // second row tile: 32 slab rows further
  static constexpr int ob(int t, int s) { return SB + (t * KC + s) * TN * 4; }             // + 4h rows of the weight stage in the address register
  // staging of the NEXT chunk (into the other stage), spread over this chunk's groups: piece K of NP goes in front of group K * GSPAN / NP,
  // GSPAN = the first three quarters of the chunk's GTOTAL groups (the last pieces still have a quarter of the chunk to land before the barrier)
  template <int G, int GTOTAL, int NP, int K = 0, class D>
  static __device__ __forceinline__ void issue(D& dma) {
    if constexpr (K < NP) {
      constexpr int GSPAN = (GTOTAL * GN_DMA_SPAN_NUM + GN_DMA_SPAN_DEN - 1) / GN_DMA_SPAN_DEN;
      if constexpr ((K * GSPAN) / NP == G) dma(std::integral_constant<int, K>{}, std::integral_constant<int, 1 - STAGE>{});
      issue<G, GTOTAL, NP, K + 1>(dma);
    }
  }
  template <int NP, int K = 0, class D>
  static __device__ __forceinline__ void issue_all(D& dma) {
    if constexpr (K < NP) {
      dma(std::integral_constant<int, K>{}, std::integral_constant<int, 1 - STAGE>{});
      issue_all<NP, K + 1>(dma);
    }
  }
  // ---- 64 x 64 wave tile ----
  template <int NP, int I, int S, class D>
  static __device__ __forceinline__ void run(f32x16 (&acc)[2][2], f32x4 (&A)[2][2], float (&B)[2][2], const unsigned (&addr_a)[NTAPS], unsigned addr_b, D& dma) {
    issue<I * 4 + S, NTAPS * 4, NP>(dma);
    f32x4(&ac)[2] = A[I & 1];
    f32x4(&an)[2] = A[(I + 1) & 1];
    float(&bc)[2] = B[S & 1];
    float(&bn)[2] = B[(S + 1) & 1];
    constexpr bool last_tap = I + 1 == NTAPS;
    if constexpr (last_tap && S == 3) {
      pipe_last(acc[0][0], acc[0][1], acc[1][0], acc[1][1], ac[0][3], ac[1][3], bc[0], bc[1]);
    } else {
      constexpr int NI = S == 3 ? I + 1 : I, NS = (S + 1) & 3;
      if constexpr (S == 0 && !last_tap)
        pipe_group_a<SB, SB + OA1, ob(NI, NS), ob(NI, NS) + 128, 4>(acc[0][0], acc[0][1], acc[1][0], acc[1][1], ac[0][S], ac[1][S], bc[0], bc[1], bn[0], bn[1], an[0],
                                                                     an[1], addr_a[last_tap ? I : I + 1], addr_b);
      else
        pipe_group<ob(NI, NS), ob(NI, NS) + 128, (S == 1 && !last_tap) ? 4 : 2>(acc[0][0], acc[0][1], acc[1][0], acc[1][1], ac[0][S], ac[1][S], bc[0], bc[1], bn[0],
                                                                                 bn[1], addr_b);
      run<NP, NI, NS>(acc, A, B, addr_a, addr_b, dma);
    }
  }
  template <int NP, class D>
  static __device__ __forceinline__ void chunk(f32x16 (&acc)[2][2], const unsigned (&addr_a)[NTAPS], unsigned addr_b, D& dma) {
    f32x4 A[2][2];
    float B[2][2];
    pipe_first<SB, SB + OA1, ob(0, 0), ob(0, 0) + 128>(A[0][0], A[0][1], B[0][0], B[0][1], addr_a[0], addr_b);
    run<NP, 0, 0>(acc, A, B, addr_a, addr_b, dma);
  }
  // ---- narrow wave tile (acc[2][1]).  The taps visited are FIRST, FIRST + STEP, ... (NSEQ of them); with ALT the taps of odd index
  // accumulate into the second set (the merged two-phase launch) ----
  template <int NP, int FIRST, int STEP, int NSEQ, bool ALT, int I, int S, class D>
  static __device__ __forceinline__ void run_n(f32x16 (&accA)[2][1], f32x16 (&accB)[2][1], f32x4 (&A)[2][2], float (&B)[2], const unsigned (&addr_a)[NTAPS],
                                               unsigned addr_b, D& dma) {
    issue<I * 4 + S, NSEQ * 4, NP>(dma);
    constexpr int T = FIRST + I * STEP;
    f32x16(&acc)[2][1] = (ALT && (T & 1)) ? accB : accA;
    f32x4(&ac)[2] = A[I & 1];
    f32x4(&an)[2] = A[(I + 1) & 1];
    constexpr bool last_tap = I + 1 == NSEQ;
    if constexpr (last_tap && S == 3) {
      pipe_last_n(acc[0][0], acc[1][0], ac[0][3], ac[1][3], B[S & 1]);
    } else {
      constexpr int NI = S == 3 ? I + 1 : I, NS = (S + 1) & 3, NT_ = FIRST + NI * STEP;
      if constexpr (S == 0 && !last_tap)
        pipe_group_na<SB, SB + OA1, ob(NT_, NS), 3>(acc[0][0], acc[1][0], ac[0][S], ac[1][S], B[S & 1], B[(S + 1) & 1], an[0], an[1],
                                                    addr_a[last_tap ? T : T + STEP], addr_b);
      else
        pipe_group_n<ob(NT_, NS), (S == 1 && !last_tap) ? 3 : 1>(acc[0][0], acc[1][0], ac[0][S], ac[1][S], B[S & 1], B[(S + 1) & 1], addr_b);
      run_n<NP, FIRST, STEP, NSEQ, ALT, NI, NS>(accA, accB, A, B, addr_a, addr_b, dma);
    }
  }
  template <int NP, int FIRST, int STEP, int NSEQ, bool ALT, class D>
  static __device__ __forceinline__ void chunk_seq(f32x16 (&accA)[2][1], f32x16 (&accB)[2][1], const unsigned (&addr_a)[NTAPS], unsigned addr_b, D& dma) {
    f32x4 A[2][2];
    float B[2];
    pipe_first_n<SB, SB + OA1, ob(FIRST, 0)>(A[0][0], A[0][1], B[0], addr_a[FIRST], addr_b);
    run_n<NP, FIRST, STEP, NSEQ, ALT, 0, 0>(accA, accB, A, B, addr_a, addr_b, dma);
  }
  template <int NP, class D>
  static __device__ __forceinline__ void chunk(f32x16 (&acc)[2][1], const unsigned (&addr_a)[NTAPS], unsigned addr_b, D& dma) {
    chunk_seq<NP, 0, 1, NTAPS, false>(acc, acc, addr_a, addr_b, dma);
  }
  // merged two-phase launch: taps with even index feed accumulator set A, taps with odd index set B
  template <int NP, class D>
  static __device__ __forceinline__ void chunk_m(f32x16 (&accA)[2][1], f32x16 (&accB)[2][1], const unsigned (&addr_a)[NTAPS], unsigned addr_b, D& dma) {
    chunk_seq<NP, 0, 1, NTAPS, true>(accA, accB, addr_a, addr_b, dma);
  }
  // phase-split form of the merged launch: a wave owns ONE output phase and runs only the taps of its parity PAR (3 or 2 of the 5)
  template <int PAR, int NP, class D>
  static __device__ __forceinline__ void chunk_p(f32x16 (&acc)[2][1], const unsigned (&addr_a)[NTAPS], unsigned addr_b, D& dma) {
    chunk_seq<NP, PAR, 2, (NTAPS + 1 - PAR) / 2, false>(acc, acc, addr_a, addr_b, dma);
  }
};

// SPAN = consecutive input rows the taps cover (= NTAPS, except in the merged two-phase launch where five taps cover three rows);
// MERGE (narrow waves only): taps of even / odd index accumulate two output phases.  1: every wave keeps two accumulator sets (A / B) and runs
// all five taps; 2: twice the waves, each owning ONE phase and running only the taps of its parity.
template <int WAVES_M, int WAVES_N, int NTAPS, int IS, int KC = 8, int WN = 2, int SPAN = NTAPS, int MERGE = 0>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N * (MERGE == 2 ? 2 : 1), (64 * WAVES_M * WAVES_N * (MERGE == 2 ? 2 : 1) > 256 ? 1 : 2)) void conv_mfma_pipe_kernel(ConvArgs a, int m_tiles, int n_tiles, int patch) {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(!MERGE || WN == 1, "the merged launch runs on narrow waves");
  constexpr int WM = 2;
  constexpr int TM = WAVES_M * WM * 32;
  constexpr int TN = WAVES_N * WN * 32;
  constexpr int NT = 64 * WAVES_M * WAVES_N * (MERGE == 2 ? 2 : 1);
  constexpr int R = IS * (TM - 1) + SPAN;                  // staged input rows: the launcher checks that the taps are consecutive
  constexpr int RPER = (R + IS - 1) / IS;
  constexpr int SLAB = IS * RPER * KC;                     // floats
  constexpr int BUF = SLAB + NTAPS * KC * TN;
  constexpr int STAGE_BYTES = BUF * 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  const int tid = threadIdx.x, lane = tid & 63, wave_all = tid >> 6;
  const int ph = MERGE == 2 ? wave_all / (WAVES_M * WAVES_N) : 0;              // phase-split merged launch: the wave's output phase
  const int wave = MERGE == 2 ? wave_all % (WAVES_M * WAVES_N) : wave_all;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int i32 = lane & 31, h = lane >> 5;
  // block -> tile.  Plain order is (slab, column tile) with the column tile fastest: block i runs on XCD i mod 8, so every XCD meets
  // every input slab and the slab crosses the fabric 8 times.  Patch order (patch >= 0) hands each XCD, out of every 512 consecutive
  // blocks, the 64 blocks of ONE patch of (64 / PN) slabs x PN column tiles (PN = 8, or 4 when there are only 4 column tiles) -- the
  // set an XCD has in flight at two blocks per CU -- so its L2 serves each slab chunk to PN blocks and each weight chunk to 64 / PN.
  // patch = log2(column tiles / PN) | log2(PN) << 8.  Blocks past the last whole 512 keep the plain order (both orders cover the
  // same leading slabs).  Measured on G 512 -> 1024: FETCH_SIZE -25 %, +0.3 % on the step.
  const int bid = blockIdx.x;
  int n_lin, slab;
  if (patch >= 0 && bid < (int)(gridDim.x & ~511u)) {
    const int ps = patch & 255, pn = patch >> 8;
    const int r = bid & 511, p = (bid >> 9) * 8 + (r & 7), idx = r >> 3;
    slab = ((p >> ps) << (6 - pn)) + (idx >> pn);
    n_lin = ((p & ((1 << ps) - 1)) << pn) + (idx & ((1 << pn) - 1));
  } else {
    n_lin = bid % n_tiles;
    slab = bid / n_tiles;
  }
  const int n_tile = __builtin_amdgcn_readfirstlane(n_lin);
  const int m_tile = __builtin_amdgcn_readfirstlane(slab % m_tiles);
  const int b = __builtin_amdgcn_readfirstlane(slab / m_tiles);
  const int m0 = m_tile * TM, n0 = n_tile * TN;

  int minoff = a.t.off[0];
#pragma unroll
  for (int j = 1; j < NTAPS; ++j) minoff = min(minoff, a.t.off[j]);

  f32x16 acc[WM][WN];
  f32x16 accB[WM][1];                      // second accumulator set of the merged launch (unused, and optimised away, otherwise)
#pragma unroll
  for (int mt = 0; mt < WM; ++mt)
#pragma unroll
    for (int nt = 0; nt < WN; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
  if constexpr (MERGE == 1) {
#pragma unroll
    for (int mt = 0; mt < WM; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) accB[mt][0][r] = 0.f;
  }

  const int t_base = IS * m0 + minoff;
  // descriptor inputs made PROVABLY wave-uniform (pointer halves and byte count through readfirstlane): otherwise hipcc wraps every
  // buffer_load ... lds in a waterfall loop (v_readfirstlane x4, compare, saveexec, load, loop), ~12 instructions per DMA piece
  const uintptr_t xbp = (uintptr_t)(a.x + (size_t)b * a.Lin * a.Cin);
  const unsigned xb_lo = __builtin_amdgcn_readfirstlane((unsigned)xbp), xb_hi = __builtin_amdgcn_readfirstlane((unsigned)(xbp >> 32));
  const int xbytes = __builtin_amdgcn_readfirstlane(a.Lin * a.Cin * 4);
  const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)xb_hi << 32) | xb_lo), 0, xbytes, 0x00020000);

  constexpr int S_COUNT = SLAB / 4;                        // 16-byte granules of one slab stage, in LDS order
  constexpr int S_ITEMS = (S_COUNT + NT - 1) / NT;
  constexpr int W_TOTAL = NTAPS * KC * (TN / 4);
  constexpr int W_ITEMS = (W_TOTAL + NT - 1) / NT;
  int soff[S_ITEMS];
#pragma unroll
  for (int it = 0; it < S_ITEMS; ++it) {
    const int id = tid + it * NT;
    const int lr = id / (KC / 4), c4 = (id % (KC / 4)) ^ ((lr >> 3) & 1);            // the row's two granules swapped in rows 8..15 mod 16 (PipeChunk)
    const int r = (IS == 1) ? lr : (lr < RPER ? 2 * lr : 2 * (lr - RPER) + 1);
    soff[it] = (id < S_COUNT && r < R) ? ((t_base + r) * a.Cin + 4 * c4) * 4 : 0x40000000;   // out of range -> the descriptor returns 0
  }
  // weights through a buffer descriptor too (round 3): a 32-bit lane offset and the chunk's offset in an SGPR instead of a 64-bit pointer per
  // lane advanced by VALU adds every chunk -- scripts/mfma_peak.hip: six global_load_lds per chunk cost the matrix pipe 1.9 %, six
  // buffer_load ... lds 0.9 %
  int maxw = a.t.widx[0];
#pragma unroll
  for (int j = 1; j < NTAPS; ++j) maxw = max(maxw, a.t.widx[j]);
  const uintptr_t wbp = (uintptr_t)a.w;
  const unsigned wb_lo = __builtin_amdgcn_readfirstlane((unsigned)wbp), wb_hi = __builtin_amdgcn_readfirstlane((unsigned)(wbp >> 32));
  const int wbytes = __builtin_amdgcn_readfirstlane((maxw + 1) * a.Cin * a.Cout * 4);
  const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)wb_hi << 32) | wb_lo), 0, wbytes, 0x00020000);
  int woff[W_ITEMS];
#pragma unroll
  for (int it = 0; it < W_ITEMS; ++it) {
    const int id = min(tid + it * NT, W_TOTAL - 1);
    const int n4 = id % (TN / 4);
    const int kk = (id / (TN / 4)) % KC;
    const int j = id / ((TN / 4) * KC);
    woff[it] = ((a.t.widx[j] * a.Cin + kk) * a.Cout + n0 + 4 * n4) * 4;
  }
  // One staging piece = one wave-wide 16-byte LDS-DMA of every wave of the block: pieces 0 .. S_ITEMS-1 the input slab, the rest the weight
  // tile, of the chunk at channel c0_next, into stage STG.  The chunk loop does not issue them in one burst after the barrier but one every
  // few MFMA groups (PipeChunk::issue): in scripts/mfma_peak.hip the burst costs the matrix pipe 0.9 % at four blocks per CU (5 % at one),
  // the spread issue 0.1 %.
  constexpr int NPIECES = S_ITEMS + W_ITEMS;
  int c0_next = 0;
  const int wv64 = __builtin_amdgcn_readfirstlane(tid & ~63);       // the wave's first thread as a scalar: the LDS-DMA destination (M0) needs no per-piece v_readfirstlane
  auto dma_piece = [&](auto kc, auto stg) {
    constexpr int k = decltype(kc)::value;
    float* stage = smem + decltype(stg)::value * BUF;
    if constexpr (k < S_ITEMS) {
      if ((k + 1) * NT <= S_COUNT || tid + k * NT < S_COUNT)
        gn_buffer_load_lds(xsrd, (lptr_t)(stage + (k * NT + wv64) * 4), 16, soff[k], c0_next * 4, 0, 0);
    } else {
      constexpr int it = k - S_ITEMS;
      if ((it + 1) * NT <= W_TOTAL || wv64 + it * NT < W_TOTAL)
        gn_buffer_load_lds(wsrd, (lptr_t)(stage + SLAB + (it * NT + wv64) * 4), 16, woff[it], c0_next * a.Cout * 4, 0, 0);
    }
  };

  // chunk-invariant LDS byte addresses of this lane's operands in stage 0
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  unsigned addr_a[NTAPS];
#pragma unroll
  for (int j = 0; j < NTAPS; ++j) {
    const int d = a.t.off[j] - minoff;
    const int rowbase = (IS == 1) ? d : ((d & 1) * RPER + (d >> 1));
    const int row = rowbase + wm * WM * 32 + i32;
    addr_a[j] = lds0 + (row * 2 + (h ^ ((row >> 3) & 1))) * 16;                        // granule h of the row: channels 4h .. 4h+3
  }
  const unsigned addr_b = lds0 + (SLAB + 4 * h * TN + wn * WN * 32 + i32) * 4;         // weight rows 4h + s of the tap, s = k-step

  const int n_chunks = a.Cin / KC;
  PipeChunk<TN, KC, NTAPS, STAGE_BYTES, 1>::template issue_all<NPIECES>(dma_piece);       // chunk 0 into stage 0, in one burst
  __syncthreads();                                         // drains the LDS-DMA (vmcnt(0)) in front of the barrier

  for (int ch = 0; ch < n_chunks; ch += 2) {
    c0_next = min(ch + 1, n_chunks - 1) * KC;                                     // chunk ch+1 flies during this chunk's MFMAs
    if constexpr (MERGE == 1) PipeChunk<TN, KC, NTAPS, STAGE_BYTES, 0>::template chunk_m<NPIECES>(acc, accB, addr_a, addr_b, dma_piece);
    else if constexpr (MERGE == 2) {
      if (ph) PipeChunk<TN, KC, NTAPS, STAGE_BYTES, 0>::template chunk_p<1, NPIECES>(acc, addr_a, addr_b, dma_piece);
      else PipeChunk<TN, KC, NTAPS, STAGE_BYTES, 0>::template chunk_p<0, NPIECES>(acc, addr_a, addr_b, dma_piece);
    } else PipeChunk<TN, KC, NTAPS, STAGE_BYTES, 0>::template chunk<NPIECES>(acc, addr_a, addr_b, dma_piece);
    __syncthreads();
    if (ch + 1 < n_chunks) {
      c0_next = min(ch + 2, n_chunks - 1) * KC;
      if constexpr (MERGE == 1) PipeChunk<TN, KC, NTAPS, STAGE_BYTES, 1>::template chunk_m<NPIECES>(acc, accB, addr_a, addr_b, dma_piece);
      else if constexpr (MERGE == 2) {
        if (ph) PipeChunk<TN, KC, NTAPS, STAGE_BYTES, 1>::template chunk_p<1, NPIECES>(acc, addr_a, addr_b, dma_piece);
        else PipeChunk<TN, KC, NTAPS, STAGE_BYTES, 1>::template chunk_p<0, NPIECES>(acc, addr_a, addr_b, dma_piece);
      } else PipeChunk<TN, KC, NTAPS, STAGE_BYTES, 1>::template chunk<NPIECES>(acc, addr_a, addr_b, dma_piece);
      __syncthreads();
    }
  }
  // MFMA results written inside asm: the compiler inserts no wait states for its own readers of acc
  if constexpr (WN == 2) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
  else if constexpr (MERGE == 1) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0][0]), "+v"(acc[1][0]), "+v"(accB[0][0]), "+v"(accB[1][0]));
  else asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0][0]), "+v"(acc[1][0]));

  const int m_base = m0 + __builtin_amdgcn_readfirstlane(wm) * WM * 32, n_base = n0 + wn * WN * 32;
  const int mode = a.gy ? (a.gmask ? 3 : 2) : (a.mask ? 1 : 0);
  pipe_epilogue_dispatch<WN>(a, acc, b, m_base, n_base, i32, h, (MERGE == 2 && ph) ? a.t.out_off_odd : a.t.out_off, mode);
  if constexpr (MERGE == 1) pipe_epilogue_dispatch<1>(a, accB, b, m_base, n_base, i32, h, a.t.out_off_odd, mode);

  // BatchNorm statistics of the output on the way (the conv -> BatchNormalization layers of the generator): every lane sums its 32
  // values of each column in fp64 (rows past M excluded), the two lane halves and the waves stacked in M combine through a shuffle
  // and LDS (every wave is past the loop's last barrier, the stages are free), one fp64 partial per block and column
  if (a.stat_part) {
    double* red = reinterpret_cast<double*>(smem);
#pragma unroll
    for (int nt = 0; nt < WN; ++nt) {
      const int n = n_base + nt * 32 + i32;
      const float bias = a.bias ? a.bias[n] : 0.f;
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int mt = 0; mt < WM; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m_base + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (row < a.M) {
            const double v = (double)(acc[mt][nt][r] + bias);
            s1 += v; s2 += v * v;
          }
        }
      s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
      if (h == 0) {
        const int slot = (wm * TN + wn * WN * 32 + nt * 32 + i32) * 2;
        red[slot] = s1; red[slot + 1] = s2;
      }
    }
    __syncthreads();
    if (tid < TN) {
      double t1 = 0.0, t2 = 0.0;
#pragma unroll
      for (int w = 0; w < WAVES_M; ++w) { t1 += red[(w * TN + tid) * 2]; t2 += red[(w * TN + tid) * 2 + 1]; }
      double* dst = a.stat_part + (size_t)(b * m_tiles + m_tile) * 2 * a.Cout;
      dst[n0 + tid] = t1;
      dst[a.Cout + n0 + tid] = t2;
    }
  }
#endif
}

template <int WAVES_M, int WAVES_N, int NTAPS, int IS, int KC = 8, int WN = 2, int SPAN = NTAPS, int MERGE = 0>
static int launch_conv_pipe(const ConvArgs& a, hipStream_t s) {
  constexpr int TM = WAVES_M * 64, TN = WAVES_N * WN * 32;
  constexpr int R = IS * (TM - 1) + SPAN, RPER = (R + IS - 1) / IS;
  constexpr size_t lds = 2 * sizeof(float) * ((size_t)IS * RPER * KC + (size_t)NTAPS * KC * TN);
  static_assert(lds <= 160 * 1024, "stage too large");
  static_assert(lds / 2 + (NTAPS * KC + 8) * TN * 4 < 65536, "ds_read offsets must fit 16 bits");
  if (lds > 64 * 1024) {
    static unsigned long long lds_done = 0;
    allow_big_lds((const void*)conv_mfma_pipe_kernel<WAVES_M, WAVES_N, NTAPS, IS, KC, WN, SPAN, MERGE>, &lds_done);
  }
  const int m_tiles = (a.M + TM - 1) / TM, n_tiles = a.Cout / TN;
  const size_t blocks = (size_t)m_tiles * n_tiles * a.B;
  if (blocks == 0 || blocks > 0x7fffffffull) {
    set_error("conv_mfma_pipe: bad grid %zu", blocks);
    return GN_EINVAL;
  }
  static const bool no_patch = getenv("GN_CONV_NOPATCH") != nullptr;          // A/B switch
  int patch = -1;
  {
    const int pn = n_tiles % 8 == 0 ? 3 : (n_tiles == 4 ? 2 : -1);
    const int ng = pn >= 0 ? n_tiles >> pn : 0;
    if (!no_patch && pn >= 0 && ng <= 8 && (ng & (ng - 1)) == 0) patch = __builtin_ctz(ng) | (pn << 8);
  }
  prof_begin(s);
  hipLaunchKernelGGL((conv_mfma_pipe_kernel<WAVES_M, WAVES_N, NTAPS, IS, KC, WN, SPAN, MERGE>), dim3((unsigned)blocks), dim3(64 * WAVES_M * WAVES_N * (MERGE == 2 ? 2 : 1)), lds, s, a, m_tiles, n_tiles,
                     patch);
  prof_end(s, 2.0 * a.B * (double)a.M * a.t.ntaps * a.Cin * a.Cout, 0, 4.0 * ((double)a.B * a.Lin * a.Cin + (double)a.t.ntaps * a.Cin * a.Cout + (double)a.B * a.M * a.Cout));
  int rc = check_launch("conv_mfma_pipe");
  if (rc || !a.stat_part) return rc;
  *a.stat_done = 1;
  return colred_finalize(a.stat_part, a.stat_sums, (size_t)2 * a.Cout, a.B * m_tiles, s);
}


// Selection for conv_mfma_dispatch (conv_mfma.hip): tile (tall 256 x 64 or square 128 x 128), tap count, input stride.  *launched stays
// false when the shape is outside what the pipelined kernel covers (the caller then runs the plain DMA kernel).
int conv_pipe_try(const ConvArgs& a, bool tall, hipStream_t s, bool* launched) {
  static const bool no_pipe = getenv("GN_CONV_NOPIPE") != nullptr;      // A/B switch
  *launched = false;
  const int nt = a.t.ntaps;
  int minoff = a.t.off[0], maxoff = a.t.off[0];
  for (int j = 1; j < nt; ++j) {
    minoff = std::min(minoff, a.t.off[j]);
    maxoff = std::max(maxoff, a.t.off[j]);
  }
  if (no_pipe || nt < 2 || nt > 5 || maxoff - minoff + 1 > nt || (size_t)a.Ly * a.Cout * 4 >= 0x40000000ull ||
      (size_t)a.Lin * a.Cin * 4 >= 0x40000000ull)      // the input descriptor's byte count and its out-of-range sentinel offset 0x40000000 must stay apart
    return GN_OK;
  if (a.t.in_stride != 1 && !(a.t.in_stride == 2 && nt == 5)) return GN_OK;
  *launched = true;
  if (a.stat_part && (a.act != GN_ACT_LINEAR || a.mask || a.gy || a.t.out_stride != 1)) {   // statistics are defined for the plain linear forward only
    ConvArgs a2 = a;
    a2.stat_part = nullptr;
    return conv_pipe_try(a2, tall, s, launched);
  }
  // Under-filled grids (the script's own batch 8 / n_pix 1024: the q-branch's last convolution is 128 tall blocks for 256 CUs), measured in
  // round 3 at batch 8 (CNN step / GAN iteration, ms; baseline 2.34 / 4.85): smaller BLOCKS of full 64 x 64 waves (128 x 64 with 2 waves,
  // 64 x 64 with 1) 2.31 / 5.36 -- more CUs but the same number of SIMDs; narrow WAVES (64 x 32) in the unchanged block tile 2.30 / 4.89 --
  // more waves on the same CUs; narrow waves with 16-channel chunks 3.06 / 6.34.
  // What fills the chip is both at once: NARROW waves (64 x 32 each, pipe_group_n) in SMALLER blocks of 64 columns -- 8, 4 or 2 waves for 256,
  // 128 or 64 rows -- so that a launch with fewer than two 64 x 64 wave tiles per SIMD gets at least one block per CU (256 blocks: 1.62 /
  // 4.28; requiring two, 512 blocks, 1.71 / 4.54; 128 blocks 1.78 / 4.67), and every block still spreads over the CU's SIMDs.
  static const bool no_narrow = getenv("GN_CONV_NONARROW") != nullptr;          // A/B switch
  constexpr int narrow_below = 2048;
  const size_t wave_tiles = (size_t)a.B * (size_t)((a.M + 63) / 64) * (size_t)(a.Cout / 64);
  const bool narrow_wave = !no_narrow && a.Cout % 64 == 0 && wave_tiles < (size_t)narrow_below;
  int nwm = 4;
  if (narrow_wave) {
    auto blocks_of = [&](int wm_) { return (size_t)a.B * (size_t)((a.M + 64 * wm_ - 1) / (64 * wm_)) * (size_t)(a.Cout / 64); };
    constexpr int min_blocks = 256;
    while (nwm > 1 && blocks_of(nwm) < (size_t)min_blocks) nwm >>= 1;
  }
#define GN_PIPE(NT_, IS_)                                                                    \
  do {                                                                                       \
    if (narrow_wave) {                                                                       \
      if (nwm == 4) return launch_conv_pipe<4, 2, NT_, IS_, 8, 1>(a, s);                     \
      if (nwm == 2) return launch_conv_pipe<2, 2, NT_, IS_, 8, 1>(a, s);                     \
      return launch_conv_pipe<1, 2, NT_, IS_, 8, 1>(a, s);                                   \
    }                                                                                        \
    return tall ? launch_conv_pipe<4, 1, NT_, IS_>(a, s) : launch_conv_pipe<2, 2, NT_, IS_>(a, s);                       \
  } while (0)
  // (256 x 128 blocks of 8 waves for the stride-2 forward -- 74 KiB of LDS, two blocks per CU = a fourth wave per SIMD -- measured in round 3:
  // -2.5 to -4 % per launch in the layer sweep, nothing on the step (1420.0 / 1420.6 against 1420.8 / 1419.1 waveforms/s); the same blocks on the
  // stride-1 5-tap launches and on the 2- / 3-tap phases LOSE 0.3 % of the step.  Removed in round 5.)
  // (4-channel chunks for the stride-2 forward -- 27 instead of 53 KiB of LDS, a fourth block per CU, but a barrier per 40 MFMAs -- were
  // measured: 140.3 -> 135.5 TFLOP/s.  Eight channels per chunk is the optimum in both directions.)
  if (a.t.in_stride == 2) { GN_PIPE(5, 2); }
  // (16-channel chunks for the 2- / 3-tap square tile -- twice the MFMAs per barrier, 49 / 66 KiB of LDS -- were measured on the
  // stride-2 data gradient: 140.7 -> 139.3 / 136.9 TFLOP/s; the lost block per CU costs more than the barriers. KC stays 8.)
  switch (nt) {
    case 2: GN_PIPE(2, 1);
    case 3: GN_PIPE(3, 1);
    case 4: GN_PIPE(4, 1);
    default: GN_PIPE(5, 1);
  }
#undef GN_PIPE
}

// Both output phases of a stride-2, 5-tap data gradient in ONE launch (round 3).  The two phases read the same three dy rows per output pair
// (phase of tap kk alternates with kk), so the block stages the slab once and every K-chunk carries all five taps: narrow waves with two
// accumulator sets chosen by tap parity (PipeChunk::run_m, two epilogues), or twice the waves with one phase each (PipeChunk::run_p).  Two separate launches of 3 and 2 taps each walk every channel
// chunk; at the script's own batch 8 that made the data gradient of a stride-2 layer twice as slow as its forward.
// a.t: ntaps 5 in kernel-tap order (even index <-> rows out_stride*m + out_off, odd index <-> out_off_odd), offsets spanning 3 rows.
int conv_pipe_try_merged(const ConvArgs& a, hipStream_t s, bool* launched) {
  // A/B switches: the merged kernel is a member of the pipelined LDS-DMA family, so every switch that takes that family out (to run and test
  // the fallback kernels) takes it out too
  static const bool off = getenv("GN_CONV_NOMERGE") != nullptr || getenv("GN_CONV_NOPIPE") != nullptr || getenv("GN_CONV_NODMA") != nullptr;
  constexpr int merge_below = 2048;
  *launched = false;
  if (off || a.t.ntaps != 5 || a.t.in_stride != 1 || a.t.out_stride != 2 || a.stat_part || a.mask || a.bias) return GN_OK;
  int minoff = a.t.off[0], maxoff = a.t.off[0];
  for (int j = 1; j < 5; ++j) {
    minoff = std::min(minoff, a.t.off[j]);
    maxoff = std::max(maxoff, a.t.off[j]);
  }
  if (maxoff - minoff + 1 > 3 || a.Cin % 8 || a.Cout % 64 || (size_t)a.Ly * a.Cout * 4 >= 0x40000000ull || (size_t)a.Lin * a.Cin * 4 >= 0x40000000ull) return GN_OK;
  const size_t wave_tiles = (size_t)a.B * (size_t)((a.M + 63) / 64) * (size_t)(a.Cout / 64) * 2;       // 64 x 64 tiles of both phases
  if (wave_tiles >= (size_t)merge_below) return GN_OK;
  auto blocks_of = [&](int wm_) { return (size_t)a.B * (size_t)((a.M + 64 * wm_ - 1) / (64 * wm_)) * (size_t)(a.Cout / 64); };
  int nwm = 4;
  while (nwm > 1 && blocks_of(nwm) < 256) nwm >>= 1;
  *launched = true;
  // 256-row blocks: 8 narrow waves, each both phases (two accumulator sets); smaller blocks: twice the waves, each ONE phase and only the taps
  // of its parity -- the merged wave is a 64 x 64 x 2-phase tile, which left half the SIMDs idle at batch 8.  Measured at batch 8 (us per
  // launch, both-phase waves -> phase-split waves): PE q 512 -> 1024 data gradient 162 -> 111 (its forward: 94), PE q 256 -> 512 84 -> 58,
  // CNN step 1.47 -> 1.32 ms; the discriminator's 256-row launch 180 -> 186, so it keeps the two-set form.
  if (nwm == 4) return launch_conv_pipe<4, 2, 5, 1, 8, 1, 3, 1>(a, s);
  if (nwm == 2) return launch_conv_pipe<2, 2, 5, 1, 8, 1, 3, 2>(a, s);
  return launch_conv_pipe<1, 2, 5, 1, 8, 1, 3, 2>(a, s);
}

}  // namespace gn
