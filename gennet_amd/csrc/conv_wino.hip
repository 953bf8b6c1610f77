// Transform-domain (Cook-Toom / Winograd F(2,5)) fp32 Conv1D for the unit-stride 5-tap layers: forward and data gradient.
//
//   y[2t + i, n] = sum_p AT[i][p] * ( sum_c V_p[t, c] * U_p[c, n] ),   V_p[t, c] = sum_j BT[p][j] x[2t + off0 + j, c],   U_p = sum_q G[p][q] w_q
//
// six multiplies per two outputs instead of ten: 0.6 of the direct kernel's matrix-core work for the layers that carry most of the step
// (generator 128 -> 256 -> 512 -> 1024, bbhMahoGANy.py:259-283; PE q branch 64 -> 128 -> 256, :382-386).  Points {0, 1, -1, 1/2, -2, inf}:
// the set whose fp32 error stays closest to the direct k-ordered fma chain's (profiles/r05_winograd_gate1.txt: 1.4x rms on the forward map).
// Operands stay fp32 and every product runs on v_mfma_f32_32x32x2_f32 (exact fp32 fma chains in the transform domain): dtype f32.
//
// Structure = conv_pipe.hip's with the input side of a stride-2, 6-tap convolution (tile t reads rows 2t .. 2t+5: the slab is staged as an even-row
// and an odd-row plane, same conflict-free granule swap) and the six "taps" kept apart: a wave owns 32 tiles (64 output rows) x 32 columns x 6
// points = six accumulator tiles; per 8-channel chunk it reads its six raw row fragments (ds_read_b128: four channels each), forms the six
// transformed fragments in registers (26 fma per channel) and issues 24 MFMAs against the staged U tile.  THREE LDS stages: the raw fragments of
// chunk c+1 are read and transformed underneath the MFMAs of chunk c, so no wave waits for the LDS or the VALU after a barrier.
// The epilogue applies AT in registers (even rows: sum of points 0..4; odd rows: p1 - p2 + p3/2 - 2 p4 + p5) and hands the two 32 x 32 tiles to
// the shared lean epilogue (bias, activation, dropout, fused backward, BatchNorm statistics) with an output row stride of 2.
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include "common.h"
#include "conv_epilogue.h"

namespace gn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct WinoTaps {
  int wq[5];      // kernel index (into w's leading axis) of the tap at offset off0 + q
};

// U[p][ci][co] = sum_q G[p][q] * w[wq[q]][ci][co], formed in fp64 and rounded once.
__global__ void wino_u_kernel(const float* __restrict__ w, float* __restrict__ U, size_t cc, WinoTaps t) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cc) return;
  double g[5];
#pragma unroll
  for (int q = 0; q < 5; ++q) g[q] = (double)w[(size_t)t.wq[q] * cc + i];
  U[i] = (float)(0.5 * g[0]);
  U[cc + i] = (float)((g[0] + g[1] + g[2] + g[3] + g[4]) * (1.0 / 6.0));
  U[2 * cc + i] = (float)((g[0] - g[1] + g[2] - g[3] + g[4]) * (1.0 / 6.0));
  U[3 * cc + i] = (float)((16.0 * g[0] + 8.0 * g[1] + 4.0 * g[2] + 2.0 * g[3] + g[4]) * (1.0 / 15.0));
  U[4 * cc + i] = (float)((g[0] - 2.0 * g[1] + 4.0 * g[2] - 8.0 * g[3] + 16.0 * g[4]) * (1.0 / 30.0));
  U[5 * cc + i] = (float)(0.5 * g[4]);
}

// ---------------------------------------------------------------------------------------------
// Hand-scheduled chunk (the reason is conv_pipe.hip's: hipcc sinks the LDS reads next to their use and drains lgkmcnt in front of every short
// MFMA group; here it also put the whole transform behind the chunk's last MFMA).  A chunk of a wave = 24 MFMA slots (point p = slot / 4,
// k-step s = slot % 4), each one asm statement:
//   slot 4p     issues the two paired reads of the NEXT group's four U values (ds_read2st64_b32: rows s, s+1 of a point are 256 bytes apart; after
//               the last group: group 0 of the next chunk, whose stage has been complete since the previous barrier), slot 0 also this lane's six
//               raw row fragments of the NEXT chunk; a counted lgkmcnt retires exactly what the slot consumes (LDS returns in order);
//   slots >= 4  are followed by one or two of the 24 (point, channel) values of the next chunk's transformed fragments: ~5 VALU per MFMA, under it;
//   the staging pieces of chunk + 2 go behind the first slots.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void wino_slot(f32x16& c, float a, float b) {
  asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b) : "memory");
}
template <int O, int WAIT>
__device__ __forceinline__ void wino_slot_rb(f32x16& c, float a, float b, f32x2& nb01, f32x2& nb23, unsigned addr_b) {
  asm volatile(
      "ds_read2st64_b32 %1, %5 offset0:%6 offset1:%7\n\t"
      "ds_read2st64_b32 %2, %5 offset0:%8 offset1:%9\n\t"
      "s_waitcnt lgkmcnt(%10)\n\t"
      "v_mfma_f32_32x32x2_f32 %0, %3, %4, %0"
      : "+v"(c), "=&v"(nb01), "=&v"(nb23)
      : "v"(a), "v"(b), "v"(addr_b), "i"(O), "i"(O + 1), "i"(O + 2), "i"(O + 3), "i"(WAIT)
      : "memory");
}
// ... and the raw fragments of the next chunk (slot 0)
template <int O>
__device__ __forceinline__ void wino_slot_rba(f32x16& c, float a, float b, f32x2& nb01, f32x2& nb23, unsigned addr_b, f32x4 (&d)[6], const unsigned (&addr_a)[6]) {
  asm volatile(
      "ds_read2st64_b32 %1, %11 offset0:%18 offset1:%19\n\t"
      "ds_read2st64_b32 %2, %11 offset0:%20 offset1:%21\n\t"
      "ds_read_b128 %3, %12\n\t"
      "ds_read_b128 %4, %13\n\t"
      "ds_read_b128 %5, %14\n\t"
      "ds_read_b128 %6, %15\n\t"
      "ds_read_b128 %7, %16\n\t"
      "ds_read_b128 %8, %17\n\t"
      "s_waitcnt lgkmcnt(8)\n\t"                        // this group's U values were read across the barrier: nothing else retires them
      "v_mfma_f32_32x32x2_f32 %0, %9, %10, %0"
      : "+v"(c), "=&v"(nb01), "=&v"(nb23), "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5])
      : "v"(a), "v"(b), "v"(addr_b), "v"(addr_a[0]), "v"(addr_a[1]), "v"(addr_a[2]), "v"(addr_a[3]), "v"(addr_a[4]), "v"(addr_a[5]), "i"(O), "i"(O + 1), "i"(O + 2),
        "i"(O + 3)
      : "memory");
}
// ... slot 4: its wait also retires the raw fragments (issued before the reads this slot adds); they are operands so that their readers depend on it
template <int O>
__device__ __forceinline__ void wino_slot_rbw(f32x16& c, float a, float b, f32x2& nb01, f32x2& nb23, unsigned addr_b, f32x4 (&d)[6]) {
  asm volatile(
      "ds_read2st64_b32 %1, %11 offset0:%12 offset1:%13\n\t"
      "ds_read2st64_b32 %2, %11 offset0:%14 offset1:%15\n\t"
      "s_waitcnt lgkmcnt(2)\n\t"
      "v_mfma_f32_32x32x2_f32 %0, %9, %10, %0"
      : "+v"(c), "=&v"(nb01), "=&v"(nb23), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5])
      : "v"(a), "v"(b), "v"(addr_b), "i"(O), "i"(O + 1), "i"(O + 2), "i"(O + 3)
      : "memory");
}

// one (point, channel) value of the transformed fragment, as asm: hipcc moves plain fma code away from the MFMA slots it is meant to sit under
// (all of it in front of the chunk or behind it) and packs channel pairs into v_pk_fma_f32, which costs the matrix pipe more than two v_fma_f32
// (MI355X_MICROARCH.md, cycle constants).  3 and 5 are not inline constants: VOP2 literals.
template <int P>
__device__ __forceinline__ void wino_piece(float& o, float d0, float d1, float d2, float d3, float d4, float d5) {
  if constexpr (P == 0)
    asm volatile("v_add_f32 %0, %5, %5\n\tv_fmac_f32 %0, 0x40400000, %4\n\tv_fmac_f32 %0, -4.0, %3\n\tv_fmac_f32 %0, 0xc0400000, %2\n\tv_fmac_f32 %0, 2.0, %1"
                 : "=&v"(o) : "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4));
  else if constexpr (P == 1)
    asm volatile("v_fma_f32 %0, %4, 2.0, %2\n\tv_fmac_f32 %0, 0x40a00000, %3\n\tv_fmac_f32 %0, -2.0, %1" : "=&v"(o) : "v"(d1), "v"(d2), "v"(d3), "v"(d4));
  else if constexpr (P == 2)
    asm volatile("v_fma_f32 %0, %4, -2.0, -%3\n\tv_fmac_f32 %0, 0x40a00000, %2\n\tv_fmac_f32 %0, -2.0, %1" : "=&v"(o) : "v"(d1), "v"(d2), "v"(d3), "v"(d4));
  else if constexpr (P == 3)
    asm volatile("v_sub_f32 %0, %2, %4\n\tv_fmac_f32 %0, -2.0, %3\n\tv_fmac_f32 %0, 2.0, %1" : "=&v"(o) : "v"(d1), "v"(d2), "v"(d3), "v"(d4));
  else if constexpr (P == 4)
    asm volatile("v_sub_f32 %0, %1, %3\n\tv_fmac_f32 %0, 2.0, %4\n\tv_fmac_f32 %0, -2.0, %2" : "=&v"(o) : "v"(d1), "v"(d2), "v"(d3), "v"(d4));
  else
    asm volatile("v_add_f32 %0, %5, %5\n\tv_fmac_f32 %0, 0x40400000, %4\n\tv_fmac_f32 %0, -4.0, %3\n\tv_fmac_f32 %0, 0xc0400000, %2\n\tv_fmac_f32 %0, 2.0, %1"
                 : "=&v"(o) : "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5));
}
__device__ __forceinline__ void wino_bt_all(const f32x4 (&d)[6], float (&v)[6][4]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    wino_piece<0>(v[0][e], d[0][e], d[1][e], d[2][e], d[3][e], d[4][e], d[5][e]);
    wino_piece<1>(v[1][e], d[0][e], d[1][e], d[2][e], d[3][e], d[4][e], d[5][e]);
    wino_piece<2>(v[2][e], d[0][e], d[1][e], d[2][e], d[3][e], d[4][e], d[5][e]);
    wino_piece<3>(v[3][e], d[0][e], d[1][e], d[2][e], d[3][e], d[4][e], d[5][e]);
    wino_piece<4>(v[4][e], d[0][e], d[1][e], d[2][e], d[3][e], d[4][e], d[5][e]);
    wino_piece<5>(v[5][e], d[0][e], d[1][e], d[2][e], d[3][e], d[4][e], d[5][e]);
  }
}

template <int TN, int ABL = 0>
struct WinoChunk {
  static_assert(TN == 64, "ds_read2st64_b32 pairs two U rows 256 bytes apart");
  // transform pieces of the next chunk behind slot I: the 24 (point, channel) values over slots 4 .. 23
  template <int I>
  static __device__ __forceinline__ void pieces(const f32x4 (&d)[6], float (&vn)[6][4]) {
    if constexpr (I >= 4) {
      constexpr int lo = (I - 4) * 24 / 20, hi = (I - 3) * 24 / 20;
      if constexpr (lo < hi) piece<lo>(d, vn);
      if constexpr (lo + 1 < hi) piece<lo + 1>(d, vn);
    }
  }
  template <int K>
  static __device__ __forceinline__ void piece(const f32x4 (&d)[6], float (&vn)[6][4]) {
    constexpr int P = K / 4, E = K % 4;
    wino_piece<P>(vn[P][E], d[0][E], d[1][E], d[2][E], d[3][E], d[4][E], d[5][E]);
  }
  template <int NPIECES, int I = 0, class D>
  static __device__ __forceinline__ void run(f32x16 (&acc)[6], const float (&v)[6][4], float (&vn)[6][4], f32x4 (&d)[6], f32x2 (&B)[2][2], unsigned addr_b,
                                             unsigned addr_b_next, const unsigned (&addr_a)[6], D& dma) {
    if constexpr (I < 24) {
      constexpr int P = I / 4, S = I % 4;
      f32x2(&bc)[2] = B[P & 1];
      f32x2(&bn)[2] = B[(P + 1) & 1];
      const float bv = bc[S >> 1][S & 1];
      if constexpr ((ABL & 8) != 0) wino_slot(acc[P], v[P][S], bv);
      else if constexpr (I == 0) wino_slot_rba<8>(acc[0], v[0][0], bv, bn[0], bn[1], addr_b, d, addr_a);
      else if constexpr (I == 4) wino_slot_rbw<16>(acc[1], v[1][0], bv, bn[0], bn[1], addr_b, d);
      else if constexpr (I == 20) wino_slot_rb<0, 2>(acc[5], v[5][0], bv, bn[0], bn[1], addr_b_next);
      else if constexpr (S == 0) wino_slot_rb<(P + 1) * 8, 2>(acc[P], v[P][0], bv, bn[0], bn[1], addr_b);
      else wino_slot(acc[P], v[P][S], bv);
      if constexpr (I >= 1 && I - 1 < NPIECES) dma(std::integral_constant<int, I - 1>{});
      if constexpr (!(ABL & 2)) pieces<I>(d, vn);
      run<NPIECES, I + 1>(acc, v, vn, d, B, addr_b, addr_b_next, addr_a, dma);
    }
  }
};

// ABL: timing ablations, compiled only under -DGN_ABLATION (never into the shipped library; results are wrong for ABL != 0): bit 0 = no staging in the
// loop, bit 1 = no transform, bit 2 = no barrier, bit 3 = no LDS reads in the loop
template <int WAVES_M, int WAVES_N, int ABL = 0>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, (64 * WAVES_M * WAVES_N > 256 ? 1 : 2)) void conv_wino_kernel(ConvArgs a, const float* __restrict__ U, int off0,
                                                                                                                   int m_tiles, int n_tiles, int patch) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int KC = 8, NP = 6;
  constexpr int TT = WAVES_M * 32;                 // tiles (output row pairs) per block
  constexpr int TN = WAVES_N * 32;
  constexpr int NT = 64 * WAVES_M * WAVES_N;
  constexpr int RPER = TT + 2;                     // rows per parity plane: the block reads input rows 0 .. 2 TT + 3 of its window
  constexpr int SLAB = 2 * RPER * KC;              // floats
  constexpr int BUF = SLAB + NP * KC * TN;
  constexpr int STAGE_BYTES = BUF * 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  typedef __attribute__((address_space(3))) void* lptr_t;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int i32 = lane & 31, h = lane >> 5;
  // block -> tile: the XCD patch order of conv_pipe.hip
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
  const int t0 = m_tile * TT, n0 = n_tile * TN;

  f32x16 acc[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

  const int t_base = 2 * t0 + off0;                // input row of slab row 0
  const uintptr_t xbp = (uintptr_t)(a.x + (size_t)b * a.Lin * a.Cin);
  const unsigned xb_lo = __builtin_amdgcn_readfirstlane((unsigned)xbp), xb_hi = __builtin_amdgcn_readfirstlane((unsigned)(xbp >> 32));
  const int xbytes = __builtin_amdgcn_readfirstlane(a.Lin * a.Cin * 4);
  const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)xb_hi << 32) | xb_lo), 0, xbytes, 0x00020000);

  constexpr int S_COUNT = SLAB / 4;                // 16-byte granules of one slab stage, in LDS order
  constexpr int S_ITEMS = (S_COUNT + NT - 1) / NT;
  constexpr int W_TOTAL = NP * KC * (TN / 4);
  constexpr int W_ITEMS = (W_TOTAL + NT - 1) / NT;
  int soff[S_ITEMS];
#pragma unroll
  for (int it = 0; it < S_ITEMS; ++it) {
    const int id = tid + it * NT;
    const int lr = id / 2, c4 = (id % 2) ^ ((lr >> 3) & 1);                // the row's two granules swapped in LDS rows 8..15 mod 16
    const int r = lr < RPER ? 2 * lr : 2 * (lr - RPER) + 1;                // plane 0: even rows, plane 1: odd rows
    soff[it] = (id < S_COUNT) ? ((t_base + r) * a.Cin + 4 * c4) * 4 : 0x40000000;   // rows outside [0, Lin): the descriptor returns 0
  }
  const uintptr_t wbp = (uintptr_t)U;
  const unsigned wb_lo = __builtin_amdgcn_readfirstlane((unsigned)wbp), wb_hi = __builtin_amdgcn_readfirstlane((unsigned)(wbp >> 32));
  const int wbytes = __builtin_amdgcn_readfirstlane(NP * a.Cin * a.Cout * 4);
  const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)wb_hi << 32) | wb_lo), 0, wbytes, 0x00020000);
  int woff[W_ITEMS];
#pragma unroll
  for (int it = 0; it < W_ITEMS; ++it) {
    const int id = min(tid + it * NT, W_TOTAL - 1);
    const int n4 = id % (TN / 4);
    const int kk = (id / (TN / 4)) % KC;
    const int p = id / ((TN / 4) * KC);
    woff[it] = ((p * a.Cin + kk) * a.Cout + n0 + 4 * n4) * 4;
  }
  int c0_next = 0, st_next = 0;
  bool in_loop = false;
  auto dma_piece = [&](auto kc) {
    constexpr int k = decltype(kc)::value;
    if ((ABL & 1) && in_loop) return;
    float* stg = smem + st_next * BUF;
    if constexpr (k < S_ITEMS) {
      if ((k + 1) * NT <= S_COUNT || tid + k * NT < S_COUNT)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (lptr_t)(stg + (k * NT + (tid & ~63)) * 4), 16, soff[k], c0_next * 4, 0, 0);
    } else {
      constexpr int it = k - S_ITEMS;
      if ((it + 1) * NT <= W_TOTAL || (tid & ~63) + it * NT < W_TOTAL)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, (lptr_t)(stg + SLAB + (it * NT + (tid & ~63)) * 4), 16, woff[it], c0_next * a.Cout * 4, 0, 0);
    }
  };
  constexpr int NPIECES = S_ITEMS + W_ITEMS;
  auto dma_all = [&]() {
    dma_piece(std::integral_constant<int, 0>{});
    if constexpr (NPIECES > 1) dma_piece(std::integral_constant<int, 1>{});
    if constexpr (NPIECES > 2) dma_piece(std::integral_constant<int, 2>{});
    if constexpr (NPIECES > 3) dma_piece(std::integral_constant<int, 3>{});
    if constexpr (NPIECES > 4) dma_piece(std::integral_constant<int, 4>{});
    if constexpr (NPIECES > 5) dma_piece(std::integral_constant<int, 5>{});
    static_assert(NPIECES <= 6, "staging pieces");
  };

  // chunk-invariant byte addresses of this lane's operands in stage 0
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  unsigned base_a[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int row = (j & 1) * RPER + (j >> 1) + wm * 32 + i32;
    base_a[j] = lds0 + (row * 2 + (h ^ ((row >> 3) & 1))) * 16;            // granule h of the row: channels 4h .. 4h+3
  }
  const unsigned base_b = lds0 + (SLAB + 4 * h * TN + wn * 32 + i32) * 4;  // U rows 4h + s of a point, s = k-step

  const int n_chunks = a.Cin / KC;
  c0_next = 0; st_next = 0; dma_all();
  c0_next = min(1, n_chunks - 1) * KC; st_next = 1; dma_all();
  __syncthreads();                                  // drains the LDS-DMA (vmcnt(0)) in front of the barrier

  float V0[6][4], V1[6][4];
  f32x4 d[6];
  f32x2 Bq[2][2];
  {
    const char* sb = reinterpret_cast<const char*>(smem);
#pragma unroll
    for (int j = 0; j < 6; ++j) d[j] = *reinterpret_cast<const f32x4*>(sb + (base_a[j] - lds0));
    wino_bt_all(d, V0);
    const float* bp = reinterpret_cast<const float*>(sb + (base_b - lds0));
    Bq[0][0][0] = bp[0]; Bq[0][0][1] = bp[TN]; Bq[0][1][0] = bp[2 * TN]; Bq[0][1][1] = bp[3 * TN];
  }
  int st = 0;
  in_loop = true;
  for (int ch = 0; ch < n_chunks; ch += 2) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (half == 1 && ch + 1 >= n_chunks) break;
      const int st1 = st == 2 ? 0 : st + 1, st2 = st1 == 2 ? 0 : st1 + 1;
      c0_next = min(ch + half + 2, n_chunks - 1) * KC;                      // stage st2 held chunk - 1: every wave is past its last read of it
      st_next = st2;
      unsigned addr_a[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) addr_a[j] = base_a[j] + st1 * STAGE_BYTES;     // chunk + 1 landed before the previous barrier
      const unsigned addr_b = base_b + st * STAGE_BYTES, addr_b_next = base_b + st1 * STAGE_BYTES;
      if (half == 0) WinoChunk<TN, ABL>::template run<NPIECES>(acc, V0, V1, d, Bq, addr_b, addr_b_next, addr_a, dma_piece);
      else WinoChunk<TN, ABL>::template run<NPIECES>(acc, V1, V0, d, Bq, addr_b, addr_b_next, addr_a, dma_piece);
      if constexpr (!(ABL & 4)) __syncthreads();
      st = st1;
    }
  }
  // MFMA results written inside asm: the compiler inserts no wait states for its own readers of acc
  asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]));

  // AT: even rows = p0 + p1 + p2 + p3 + p4, odd rows = p1 - p2 + p3 / 2 - 2 p4 + p5
  f32x16 out[2][1][1];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    out[0][0][0][r] = (((acc[0][r] + acc[1][r]) + acc[2][r]) + acc[3][r]) + acc[4][r];
    out[1][0][0][r] = __builtin_fmaf(-2.f, acc[4][r], __builtin_fmaf(0.5f, acc[3][r], acc[1][r] - acc[2][r])) + acc[5][r];
  }
  const int m_base = t0 + __builtin_amdgcn_readfirstlane(wm) * 32, n_base = n0 + wn * 32;
  const int mode = a.gy ? (a.gmask ? 3 : 2) : (a.mask ? 1 : 0);
  ConvArgs a2 = a;
  a2.t.out_stride = 2;
  pipe_epilogue_dispatch<1, 1>(a2, out[0], b, m_base, n_base, i32, h, a.t.out_off, mode);
  pipe_epilogue_dispatch<1, 1>(a2, out[1], b, m_base, n_base, i32, h, a.t.out_off + 1, mode);

  if (a.stat_part) {                                 // BatchNorm statistics of the output on the way, as in conv_pipe.hip
    double* red = reinterpret_cast<double*>(smem);
    const int n = n_base + i32;
    const float bias = a.bias ? a.bias[n] : 0.f;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 2 * (m_base + (r & 3) + 8 * (r >> 2) + 4 * h) + ph;
        if (row < a.M) {
          const double v = (double)(out[ph][0][0][r] + bias);
          s1 += v; s2 += v * v;
        }
      }
    s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
    if (h == 0) {
      const int slot = (wm * TN + wn * 32 + i32) * 2;
      red[slot] = s1; red[slot + 1] = s2;
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

template <int WAVES_M, int WAVES_N, int ABL = 0>
static int launch_conv_wino(const ConvArgs& a, const float* U, int off0, hipStream_t s) {
  constexpr int TT = WAVES_M * 32, TN = WAVES_N * 32;
  constexpr size_t lds = 3 * sizeof(float) * ((size_t)2 * (TT + 2) * 8 + (size_t)6 * 8 * TN);
  static_assert(lds <= 160 * 1024, "stages too large");
  if (lds > 64 * 1024) {
    static unsigned long long lds_done = 0;
    allow_big_lds((const void*)conv_wino_kernel<WAVES_M, WAVES_N, ABL>, &lds_done);
  }
  const int tiles = (a.M + 1) / 2;
  const int m_tiles = (tiles + TT - 1) / TT, n_tiles = a.Cout / TN;
  const size_t blocks = (size_t)m_tiles * n_tiles * a.B;
  if (blocks == 0 || blocks > 0x7fffffffull) {
    set_error("conv_wino: bad grid %zu", blocks);
    return GN_EINVAL;
  }
  int patch = -1;
  {
    const int pn = n_tiles % 8 == 0 ? 3 : (n_tiles == 4 ? 2 : -1);
    const int ng = pn >= 0 ? n_tiles >> pn : 0;
    if (pn >= 0 && ng <= 8 && (ng & (ng - 1)) == 0) patch = __builtin_ctz(ng) | (pn << 8);
  }
  prof_begin(s);
  hipLaunchKernelGGL((conv_wino_kernel<WAVES_M, WAVES_N, ABL>), dim3((unsigned)blocks), dim3(64 * WAVES_M * WAVES_N), lds, s, a, U, off0, m_tiles, n_tiles, patch);
  // flop = the ALGORITHMIC count of the convolution (10 multiplies per output pair and channel pair); the kernel executes 0.6 of it
  prof_end(s, 2.0 * a.B * (double)a.M * 5 * a.Cin * a.Cout, 5, 4.0 * ((double)a.B * a.Lin * a.Cin + 5.0 * a.Cin * a.Cout + (double)a.B * a.M * a.Cout));
  int rc = check_launch("conv_wino");
  if (rc || !a.stat_part) return rc;
  *a.stat_done = 1;
  return colred_finalize(a.stat_part, a.stat_sums, (size_t)2 * a.Cout, a.B * m_tiles, s);
}

size_t conv_wino_workspace_bytes(int Cin, int Cout) { return (size_t)6 * Cin * Cout * sizeof(float); }

bool conv_wino_supported(const ConvArgs& a) {
  if (a.t.ntaps != 5 || a.t.in_stride != 1 || a.t.out_stride != 1 || a.t.out_off != 0) return false;
  if (a.Cin % 8 || a.Cout % 64 || a.Cin < 8) return false;
  if ((size_t)a.Ly * a.Cout * 4 >= 0x40000000ull || (size_t)a.Lin * a.Cin * 4 >= 0x40000000ull || (size_t)6 * a.Cin * a.Cout * 4 >= 0x40000000ull) return false;
  bool seen[5] = {false, false, false, false, false};
  int minoff = a.t.off[0];
  for (int j = 1; j < 5; ++j) minoff = std::min(minoff, a.t.off[j]);
  for (int j = 0; j < 5; ++j) {
    const int q = a.t.off[j] - minoff;
    if (q < 0 || q > 4 || seen[q]) return false;
    seen[q] = true;
  }
  if (a.stat_part && (a.act != GN_ACT_LINEAR || a.mask || a.gy)) return false;
  return true;
}

// ws: conv_wino_workspace_bytes(Cin, Cout) of device memory for the transformed kernel of THIS launch (stream-ordered reuse)
int conv_wino_run(const ConvArgs& a, void* ws, size_t ws_bytes, hipStream_t s) {
  if (!conv_wino_supported(a)) {
    set_error("conv_wino: unsupported shape (5 consecutive taps, unit strides, Cin %% 8 == 0, Cout %% 64 == 0)");
    return GN_EINVAL;
  }
  if (!ws || ws_bytes < conv_wino_workspace_bytes(a.Cin, a.Cout)) {
    set_error("conv_wino: workspace too small (%zu < %zu)", ws_bytes, conv_wino_workspace_bytes(a.Cin, a.Cout));
    return GN_EWORKSPACE;
  }
  int minoff = a.t.off[0];
  for (int j = 1; j < 5; ++j) minoff = std::min(minoff, a.t.off[j]);
  WinoTaps t;
  for (int j = 0; j < 5; ++j) t.wq[a.t.off[j] - minoff] = a.t.widx[j];
  const size_t cc = (size_t)a.Cin * a.Cout;
  hipLaunchKernelGGL(wino_u_kernel, dim3(cdiv(cc, 256)), dim3(256), 0, s, a.w, (float*)ws, cc, t);
  int rc = check_launch("wino_u");
  if (rc) return rc;
  const int tiles = (a.M + 1) / 2;
  static const int force = getenv("GN_WINO_TILE") ? atoi(getenv("GN_WINO_TILE")) : 0;      // A/B switch: 1 = 2x2 waves, 2 = 4x2, 3 = 4x1... (development)
#ifdef GN_ABLATION
  {
    static const int abl = getenv("GN_WINO_ABL") ? atoi(getenv("GN_WINO_ABL")) : 0;
    const float* Up = (const float*)ws;
    switch (abl) {
      case 1: return launch_conv_wino<2, 2, 1>(a, Up, minoff, s);
      case 2: return launch_conv_wino<2, 2, 2>(a, Up, minoff, s);
      case 3: return launch_conv_wino<2, 2, 3>(a, Up, minoff, s);
      case 4: return launch_conv_wino<2, 2, 4>(a, Up, minoff, s);
      case 5: return launch_conv_wino<2, 2, 5>(a, Up, minoff, s);
      case 7: return launch_conv_wino<2, 2, 7>(a, Up, minoff, s);
      case 8: return launch_conv_wino<2, 2, 8>(a, Up, minoff, s);
      case 15: return launch_conv_wino<2, 2, 15>(a, Up, minoff, s);
      default: break;
    }
  }
#endif
  if (force == 1 || (force == 0 && tiles < 128)) return launch_conv_wino<2, 2>(a, (const float*)ws, minoff, s);
  return launch_conv_wino<4, 2>(a, (const float*)ws, minoff, s);
}

}  // namespace gn
