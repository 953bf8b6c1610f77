// Transform-domain fp32 convolution for the STRIDE-2 5-tap layers (the discriminator's folded second Conv2D, bbhMahoGANy.py:447; the q branch's Conv1D(512, 5, 2)
// and Conv1D(1024, 5, 2), :385-386; the mc branch :365-371): forward and data gradient.
//
// A stride-2 5-tap convolution is a 3-tap plus a 2-tap UNIT-stride convolution on the even and the odd input rows,
//     y[t] = a[t] w0 + a[t+1] w2 + a[t+2] w4 + b[t] w1 + b[t+1] w3,        a[n] = x[2n + off0], b[n] = x[2n + 1 + off0],
// and its data gradient is two unit-stride phases of 3 and 2 taps over the same dy rows.  Cook-Toom F(2,3) on points {0, 1, -1, inf} and F(2,2) on {0, 1, inf}:
// 4 + 3 = SEVEN multiplies per output pair instead of ten, with input transforms made of additions only (six packed instructions against F(2,5)'s eighteen):
//     v0 = a0 - a2, v1 = a1 + a2, v2 = a2 - a1, v3 = a1 - a3 | v4 = b0 - b1, v5 = b1, v6 = b2 - b1
//     u0 = g0, u1 = (g0 + g1 + g2) / 2, u2 = (g0 - g1 + g2) / 2, u3 = g2 | u4 = h0, u5 = h0 + h1, u6 = h1
//     3-tap: y0 = m0 + m1 + m2, y1 = m1 - m2 - m3 | 2-tap: y0 = m4 + m5, y1 = m5 + m6            (m_p = sum over channels of v_p u_p)
// KIND 1 (forward): tile = output pair (2 tau, 2 tau + 1) reads the SEVEN input rows 4 tau + off0 .. + 6 (a = rows 0, 2, 4, 6; b = rows 1, 3, 5); both sub-convolutions
//   add into the same two output rows.  The slab is staged as FOUR row planes (row mod 4) so that a wave's 16 tiles are 512 contiguous bytes per row offset.
// KIND 2 (data gradient, the merged two-phase description of capi.hip dgrad_impl): tile = two consecutive dy positions, FOUR dy rows 2 tau + minoff .. + 3; the 3-tap
//   phase writes dx rows 4 tau + {0, 2} + its phase, the 2-tap phase (its rows start SB = 0 or 1 rows later) 4 tau + {0, 2} + the other phase: four output rows per tile.
// Everything else is conv_wino.hip's kernel: v_mfma_f32_16x16x4_f32, wave = 16 tiles x 64 columns x 7 points (112 accumulator registers), transforms as packed fp32
// on channel pairs between the MFMA slots, U image written by the transform kernel in the lanes' read order, three LDS stages, shared epilogue.
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include "common.h"
#include "conv_epilogue.h"
#include "wino_common.h"

namespace gn {

struct WinoS2Taps {
  int wq[5];      // kernel indices: [3-tap g0, 2-tap h0, g1, h1, g2], each group by ascending input offset
};

// written as [chunk = ci / 8][column tile = co / 64][p (7)][cth][kq][n16][ctl][s] (conv_wino.hip's order with seven points)
__global__ void wino_s2_u_kernel(const float* __restrict__ w, float* __restrict__ U, int Cin, int Cout, WinoS2Taps t) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t cc = (size_t)Cin * Cout;
  if (i >= cc) return;
  const int ci = (int)(i / Cout), co = (int)(i % Cout);
  double g[5];
#pragma unroll
  for (int q = 0; q < 5; ++q) g[q] = (double)w[(size_t)t.wq[q] * cc + i];
  float u[7];
  u[0] = (float)g[0];
  u[1] = (float)(0.5 * (g[0] + g[2] + g[4]));
  u[2] = (float)(0.5 * (g[0] - g[2] + g[4]));
  u[3] = (float)g[4];
  u[4] = (float)g[1];
  u[5] = (float)(g[1] + g[3]);
  u[6] = (float)g[3];
  const int chunk = ci >> 3, c = ci & 7, kq = c >> 1, s = c & 1;
  const int tile = co >> 6, nn = co & 63, ct = nn >> 4, n16 = nn & 15, cth = ct >> 1, ctl = ct & 1;
  const size_t base = ((size_t)chunk * (Cout >> 6) + tile) * 3584 + ((cth * 4 + kq) * 16 + n16) * 4 + ctl * 2 + s;
#pragma unroll
  for (int p = 0; p < 7; ++p) U[base + p * 512] = u[p];
}

#define GN_PK_ADD(o, x, y) asm volatile("v_pk_add_f32 %0, %1, %2" : "=&v"(o) : "v"(x), "v"(y))
#define GN_PK_COPY(o, x) asm volatile("v_pk_mul_f32 %0, %1, 1.0 op_sel_hi:[1,0]" : "=&v"(o) : "v"(x))

// the seven transformed fragments, one instruction each; NR raw rows d[0 .. NR-1]
template <int KIND, int SB, int K, int NR>
__device__ __forceinline__ void s2_piece(const f32x2 (&d)[NR], f32x2 (&v)[7]) {
  constexpr int A0 = 0, A1 = KIND == 1 ? 2 : 1, A2 = KIND == 1 ? 4 : 2, A3 = KIND == 1 ? 6 : 3;
  constexpr int B0 = KIND == 1 ? 1 : SB, B1 = KIND == 1 ? 3 : SB + 1, B2 = KIND == 1 ? 5 : SB + 2;
  if constexpr (K == 0) GN_PK_SUB(v[0], d[A0], d[A2]);
  else if constexpr (K == 1) GN_PK_ADD(v[1], d[A1], d[A2]);
  else if constexpr (K == 2) GN_PK_SUB(v[2], d[A2], d[A1]);
  else if constexpr (K == 3) GN_PK_SUB(v[3], d[A1], d[A3]);
  else if constexpr (K == 4) GN_PK_SUB(v[4], d[B0], d[B1]);
  else if constexpr (K == 5) GN_PK_COPY(v[5], d[B1]);
  else GN_PK_SUB(v[6], d[B2], d[B1]);
}
template <int KIND, int SB, int NR, int K0, int K1>
__device__ __forceinline__ void s2_run(const f32x2 (&d)[NR], f32x2 (&v)[7]) {
  if constexpr (K0 < K1) {
    s2_piece<KIND, SB, K0, NR>(d, v);
    s2_run<KIND, SB, NR, K0 + 1, K1>(d, v);
  }
}
template <int KIND, int SB, int NR, int K = 0>
__device__ __forceinline__ void s2_bt_all(const f32x2 (&d)[NR], f32x2 (&v)[7]) {
  if constexpr (K < 7) {
    s2_piece<KIND, SB, K, NR>(d, v);
    s2_bt_all<KIND, SB, NR, K + 1>(d, v);
  }
}

__device__ __forceinline__ void s2_slot(f32x4& c, float a, float b) {
  asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b) : "memory");
}
template <int O, int WAIT>
__device__ __forceinline__ void s2_slot_rb(f32x4& c, float a, float b, f32x4& nb0, f32x4& nb1, unsigned addr_b) {
  asm volatile(
      "ds_read_b128 %1, %5 offset:%6\n\t"
      "ds_read_b128 %2, %5 offset:%7\n\t"
      "s_waitcnt lgkmcnt(%8)\n\t"
      "v_mfma_f32_16x16x4_f32 %0, %3, %4, %0"
      : "+v"(c), "=&v"(nb0), "=&v"(nb1)
      : "v"(a), "v"(b), "v"(addr_b), "i"(O), "i"(O + 1024), "i"(WAIT)
      : "memory");
}
// slot 0: the next point's U values and the raw rows of the NEXT chunk; row j sits in plane j % NPL at plane row + j / NPL (PB = bytes per plane)
template <int O, int PB>
__device__ __forceinline__ void s2_slot_rba7(f32x4& c, float a, float b, f32x4& nb0, f32x4& nb1, unsigned addr_b, f32x2 (&d)[7], unsigned addr_a) {
  asm volatile(
      "ds_read_b128 %1, %12 offset:%14\n\t"
      "ds_read_b128 %2, %12 offset:%15\n\t"
      "ds_read_b64 %3, %13\n\t"
      "ds_read_b64 %4, %13 offset:%16\n\t"
      "ds_read_b64 %5, %13 offset:%17\n\t"
      "ds_read_b64 %6, %13 offset:%18\n\t"
      "ds_read_b64 %7, %13 offset:32\n\t"
      "ds_read_b64 %8, %13 offset:%19\n\t"
      "ds_read_b64 %9, %13 offset:%20\n\t"
      "s_waitcnt lgkmcnt(9)\n\t"
      "v_mfma_f32_16x16x4_f32 %0, %10, %11, %0"
      : "+v"(c), "=&v"(nb0), "=&v"(nb1), "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6])
      : "v"(a), "v"(b), "v"(addr_b), "v"(addr_a), "i"(O), "i"(O + 1024), "i"(PB), "i"(2 * PB), "i"(3 * PB), "i"(PB + 32), "i"(2 * PB + 32)
      : "memory");
}
template <int O, int PB>
__device__ __forceinline__ void s2_slot_rba4(f32x4& c, float a, float b, f32x4& nb0, f32x4& nb1, unsigned addr_b, f32x2 (&d)[4], unsigned addr_a) {
  asm volatile(
      "ds_read_b128 %1, %9 offset:%11\n\t"
      "ds_read_b128 %2, %9 offset:%12\n\t"
      "ds_read_b64 %3, %10\n\t"
      "ds_read_b64 %4, %10 offset:%13\n\t"
      "ds_read_b64 %5, %10 offset:32\n\t"
      "ds_read_b64 %6, %10 offset:%14\n\t"
      "s_waitcnt lgkmcnt(6)\n\t"
      "v_mfma_f32_16x16x4_f32 %0, %7, %8, %0"
      : "+v"(c), "=&v"(nb0), "=&v"(nb1), "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])
      : "v"(a), "v"(b), "v"(addr_b), "v"(addr_a), "i"(O), "i"(O + 1024), "i"(PB), "i"(PB + 32)
      : "memory");
}
// slot 8: its wait also retires the raw rows (issued before the reads this slot adds); they are operands so that their readers depend on it
template <int O>
__device__ __forceinline__ void s2_slot_rbw7(f32x4& c, float a, float b, f32x4& nb0, f32x4& nb1, unsigned addr_b, f32x2 (&d)[7]) {
  asm volatile(
      "ds_read_b128 %1, %12 offset:%13\n\t"
      "ds_read_b128 %2, %12 offset:%14\n\t"
      "s_waitcnt lgkmcnt(2)\n\t"
      "v_mfma_f32_16x16x4_f32 %0, %10, %11, %0"
      : "+v"(c), "=&v"(nb0), "=&v"(nb1), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6])
      : "v"(a), "v"(b), "v"(addr_b), "i"(O), "i"(O + 1024)
      : "memory");
}
template <int O>
__device__ __forceinline__ void s2_slot_rbw4(f32x4& c, float a, float b, f32x4& nb0, f32x4& nb1, unsigned addr_b, f32x2 (&d)[4]) {
  asm volatile(
      "ds_read_b128 %1, %9 offset:%10\n\t"
      "ds_read_b128 %2, %9 offset:%11\n\t"
      "s_waitcnt lgkmcnt(2)\n\t"
      "v_mfma_f32_16x16x4_f32 %0, %7, %8, %0"
      : "+v"(c), "=&v"(nb0), "=&v"(nb1), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3])
      : "v"(a), "v"(b), "v"(addr_b), "i"(O), "i"(O + 1024)
      : "memory");
}

// 56 MFMA slots per chunk (point p = slot / 8, k-step s = (slot / 4) % 2, column tile ct = slot % 4); conv_wino.hip's choreography with seven points
// PAR: with SEVEN points the double-buffered U values change parity from chunk to chunk (the next chunk's point 0 is prefetched behind point 6 into the buffer
// point 7 would use): point p of a chunk of parity PAR lives in B[(p + PAR) & 1]; the chunk loop alternates PAR with its two unrolled halves.
template <int KIND, int SB, int RPER, int NR, int PAR>
struct S2Chunk {
  template <int NPIECES, int I = 0, class D>
  static __device__ __forceinline__ void run(f32x4 (&acc)[7][4], const f32x2 (&v)[7], f32x2 (&vn)[7], f32x2 (&d)[NR], f32x4 (&B)[2][2], unsigned addr_b,
                                             unsigned addr_b_next, unsigned addr_a, D& dma) {
    if constexpr (I < 56) {
      constexpr int P = I / 8, S = (I / 4) % 2, CT = I % 4;
      f32x4(&bc)[2] = B[(P + PAR) & 1];
      f32x4(&bn)[2] = B[(P + PAR + 1) & 1];
      const float bv = bc[CT >> 1][2 * (CT & 1) + S];
      const float av = v[P][S];
      if constexpr (I == 0) {
        if constexpr (NR == 7) s2_slot_rba7<2048, RPER * 32>(acc[0][0], av, bv, bn[0], bn[1], addr_b, d, addr_a);
        else s2_slot_rba4<2048, RPER * 32>(acc[0][0], av, bv, bn[0], bn[1], addr_b, d, addr_a);
      } else if constexpr (I == 8) {
        if constexpr (NR == 7) s2_slot_rbw7<2 * 2048>(acc[1][0], av, bv, bn[0], bn[1], addr_b, d);
        else s2_slot_rbw4<2 * 2048>(acc[1][0], av, bv, bn[0], bn[1], addr_b, d);
      } else if constexpr (I == 48) s2_slot_rb<0, 2>(acc[6][0], av, bv, bn[0], bn[1], addr_b_next);
      else if constexpr (I % 8 == 0) s2_slot_rb<(P + 1) * 2048, 2>(acc[P][0], av, bv, bn[0], bn[1], addr_b);
      else s2_slot(acc[P][CT], av, bv);
      if constexpr ((I & 1) && (I >> 1) < NPIECES) dma(std::integral_constant<int, (I >> 1)>{});
      // the seven transform instructions in two runs (a vector instruction alone between two MFMAs of a wave costs 16 cycles, in a run 7: scripts/valu_rate.hip)
      if constexpr (I == 20) s2_run<KIND, SB, NR, 0, 4>(d, vn);
      if constexpr (I == 36) s2_run<KIND, SB, NR, 4, 7>(d, vn);
      run<NPIECES, I + 1>(acc, v, vn, d, B, addr_b, addr_b_next, addr_a, dma);
    }
  }
};

template <int KIND, int SB>
__global__ __launch_bounds__(256, 2) void conv_wino_s2_kernel(ConvArgs a, const float* __restrict__ U, int off0, int m_tiles, int n_tiles, int patch, int offA,
                                                              int offB) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int KC = 8, NP = 7, WAVES_M = 4;
  constexpr int TT = WAVES_M * 16, TN = 64, NT = 64 * WAVES_M;
  constexpr int NR = KIND == 1 ? 7 : 4;            // raw rows per tile
  constexpr int TS = KIND == 1 ? 4 : 2;            // input rows between consecutive tiles = row planes
  constexpr int RPER = TT + 1;                     // rows per plane: the block reads input rows 0 .. TS (TT - 1) + NR - 1 of its window
  constexpr int SLAB = TS * RPER * KC;             // floats
  constexpr int UT = NP * KC * TN;                 // 3584 floats
  constexpr int BUF = SLAB + UT;
  constexpr int STAGE_BYTES = BUF * 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  typedef __attribute__((address_space(3))) void* lptr_t;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n16 = lane & 15, kq = lane >> 4;
  const int bid = blockIdx.x;
  int n_lin, slab;
  if (patch >= 0 && bid < (int)(gridDim.x & ~511u)) {          // the XCD patch order of conv_pipe.hip
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

  f32x4 acc[NP][4];
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[p][ct][r] = 0.f;

  const int t_base = TS * t0 + off0;               // input row of slab row 0
  const uintptr_t xbp = (uintptr_t)(a.x + (size_t)b * a.Lin * a.Cin);
  const unsigned xb_lo = __builtin_amdgcn_readfirstlane((unsigned)xbp), xb_hi = __builtin_amdgcn_readfirstlane((unsigned)(xbp >> 32));
  const int xbytes = __builtin_amdgcn_readfirstlane(a.Lin * a.Cin * 4);
  const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)xb_hi << 32) | xb_lo), 0, xbytes, 0x00020000);

  constexpr int S_COUNT = SLAB / 4;                // 16-byte granules of one slab stage, in LDS order
  constexpr int S_ITEMS = (S_COUNT + NT - 1) / NT;
  constexpr int W_TOTAL = UT / 4;
  constexpr int W_ITEMS = (W_TOTAL + NT - 1) / NT;
  int soff[S_ITEMS];
#pragma unroll
  for (int it = 0; it < S_ITEMS; ++it) {
    const int id = tid + it * NT;
    const int lr = id >> 1, c4 = id & 1;
    const int r = TS * (lr % RPER) + lr / RPER;                              // plane = row mod TS
    soff[it] = (id < S_COUNT) ? ((t_base + r) * a.Cin + 4 * c4) * 4 : 0x40000000;   // rows outside [0, Lin): the descriptor returns 0
  }
  const uintptr_t wbp = (uintptr_t)U;
  const unsigned wb_lo = __builtin_amdgcn_readfirstlane((unsigned)wbp), wb_hi = __builtin_amdgcn_readfirstlane((unsigned)(wbp >> 32));
  const int wbytes = __builtin_amdgcn_readfirstlane(NP * a.Cin * a.Cout * 4);
  const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)wb_hi << 32) | wb_lo), 0, wbytes, 0x00020000);
  const int w_chunk_bytes = n_tiles * UT * 4;
  int c_next = 0, st_next = 0;
  const int wv64 = __builtin_amdgcn_readfirstlane(tid & ~63);       // the wave's first thread, in a scalar register: the LDS-DMA destination (M0) is then scalar arithmetic
  auto dma_piece = [&](auto kc) {
    constexpr int k = decltype(kc)::value;
    float* stg = smem + st_next * BUF;
    if constexpr (k < S_ITEMS) {
      if ((k + 1) * NT <= S_COUNT || tid + k * NT < S_COUNT)
        gn_buffer_load_lds(xsrd, (lptr_t)(stg + (k * NT + wv64) * 4), 16, soff[k], c_next * KC * 4, 0, 0);
    } else {
      constexpr int it = k - S_ITEMS;
      if ((it + 1) * NT <= W_TOTAL || wv64 + it * NT < W_TOTAL)
        gn_buffer_load_lds(wsrd, (lptr_t)(stg + SLAB + (it * NT + wv64) * 4), 16, (n_tile * UT + (tid + it * NT) * 4) * 4,
                                                 c_next * w_chunk_bytes, 0, 0);
    }
  };
  constexpr int NPIECES = S_ITEMS + W_ITEMS;
  static_assert(NPIECES <= 8, "staging pieces");
  auto dma_all = [&]() {
    dma_piece(std::integral_constant<int, 0>{});
    dma_piece(std::integral_constant<int, 1>{});
    if constexpr (NPIECES > 2) dma_piece(std::integral_constant<int, 2>{});
    if constexpr (NPIECES > 3) dma_piece(std::integral_constant<int, 3>{});
    if constexpr (NPIECES > 4) dma_piece(std::integral_constant<int, 4>{});
    if constexpr (NPIECES > 5) dma_piece(std::integral_constant<int, 5>{});
    if constexpr (NPIECES > 6) dma_piece(std::integral_constant<int, 6>{});
    if constexpr (NPIECES > 7) dma_piece(std::integral_constant<int, 7>{});
  };

  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const unsigned base_a = lds0 + (wave * 16 + n16) * 32 + kq * 8;           // plane 0, the tile's row, channel pair kq
  const unsigned base_b = lds0 + SLAB * 4 + (kq * 16 + n16) * 16;           // + (2 p + cth) * 1024

  const int n_chunks = a.Cin / KC;
  c_next = 0; st_next = 0; dma_all();
  c_next = min(1, n_chunks - 1); st_next = 1; dma_all();
  __syncthreads();

  f32x2 V0[7], V1[7], d[NR];
  f32x4 Bq[2][2];
  {
    const char* sb = reinterpret_cast<const char*>(smem);
#pragma unroll
    for (int j = 0; j < NR; ++j) d[j] = *reinterpret_cast<const f32x2*>(sb + (base_a - lds0) + (j % TS) * RPER * 32 + (j / TS) * 32);
    s2_bt_all<KIND, SB, NR>(d, V0);
    Bq[0][0] = *reinterpret_cast<const f32x4*>(sb + (base_b - lds0));
    Bq[0][1] = *reinterpret_cast<const f32x4*>(sb + (base_b - lds0) + 1024);
  }
  int st = 0;
  for (int ch = 0; ch < n_chunks; ch += 2) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (half == 1 && ch + 1 >= n_chunks) break;
      const int st1 = st == 2 ? 0 : st + 1, st2 = st1 == 2 ? 0 : st1 + 1;
      c_next = min(ch + half + 2, n_chunks - 1);
      st_next = st2;
      const unsigned addr_a = base_a + st1 * STAGE_BYTES;
      const unsigned addr_b = base_b + st * STAGE_BYTES, addr_b_next = base_b + st1 * STAGE_BYTES;
      if (half == 0) S2Chunk<KIND, SB, RPER, NR, 0>::template run<NPIECES>(acc, V0, V1, d, Bq, addr_b, addr_b_next, addr_a, dma_piece);
      else S2Chunk<KIND, SB, RPER, NR, 1>::template run<NPIECES>(acc, V1, V0, d, Bq, addr_b, addr_b_next, addr_a, dma_piece);
      __syncthreads();
      st = st1;
    }
  }
  asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[6][0]), "+v"(acc[6][1]), "+v"(acc[6][2]), "+v"(acc[6][3]));

  const int m_base = t0 + __builtin_amdgcn_readfirstlane(wave) * 16;
  const int mode = a.gy ? (a.gmask ? 3 : 2) : (a.mask ? 1 : 0);
  ConvArgs a2 = a;
  if constexpr (KIND == 1) {
    // both sub-convolutions add into output rows 2 tau and 2 tau + 1
    f32x4 out[2][4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        out[0][ct][r] = ((acc[0][ct][r] + acc[1][ct][r]) + acc[2][ct][r]) + (acc[4][ct][r] + acc[5][ct][r]);
        out[1][ct][r] = ((acc[1][ct][r] - acc[2][ct][r]) - acc[3][ct][r]) + (acc[5][ct][r] + acc[6][ct][r]);
      }
    a2.t.out_stride = 2;
    tile16_epilogue_dispatch<4>(a2, out[0], b, m_base, n0, n16, kq, a.t.out_off, mode);
    tile16_epilogue_dispatch<4>(a2, out[1], b, m_base, n0, n16, kq, a.t.out_off + 1, mode);
    if (a.stat_part) {                               // BatchNorm statistics of the output, as in conv_wino.hip
      double* red = reinterpret_cast<double*>(smem);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const float bias = a.bias ? a.bias[n0 + ct * 16 + n16] : 0.f;
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 2 * (m_base + 4 * kq + r) + ph;
            if (row < a.M) {
              const double v = (double)(out[ph][ct][r] + bias);
              s1 += v; s2 += v * v;
            }
          }
        s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        if (kq == 0) {
          const int slot = (wave * TN + ct * 16 + n16) * 2;
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
  } else {
    // two phases x two positions: output rows 4 tau + {0, 2} + offA (3-tap phase) and + offB (2-tap phase)
    f32x4 out[4][4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        out[0][ct][r] = (acc[0][ct][r] + acc[1][ct][r]) + acc[2][ct][r];
        out[1][ct][r] = (acc[1][ct][r] - acc[2][ct][r]) - acc[3][ct][r];
        out[2][ct][r] = acc[4][ct][r] + acc[5][ct][r];
        out[3][ct][r] = acc[5][ct][r] + acc[6][ct][r];
      }
    a2.t.out_stride = 4;
    tile16_epilogue_dispatch<4>(a2, out[0], b, m_base, n0, n16, kq, offA, mode);
    tile16_epilogue_dispatch<4>(a2, out[1], b, m_base, n0, n16, kq, offA + 2, mode);
    tile16_epilogue_dispatch<4>(a2, out[2], b, m_base, n0, n16, kq, offB, mode);
    tile16_epilogue_dispatch<4>(a2, out[3], b, m_base, n0, n16, kq, offB + 2, mode);
  }
#endif
}

template <int KIND, int SB>
static int launch_s2(const ConvArgs& a, const float* U, int off0, int offA, int offB, hipStream_t s) {
  constexpr int TT = 64, TN = 64, TS = KIND == 1 ? 4 : 2;
  constexpr size_t lds = 3 * sizeof(float) * ((size_t)TS * (TT + 1) * 8 + (size_t)7 * 8 * TN);
  static_assert(lds <= 160 * 1024, "stages too large");
  if (lds > 64 * 1024) {
    static unsigned long long lds_done = 0;
    allow_big_lds((const void*)conv_wino_s2_kernel<KIND, SB>, &lds_done);
  }
  const int tiles = (a.M + 1) / 2;
  const int m_tiles = (tiles + TT - 1) / TT, n_tiles = a.Cout / TN;
  const size_t blocks = (size_t)m_tiles * n_tiles * a.B;
  if (blocks == 0 || blocks > 0x7fffffffull) {
    set_error("conv_wino_s2: bad grid %zu", blocks);
    return GN_EINVAL;
  }
  int patch = -1;
  {
    const int pn = n_tiles % 8 == 0 ? 3 : (n_tiles == 4 ? 2 : -1);
    const int ng = pn >= 0 ? n_tiles >> pn : 0;
    if (pn >= 0 && ng <= 8 && (ng & (ng - 1)) == 0) patch = __builtin_ctz(ng) | (pn << 8);
  }
  prof_begin(s);
  hipLaunchKernelGGL((conv_wino_s2_kernel<KIND, SB>), dim3((unsigned)blocks), dim3(256), lds, s, a, U, off0, m_tiles, n_tiles, patch, offA, offB);
  // flop = what the kernel EXECUTES: 7 multiplies per output pair and channel, 0.7 of the convolution's algorithmic count (kind 7)
  prof_end(s, 0.7 * 2.0 * a.B * (double)a.M * 5 * a.Cin * a.Cout, 7, 4.0 * ((double)a.B * a.Lin * a.Cin + 5.0 * a.Cin * a.Cout + (double)a.B * a.M * a.Cout));
  int rc = check_launch("conv_wino_s2");
  if (rc || !a.stat_part) return rc;
  *a.stat_done = 1;
  return colred_finalize(a.stat_part, a.stat_sums, (size_t)2 * a.Cout, a.B * m_tiles, s);
}

size_t conv_wino_s2_workspace_bytes(int Cin, int Cout) { return (size_t)7 * Cin * Cout * sizeof(float); }

// 1: the forward of a stride-2 5-tap layer; 2: its data gradient in the merged two-phase description (capi.hip dgrad_impl); 0: neither
int conv_wino_s2_kind(const ConvArgs& a) {
  if (a.t.ntaps != 5 || a.Cin % 8 || a.Cout % 64 || a.Cin < 32) return 0;
  if ((size_t)a.Ly * a.Cout * 4 >= 0x40000000ull || (size_t)a.Lin * a.Cin * 4 >= 0x40000000ull || (size_t)7 * a.Cin * a.Cout * 4 >= 0x40000000ull) return 0;
  if (a.t.in_stride == 2 && a.t.out_stride == 1 && a.t.out_off == 0) {
    for (int j = 0; j < 5; ++j)
      if (a.t.off[j] != a.t.off[0] + j) return 0;
    if (a.stat_part && (a.act != GN_ACT_LINEAR || a.mask || a.gy)) return 0;
    return 1;
  }
  if (a.t.in_stride == 1 && a.t.out_stride == 2 && !a.stat_part && !a.mask && !a.bias) {
    // taps of even index: the 3-tap phase (rows out_off), consecutive descending offsets; taps of odd index: the 2-tap phase (out_off_odd)
    if (a.t.off[0] != a.t.off[2] + 1 || a.t.off[2] != a.t.off[4] + 1 || a.t.off[1] != a.t.off[3] + 1) return 0;
    const int sb = a.t.off[3] - a.t.off[4];
    if (sb != 0 && sb != 1) return 0;
    return 2;
  }
  return 0;
}

int conv_wino_s2_run(const ConvArgs& a, void* ws, size_t ws_bytes, hipStream_t s) {
  const int kind = conv_wino_s2_kind(a);
  if (!kind) {
    set_error("conv_wino_s2: unsupported shape");
    return GN_EINVAL;
  }
  if (!ws || ws_bytes < conv_wino_s2_workspace_bytes(a.Cin, a.Cout)) {
    set_error("conv_wino_s2: workspace too small (%zu < %zu)", ws_bytes, conv_wino_s2_workspace_bytes(a.Cin, a.Cout));
    return GN_EWORKSPACE;
  }
  WinoS2Taps t;
  int off0, offA = 0, offB = 0, sb = 0;
  if (kind == 1) {
    for (int j = 0; j < 5; ++j) t.wq[j] = a.t.widx[j];                 // [g0, h0, g1, h1, g2] = taps 0 .. 4 by ascending offset
    off0 = a.t.off[0];
  } else {
    t.wq[0] = a.t.widx[4]; t.wq[2] = a.t.widx[2]; t.wq[4] = a.t.widx[0];      // 3-tap phase by ASCENDING offset
    t.wq[1] = a.t.widx[3]; t.wq[3] = a.t.widx[1];                              // 2-tap phase
    off0 = a.t.off[4];
    sb = a.t.off[3] - a.t.off[4];
    offA = a.t.out_off; offB = a.t.out_off_odd;
  }
  const size_t cc = (size_t)a.Cin * a.Cout;
  hipLaunchKernelGGL(wino_s2_u_kernel, dim3(cdiv(cc, 256)), dim3(256), 0, s, a.w, (float*)ws, a.Cin, a.Cout, t);
  int rc = check_launch("wino_s2_u");
  if (rc) return rc;
  const float* Up = (const float*)ws;
  if (kind == 1) return launch_s2<1, 0>(a, Up, off0, 0, 0, s);
  return sb ? launch_s2<2, 1>(a, Up, off0, offA, offB, s) : launch_s2<2, 0>(a, Up, off0, offA, offB, s);
}

}  // namespace gn
