// Transform-domain (Cook-Toom / Winograd F(2,5)) fp32 Conv1D for the unit-stride 5-tap layers: forward and data gradient.
//
//   y[2t + i, n] = sum_p AT[i][p] * ( sum_c V_p[t, c] * U_p[c, n] ),   V_p[t, c] = sum_j BT[p][j] x[2t + off0 + j, c],   U_p = sum_q G[p][q] w_q
//
// six multiplies per two outputs instead of ten: 0.6 of the direct kernel's matrix-core work for the layers that carry most of the step
// (generator 128 -> 256 -> 512 -> 1024, bbhMahoGANy.py:259-283; PE q branch 64 -> 128 -> 256, :382-386).  Points {0, 1, -1, 1/2, -2, inf}:
// the set whose fp32 error stays closest to the direct k-ordered fma chain's (profiles/r05_winograd_gate1.txt: 1.4x rms on the forward map).
// Operands stay fp32 and every product runs on the fp32 matrix instruction (exact fp32 fma chains in the transform domain): dtype f32.
//
// What shaped the kernel (measured, profiles/r05_winograd_gate.txt): fp32 VALU work does NOT hide under the fp32 MFMA -- the instruction runs at
// the fp32 vector rate and a v_fma_f32 beside it costs its full four cycles of the SIMD -- so the input transform, done in registers by the wave
// that consumes it, must be amortised over as many output columns as the accumulator budget allows: v_mfma_f32_16x16x4_f32 (four accumulator
// registers per 16 x 16 tile) lets a wave own 16 tiles (32 output rows) x 64 columns x 6 points in 96 accumulator registers, so one transformed
// fragment feeds four column tiles, and the transform itself runs as packed fp32 (v_pk_fma_f32 on channel pairs): 14 VALU instructions per 48 MFMAs.
//
// Block = WAVES_M waves stacked in M, 64 columns; per 8-channel chunk: input slab (the rows of a stride-2, 6-tap convolution: tile t reads rows
// 2t .. 2t+5; staged as an even-row and an odd-row plane, so a wave's 16 tiles x 4 channel pairs are 512 contiguous bytes per row offset: no
// bank conflicts, one address register) and the chunk's U tile, both by LDS-DMA; the global U image is written by wino_u_kernel in exactly the
// order the lanes read it (one ds_read_b128 = two column tiles x two k-steps), so its staging is a contiguous copy.  THREE LDS stages: the raw
// fragments of chunk c+1 are read and transformed beside the MFMAs of chunk c.  The epilogue applies AT in registers (even rows: sum of points
// 0..4; odd rows: p1 - p2 + p3/2 - 2 p4 + p5) and hands the 16 x 16 tiles to the shared epilogue (bias, activation, dropout, fused backward) with an
// output row stride of 2; BatchNorm statistics leave the epilogue as in conv_pipe.hip.
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include "common.h"
#include "conv_epilogue.h"
#include "wino_common.h"

namespace gn {

struct WinoTaps {
  int wq[5];      // kernel index (into w's leading axis) of the tap at offset off0 + q
};

// U_p = sum_q G[p][q] * w_q, formed in fp64 and rounded once, written as [chunk = ci / 8][column tile = co / 64][p][cth][kq][n16][ctl][s]:
// channel ci = 8 chunk + 2 kq + s, column co = 64 tile + 16 (2 cth + ctl) + n16 -- the order in which the lanes of conv_wino_kernel read a stage.
__global__ void wino_u_kernel(const float* __restrict__ w, float* __restrict__ U, int Cin, int Cout, WinoTaps t) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t cc = (size_t)Cin * Cout;
  if (i >= cc) return;
  const int ci = (int)(i / Cout), co = (int)(i % Cout);
  double g[5];
#pragma unroll
  for (int q = 0; q < 5; ++q) g[q] = (double)w[(size_t)t.wq[q] * cc + i];
  float u[6];
  // (rows 0, 1, 2, 5 carry the factor 2 that wino_piece's rows leave out)
  u[0] = (float)g[0];
  u[1] = (float)((g[0] + g[1] + g[2] + g[3] + g[4]) * (1.0 / 3.0));
  u[2] = (float)((g[0] - g[1] + g[2] - g[3] + g[4]) * (1.0 / 3.0));
  u[3] = (float)((16.0 * g[0] + 8.0 * g[1] + 4.0 * g[2] + 2.0 * g[3] + g[4]) * (1.0 / 15.0));
  u[4] = (float)((g[0] - 2.0 * g[1] + 4.0 * g[2] - 8.0 * g[3] + 16.0 * g[4]) * (1.0 / 30.0));
  u[5] = (float)g[4];
  const int chunk = ci >> 3, c = ci & 7, kq = c >> 1, s = c & 1;
  const int tile = co >> 6, nn = co & 63, ct = nn >> 4, n16 = nn & 15, cth = ct >> 1, ctl = ct & 1;
  const size_t base = ((size_t)chunk * (Cout >> 6) + tile) * 3072 + ((cth * 4 + kq) * 16 + n16) * 4 + ctl * 2 + s;
#pragma unroll
  for (int p = 0; p < 6; ++p) U[base + p * 512] = u[p];
}

// ---------------------------------------------------------------------------------------------
// A chunk of a wave = 48 MFMA slots (point p = slot / 8, k-step s = (slot / 4) % 2, column tile ct = slot % 4), each one asm statement:
//   slot 8p     issues the two reads of the NEXT point's U values (after the last point: point 0 of the next chunk, whose stage has been complete since
//               the previous barrier), slot 0 also this lane's six raw row fragments of the NEXT chunk; a counted lgkmcnt retires exactly what the slot
//               consumes (LDS returns in order);
//   slots 12, 24, 36 carry the 14 transform instructions of the next chunk in three runs; the staging pieces of chunk + 2 sit behind slots 1, 3, ...
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void wino_slot(f32x4& c, float a, float b) {
  asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b) : "memory");
}
template <int O, int WAIT>
__device__ __forceinline__ void wino_slot_rb(f32x4& c, float a, float b, f32x4& nb0, f32x4& nb1, unsigned addr_b) {
  asm volatile(
      "ds_read_b128 %1, %5 offset:%6\n\t"
      "ds_read_b128 %2, %5 offset:%7\n\t"
      "s_waitcnt lgkmcnt(%8)\n\t"
      "v_mfma_f32_16x16x4_f32 %0, %3, %4, %0"
      : "+v"(c), "=&v"(nb0), "=&v"(nb1)
      : "v"(a), "v"(b), "v"(addr_b), "i"(O), "i"(O + 1024), "i"(WAIT)
      : "memory");
}
// ... and the raw fragments of the next chunk (slot 0); this point's U values were read across the barrier: the wait retires them
template <int O, int OP1>
__device__ __forceinline__ void wino_slot_rba(f32x4& c, float a, float b, f32x4& nb0, f32x4& nb1, unsigned addr_b, f32x2 (&d)[6], unsigned addr_a) {
  asm volatile(
      "ds_read_b128 %1, %11 offset:%13\n\t"
      "ds_read_b128 %2, %11 offset:%14\n\t"
      "ds_read_b64 %3, %12\n\t"
      "ds_read_b64 %4, %12 offset:%15\n\t"
      "ds_read_b64 %5, %12 offset:32\n\t"
      "ds_read_b64 %6, %12 offset:%16\n\t"
      "ds_read_b64 %7, %12 offset:64\n\t"
      "ds_read_b64 %8, %12 offset:%17\n\t"
      "s_waitcnt lgkmcnt(8)\n\t"
      "v_mfma_f32_16x16x4_f32 %0, %9, %10, %0"
      : "+v"(c), "=&v"(nb0), "=&v"(nb1), "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5])
      : "v"(a), "v"(b), "v"(addr_b), "v"(addr_a), "i"(O), "i"(O + 1024), "i"(OP1), "i"(OP1 + 32), "i"(OP1 + 64)
      : "memory");
}
// ... slot 8: its wait also retires the raw fragments (issued before the reads this slot adds); they are operands so that their readers depend on it
template <int O>
__device__ __forceinline__ void wino_slot_rbw(f32x4& c, float a, float b, f32x4& nb0, f32x4& nb1, unsigned addr_b, f32x2 (&d)[6]) {
  asm volatile(
      "ds_read_b128 %1, %11 offset:%12\n\t"
      "ds_read_b128 %2, %11 offset:%13\n\t"
      "s_waitcnt lgkmcnt(2)\n\t"
      "v_mfma_f32_16x16x4_f32 %0, %9, %10, %0"
      : "+v"(c), "=&v"(nb0), "=&v"(nb1), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5])
      : "v"(a), "v"(b), "v"(addr_b), "i"(O), "i"(O + 1024)
      : "memory");
}

template <int RPER, int ABL = 0>
struct WinoChunk {
  template <int NPIECES, int I = 0, class D>
  static __device__ __forceinline__ void run(f32x4 (&acc)[6][4], const f32x2 (&v)[6], f32x2 (&vn)[6], f32x2 (&d)[6], WinoT& t, f32x4 (&B)[2][2], unsigned addr_b,
                                             unsigned addr_b_next, unsigned addr_a, unsigned long long k15, unsigned long long km15, D& dma) {
    if constexpr (I < 48) {
      constexpr int P = I / 8, S = (I / 4) % 2, CT = I % 4;
      f32x4(&bc)[2] = B[P & 1];
      f32x4(&bn)[2] = B[(P + 1) & 1];
      const float bv = bc[CT >> 1][2 * (CT & 1) + S];
      const float av = v[P][S];
      if constexpr ((ABL & 8) != 0) wino_slot(acc[P][CT], av, bv);
      else if constexpr (I == 0) wino_slot_rba<2048, RPER * 32>(acc[0][0], av, bv, bn[0], bn[1], addr_b, d, addr_a);
      else if constexpr (I == 8) wino_slot_rbw<2 * 2048>(acc[1][0], av, bv, bn[0], bn[1], addr_b, d);
      else if constexpr (I == 40) wino_slot_rb<0, 2>(acc[5][0], av, bv, bn[0], bn[1], addr_b_next);
      else if constexpr (I % 8 == 0) wino_slot_rb<(P + 1) * 2048, 2>(acc[P][0], av, bv, bn[0], bn[1], addr_b);
      else wino_slot(acc[P][CT], av, bv);
      if constexpr ((I & 1) && (I >> 1) < NPIECES) dma(std::integral_constant<int, (I >> 1)>{});
      if constexpr (!(ABL & 2)) {
        // the 14 transform instructions in THREE runs (behind slots 12, 24, 36), not one every third slot: a vector instruction alone between two MFMAs of a wave
        // costs the switch as well as its own issue (scripts/valu_rate.hip: 14.5-16 cycles alone, 7.4 in a run of three or more; one, two or three runs
        // measure the same).  (This schedule is what exposed the LDS-DMA's write into v[0:3], common.h: GN_LDS_DMA_CLOBBER.)
        if constexpr (I == 12) wino_run<0, 5>(d, vn, t, k15, km15);
        if constexpr (I == 24) wino_run<5, 10>(d, vn, t, k15, km15);
        if constexpr (I == 36) wino_run<10, kWinoPieces>(d, vn, t, k15, km15);
      }
      run<NPIECES, I + 1>(acc, v, vn, d, t, B, addr_b, addr_b_next, addr_a, k15, km15, dma);
    }
  }
};

// ABL: timing ablations, compiled only under -DGN_ABLATION (never into the shipped library; results are wrong for ABL != 0): bit 0 = no staging in the
// loop, bit 1 = no transform, bit 2 = no barrier, bit 3 = no LDS reads in the loop
template <int WAVES_M, int ABL = 0>
__global__ __launch_bounds__(64 * WAVES_M, (WAVES_M > 4 ? 1 : 2)) void conv_wino_kernel(ConvArgs a, const float* __restrict__ U, int off0, int m_tiles, int n_tiles,
                                                                                        int patch) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int KC = 8, NP = 6;
  constexpr int TT = WAVES_M * 16;                 // tiles (output row pairs) per block
  constexpr int TN = 64;
  constexpr int NT = 64 * WAVES_M;
  constexpr int RPER = TT + 2;                     // rows per parity plane: the block reads input rows 0 .. 2 TT + 3 of its window
  constexpr int SLAB = 2 * RPER * KC;              // floats
  constexpr int UT = NP * KC * TN;                 // 3072 floats
  constexpr int BUF = SLAB + UT;
  constexpr int STAGE_BYTES = BUF * 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  typedef __attribute__((address_space(3))) void* lptr_t;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n16 = lane & 15, kq = lane >> 4;
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

  f32x4 acc[NP][4];
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[p][ct][r] = 0.f;

  const int t_base = 2 * t0 + off0;                // input row of slab row 0
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
    const int r = lr < RPER ? 2 * lr : 2 * (lr - RPER) + 1;                // plane 0: even rows, plane 1: odd rows
    soff[it] = (id < S_COUNT) ? ((t_base + r) * a.Cin + 4 * c4) * 4 : 0x40000000;   // rows outside [0, Lin): the descriptor returns 0
  }
  const uintptr_t wbp = (uintptr_t)U;
  const unsigned wb_lo = __builtin_amdgcn_readfirstlane((unsigned)wbp), wb_hi = __builtin_amdgcn_readfirstlane((unsigned)(wbp >> 32));
  const int wbytes = __builtin_amdgcn_readfirstlane(NP * a.Cin * a.Cout * 4);
  const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)wb_hi << 32) | wb_lo), 0, wbytes, 0x00020000);
  const int w_chunk_bytes = n_tiles * UT * 4;      // the U tiles of one channel chunk
  int c_next = 0, st_next = 0;
  bool in_loop = false;
  const int wv64 = __builtin_amdgcn_readfirstlane(tid & ~63);       // the wave's first thread, in a scalar register: the LDS-DMA destination (M0) is then scalar arithmetic
  auto dma_piece = [&](auto kc) {
    constexpr int k = decltype(kc)::value;
    if ((ABL & 1) && in_loop) return;
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
  static_assert(NPIECES <= 6, "staging pieces");
  auto dma_all = [&]() {
    dma_piece(std::integral_constant<int, 0>{});
    if constexpr (NPIECES > 1) dma_piece(std::integral_constant<int, 1>{});
    if constexpr (NPIECES > 2) dma_piece(std::integral_constant<int, 2>{});
    if constexpr (NPIECES > 3) dma_piece(std::integral_constant<int, 3>{});
    if constexpr (NPIECES > 4) dma_piece(std::integral_constant<int, 4>{});
    if constexpr (NPIECES > 5) dma_piece(std::integral_constant<int, 5>{});
  };

  // chunk-invariant byte addresses of this lane's operands in stage 0
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const unsigned base_a = lds0 + (wave * 16 + n16) * 32 + kq * 8;           // row (tile) of plane 0, channel pair kq
  const unsigned base_b = lds0 + SLAB * 4 + (kq * 16 + n16) * 16;           // + (2 p + cth) * 1024
  const unsigned long long k15 = 0x3fc000003fc00000ull, km15 = 0xbfc00000bfc00000ull;       // 1.5, -1.5 on both halves

  const int n_chunks = a.Cin / KC;
  c_next = 0; st_next = 0; dma_all();
  c_next = min(1, n_chunks - 1); st_next = 1; dma_all();
  __syncthreads();                                  // drains the LDS-DMA (vmcnt(0)) in front of the barrier

  f32x2 V0[6], V1[6], d[6];
  f32x4 Bq[2][2];
  WinoT tt;
  {
    const char* sb = reinterpret_cast<const char*>(smem);
#pragma unroll
    for (int j = 0; j < 6; ++j) d[j] = *reinterpret_cast<const f32x2*>(sb + (base_a - lds0) + (j & 1) * RPER * 32 + (j >> 1) * 32);
    wino_bt_all(d, V0, tt, k15, km15);
    Bq[0][0] = *reinterpret_cast<const f32x4*>(sb + (base_b - lds0));
    Bq[0][1] = *reinterpret_cast<const f32x4*>(sb + (base_b - lds0) + 1024);
  }
  int st = 0;
  in_loop = true;
  for (int ch = 0; ch < n_chunks; ch += 2) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (half == 1 && ch + 1 >= n_chunks) break;
      const int st1 = st == 2 ? 0 : st + 1, st2 = st1 == 2 ? 0 : st1 + 1;
      c_next = min(ch + half + 2, n_chunks - 1);                            // stage st2 held chunk - 1: every wave is past its last read of it
      st_next = st2;
      const unsigned addr_a = base_a + st1 * STAGE_BYTES;                   // chunk + 1 landed before the previous barrier
      const unsigned addr_b = base_b + st * STAGE_BYTES, addr_b_next = base_b + st1 * STAGE_BYTES;
      if (half == 0) WinoChunk<RPER, ABL>::template run<NPIECES>(acc, V0, V1, d, tt, Bq, addr_b, addr_b_next, addr_a, k15, km15, dma_piece);
      else WinoChunk<RPER, ABL>::template run<NPIECES>(acc, V1, V0, d, tt, Bq, addr_b, addr_b_next, addr_a, k15, km15, dma_piece);
      if constexpr (!(ABL & 4)) __syncthreads();
      st = st1;
    }
  }
  // MFMA results written inside asm: the compiler inserts no wait states for its own readers of acc
  asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[5][0]), "+v"(acc[5][1]), "+v"(acc[5][2]), "+v"(acc[5][3]));

  // AT: even rows = p0 + p1 + p2 + p3 + p4, odd rows = p1 - p2 + p3 / 2 - 2 p4 + p5
  f32x4 out[2][4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      out[0][ct][r] = (((acc[0][ct][r] + acc[1][ct][r]) + acc[2][ct][r]) + acc[3][ct][r]) + acc[4][ct][r];
      out[1][ct][r] = __builtin_fmaf(-2.f, acc[4][ct][r], __builtin_fmaf(0.5f, acc[3][ct][r], acc[1][ct][r] - acc[2][ct][r])) + acc[5][ct][r];
    }
  const int m_base = t0 + __builtin_amdgcn_readfirstlane(wave) * 16;
  const int mode = a.gy ? (a.gmask ? 3 : 2) : (a.mask ? 1 : 0);
  ConvArgs a2 = a;
  a2.t.out_stride = 2;
  tile16_epilogue_dispatch<4>(a2, out[0], b, m_base, n0, n16, kq, a.t.out_off, mode);
  tile16_epilogue_dispatch<4>(a2, out[1], b, m_base, n0, n16, kq, a.t.out_off + 1, mode);

  if (a.stat_part) {                                 // BatchNorm statistics of the output on the way, as in conv_pipe.hip
    double* red = reinterpret_cast<double*>(smem);   // (every wave is past the loop's last barrier and the DMA is drained: the stages are free)
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
#endif
}

template <int WAVES_M, int ABL = 0>
static int launch_conv_wino(const ConvArgs& a, const float* U, int off0, hipStream_t s) {
  constexpr int TT = WAVES_M * 16, TN = 64;
  constexpr size_t lds = 3 * sizeof(float) * ((size_t)2 * (TT + 2) * 8 + (size_t)6 * 8 * TN);
  static_assert(lds <= 160 * 1024, "stages too large");
  if (lds > 64 * 1024) {
    static unsigned long long lds_done = 0;
    allow_big_lds((const void*)conv_wino_kernel<WAVES_M, ABL>, &lds_done);
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
  hipLaunchKernelGGL((conv_wino_kernel<WAVES_M, ABL>), dim3((unsigned)blocks), dim3(64 * WAVES_M), lds, s, a, U, off0, m_tiles, n_tiles, patch);
  // flop = what the kernel EXECUTES on the matrix pipe: 6 multiplies per output pair and channel pair, 0.6 of the convolution's algorithmic count
  prof_end(s, 0.6 * 2.0 * a.B * (double)a.M * 5 * a.Cin * a.Cout, 5, 4.0 * ((double)a.B * a.Lin * a.Cin + 5.0 * a.Cin * a.Cout + (double)a.B * a.M * a.Cout));
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
  hipLaunchKernelGGL(wino_u_kernel, dim3(cdiv(cc, 256)), dim3(256), 0, s, a.w, (float*)ws, a.Cin, a.Cout, t);
  int rc = check_launch("wino_u");
  if (rc) return rc;
  const float* Up = (const float*)ws;
#ifdef GN_ABLATION
  {
    static const int abl = getenv("GN_WINO_ABL") ? atoi(getenv("GN_WINO_ABL")) : 0;
    switch (abl) {
      case 1: return launch_conv_wino<4, 1>(a, Up, minoff, s);
      case 2: return launch_conv_wino<4, 2>(a, Up, minoff, s);
      case 3: return launch_conv_wino<4, 3>(a, Up, minoff, s);
      case 4: return launch_conv_wino<4, 4>(a, Up, minoff, s);
      case 8: return launch_conv_wino<4, 8>(a, Up, minoff, s);
      case 15: return launch_conv_wino<4, 15>(a, Up, minoff, s);
      default: break;
    }
  }
#endif
  // (blocks of 8 waves -- 128 tiles, one block per CU -- were measured: 197.6 against 215.2 algorithmic TFLOP/s on G 512 -> 1024: three 4-wave blocks per
  // CU hide each other's barriers)
  return launch_conv_wino<4>(a, Up, minoff, s);
}

}  // namespace gn
