// Transform-domain (Cook-Toom F(2,5), transposed form) fp32 weight gradient of the unit-stride 5-tap layers:
//
//   dW_q[ci, co] = sum_p G[p][q] Q_p[ci, co],   Q_p[ci, co] = sum_{b, t} V_p[b, t, ci] * D_p[b, t, co]
//   V_p[t] = sum_j BT[p][j] x[2t + off0 + j]   (the forward kernel's input transform),   D_p[t] = sum_i A[p][i] dy[2t + i]   (A = AT transposed)
//
// six multiplies per output-row pair and (ci, co) instead of ten (generator Conv1D(256 / 512 / 1024, 5), bbhMahoGANy.py:266-283; PE q branch :382-384).
// Same ingredients as conv_wino.hip: v_mfma_f32_16x16x4_f32 with a wave tile of 16 ci x 64 co x 6 points (96 accumulator registers), so the expensive
// transform (x: 14 packed instructions) is amortised over 64 columns; both operand transforms run in registers as packed fp32 on PAIRS of tiles -- a
// lane's A / B operand of two consecutive k-steps (tiles kq and kq + 4 of an 8-tile pair-step) come out of ONE ds_read2st64_b32 as a register pair,
// because the LDS image keeps 16 channels per row (64 bytes: the four tile groups of a wave-wide read fall on distinct bank quarters) and tiles four
// apart sit 256 bytes apart.  The dy transform (4 packed instructions per column tile) is done just in time, one column tile ahead of its 12 MFMAs.
// Block = 4 waves (64 ci x 64 co), K-chunk = 16 tiles (32 rows) of one batch element, three LDS stages, one barrier per 96 MFMAs; K-splits and
// partial slabs as the direct kernel's (wgrad_split_plan); the reduce pass sums the splits in fp64 and applies G^T.
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include "common.h"
#include "wino_common.h"

namespace gn {

__device__ __forceinline__ void wg_slot(f32x4& c, float a, float b) {
  asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b) : "memory");
}
// group slot 0: the next column tile's dy row pair (two tile pairs each)
template <int O>
__device__ __forceinline__ void wg_slot_e(f32x4& c, float a, float b, f32x2& e0, f32x2& e1, unsigned addr_b) {
  asm volatile(
      "ds_read2st64_b32 %1, %5 offset0:%6 offset1:%7\n\t"
      "ds_read2st64_b32 %2, %5 offset0:%8 offset1:%9\n\t"
      "v_mfma_f32_16x16x4_f32 %0, %3, %4, %0"
      : "+v"(c), "=&v"(e0), "=&v"(e1)
      : "v"(a), "v"(b), "v"(addr_b), "i"(O), "i"(O + 1), "i"(O + 4), "i"(O + 5)
      : "memory");
}
// ... pair-step slot 0: also the six raw x row pairs of the NEXT pair-step (OA = 2 * its half inside the stage)
template <int O, int OA>
__device__ __forceinline__ void wg_slot_ea(f32x4& c, float a, float b, f32x2& e0, f32x2& e1, unsigned addr_b, f32x2 (&d)[6], const unsigned (&addr_a)[3]) {
  asm volatile(
      "ds_read2st64_b32 %1, %11 offset0:%15 offset1:%16\n\t"
      "ds_read2st64_b32 %2, %11 offset0:%17 offset1:%18\n\t"
      "ds_read2st64_b32 %3, %12 offset0:%19 offset1:%20\n\t"
      "ds_read2st64_b32 %4, %12 offset0:%21 offset1:%22\n\t"
      "ds_read2st64_b32 %5, %13 offset0:%19 offset1:%20\n\t"
      "ds_read2st64_b32 %6, %13 offset0:%21 offset1:%22\n\t"
      "ds_read2st64_b32 %7, %14 offset0:%19 offset1:%20\n\t"
      "ds_read2st64_b32 %8, %14 offset0:%21 offset1:%22\n\t"
      "v_mfma_f32_16x16x4_f32 %0, %9, %10, %0"
      : "+v"(c), "=&v"(e0), "=&v"(e1), "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5])
      : "v"(a), "v"(b), "v"(addr_b), "v"(addr_a[0]), "v"(addr_a[1]), "v"(addr_a[2]), "i"(O), "i"(O + 1), "i"(O + 4), "i"(O + 5), "i"(OA), "i"(OA + 1), "i"(OA + 5),
        "i"(OA + 6)
      : "memory");
}
// ... slot 12: its reads go behind the raw x fragments, so a wait for all but its own two retires them; they are operands so that their readers depend on it
template <int O>
__device__ __forceinline__ void wg_slot_e_wd(f32x4& c, float a, float b, f32x2& e0, f32x2& e1, unsigned addr_b, f32x2 (&d)[6]) {
  asm volatile(
      "ds_read2st64_b32 %1, %11 offset0:%12 offset1:%13\n\t"
      "ds_read2st64_b32 %2, %11 offset0:%14 offset1:%15\n\t"
      "s_waitcnt lgkmcnt(2)\n\t"
      "v_mfma_f32_16x16x4_f32 %0, %9, %10, %0"
      : "+v"(c), "=&v"(e0), "=&v"(e1), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5])
      : "v"(a), "v"(b), "v"(addr_b), "i"(O), "i"(O + 1), "i"(O + 4), "i"(O + 5)
      : "memory");
}
// group slot 4: the dy pair of the next column tile has to be there before its transform (WAIT = LDS operations issued after it that may stay in flight)
template <int WAIT>
__device__ __forceinline__ void wg_slot_w(f32x4& c, float a, float b, f32x2& e0, f32x2& e1) {
  asm volatile("s_waitcnt lgkmcnt(%5)\n\tv_mfma_f32_16x16x4_f32 %0, %3, %4, %0" : "+v"(c), "+v"(e0), "+v"(e1) : "v"(a), "v"(b), "i"(WAIT) : "memory");
}

// One pair-step of a wave: 8 tiles (two k-steps), 48 MFMA slots ordered column tile (12 each) > k-step > point.  H = which half of the 16-tile chunk.
// Beside the MFMAs: the dy pair of the NEXT column tile (read at group slot 0, transformed behind slots 5-8), the x pairs of the NEXT pair-step (read
// at slot 0, transformed behind 18 slots of the column tiles 1-3), the staging pieces of chunk + 2 behind the wait slots.
template <int H, int ABL = 0>
struct WgPair {
  template <int NP0, int NP1, int I = 0, class DMA>
  static __device__ __forceinline__ void run(f32x4 (&acc)[6][4], const f32x2 (&v)[6], f32x2 (&vn)[6], f32x2 (&d)[6], WinoT& t, f32x2 (&D)[2][6], unsigned addr_b,
                                             unsigned addr_b_next, const unsigned (&addr_a)[3], const unsigned (&addr_a_next)[3], unsigned long long k15,
                                             unsigned long long km15, DMA& dma) {
    if constexpr (I < 48) {
      constexpr int CT = I / 12, G = I % 12, S = G / 6, P = G % 6;
      f32x2(&dc)[6] = D[CT & 1];
      f32x2(&dn)[6] = D[(CT + 1) & 1];
      const float av = v[P][S], bv = dc[P][S];
      // where the next column tile's dy pair lives: this stage, next column tile; behind the last column tile the next pair-step's first one
      constexpr int OE = (CT < 3) ? (CT + 1) * 8 + 2 * H : (H == 0 ? 2 : 0);
      if constexpr ((ABL & 8) != 0) wg_slot(acc[P][CT], av, bv);
      else if constexpr (I == 0) {
        if constexpr (H == 0) wg_slot_ea<OE, 2>(acc[P][CT], av, bv, dn[0], dn[5], addr_b, d, addr_a);                  // x pairs of this chunk's second half
        else wg_slot_ea<OE, 0>(acc[P][CT], av, bv, dn[0], dn[5], addr_b, d, addr_a_next);                               // ... of the next chunk's first half
      } else if constexpr (I == 12) wg_slot_e_wd<OE>(acc[P][CT], av, bv, dn[0], dn[5], addr_b, d);
      else if constexpr (G == 0) {
        if constexpr (CT == 3 && H == 1) wg_slot_e<OE>(acc[P][CT], av, bv, dn[0], dn[5], addr_b_next);
        else wg_slot_e<OE>(acc[P][CT], av, bv, dn[0], dn[5], addr_b);
      } else if constexpr (G == 4) wg_slot_w<(I == 4 ? 6 : 0)>(acc[P][CT], av, bv, dn[0], dn[5]);
      else wg_slot(acc[P][CT], av, bv);
      if constexpr (G == 4) {
        constexpr int K = (H == 0 ? 0 : NP0) + CT;
        if constexpr (CT < (H == 0 ? NP0 : NP1)) dma(std::integral_constant<int, K>{});
      }
      if constexpr (!(ABL & 2)) {
        // vector instructions in RUNS (alone between two MFMAs of a wave one costs 16 cycles, in a run 7: scripts/valu_rate.hip): the dy transform's four
        // behind slot 5 of every column tile, the next x transform as 5 + 5 + 4 behind slot 9 of column tiles 1, 2, 3
        if constexpr (G == 5) {
          wino_a_piece<0>(dn[0], dn[5], dn[1], dn[2], dn[3], dn[4]); wino_a_piece<1>(dn[0], dn[5], dn[1], dn[2], dn[3], dn[4]);
          wino_a_piece<2>(dn[0], dn[5], dn[1], dn[2], dn[3], dn[4]); wino_a_piece<3>(dn[0], dn[5], dn[1], dn[2], dn[3], dn[4]);
        }
        if constexpr (CT == 1 && G == 9) wino_run<0, 5>(d, vn, t, k15, km15);
        if constexpr (CT == 2 && G == 9) wino_run<5, 10>(d, vn, t, k15, km15);
        if constexpr (CT == 3 && G == 9) wino_run<10, kWinoPieces>(d, vn, t, k15, km15);
      }
      run<NP0, NP1, I + 1>(acc, v, vn, d, t, D, addr_b, addr_b_next, addr_a, addr_a_next, k15, km15, dma);
    }
  }
};

// ABL: timing ablations, compiled only under -DGN_ABLATION (results are wrong for ABL != 0): bit 0 = no staging in the loop, bit 1 = no transforms,
// bit 2 = no barrier, bit 3 = no LDS reads in the loop
template <int ABL = 0>
__global__ __launch_bounds__(256, 2) void wgrad_wino_kernel(WgradArgs a, int off0, int cpb) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NT = 256;
  constexpr int XS = 4 * 2 * 20 * 16;              // floats: 4 sub-slabs (16 ci each) x 2 row-parity planes x 20 rows (18 used) x 16 channels
  constexpr int YS = 4 * 2 * 16 * 16;              // 4 column tiles x 2 row planes x 16 tiles x 16 columns
  constexpr int BUF = XS + YS;
  constexpr int STAGE_BYTES = BUF * 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  typedef __attribute__((address_space(3))) void* lptr_t;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n16 = lane & 15, kq = lane >> 4;
  const int ci0 = blockIdx.x * 64, co0 = blockIdx.y * 64, split = blockIdx.z;
  const int total = a.B * cpb;
  const int q_begin = split * a.chunks_per_split, q_end = min(q_begin + a.chunks_per_split, total);

  f32x4 acc[6][4];
#pragma unroll
  for (int p = 0; p < 6; ++p)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[p][ct][r] = 0.f;

  // staging: x granule g of a stage = [sub-slab g / 160][plane][row][4 channels], dy granule g = [column tile g / 128][plane][tile][4 columns]; the lane's
  // byte offset inside the batch element for chunk 0 (rows before the element: negative -> out of the descriptor's range -> 0; so are rows past its end)
  constexpr int X_ITEMS = 3, Y_ITEMS = 2;
  int xoff[X_ITEMS], yoff[Y_ITEMS];
#pragma unroll
  for (int it = 0; it < X_ITEMS; ++it) {
    const int g = tid + it * NT;
    const int sub = g / 160, rem = g % 160, plane = rem / 80, prow = (rem % 80) >> 2, c4 = rem & 3;
    xoff[it] = ((2 * prow + plane + off0) * a.Cin + ci0 + sub * 16 + 4 * c4) * 4;
  }
#pragma unroll
  for (int it = 0; it < Y_ITEMS; ++it) {
    const int g = tid + it * NT;
    const int ct = g >> 7, rem = g & 127, plane = rem >> 6, tile = (rem & 63) >> 2, c4 = rem & 3;
    yoff[it] = ((2 * tile + plane) * a.Cout + co0 + ct * 16 + 4 * c4) * 4;
  }
  const int xbytes = __builtin_amdgcn_readfirstlane(a.Lin * a.Cin * 4), ybytes = __builtin_amdgcn_readfirstlane(a.M * a.Cout * 4);
  int q_next = 0, st_next = 0;
  bool in_loop = false;
  const int wv64 = __builtin_amdgcn_readfirstlane(tid & ~63);       // the wave's first thread, in a scalar register: the LDS-DMA destination (M0) is then scalar arithmetic
  auto dma_piece = [&](auto kc) {
    constexpr int k = decltype(kc)::value;
    if ((ABL & 1) && in_loop) return;
    const int b = __builtin_amdgcn_readfirstlane(q_next / cpb), cb = __builtin_amdgcn_readfirstlane(q_next % cpb);
    float* stg = smem + st_next * BUF;
    if constexpr (k < X_ITEMS) {
      if (k < 2 || wv64 < 128) {
        // (x keeps the chunk offset in the per-lane offset: its halo rows in front of a chunk have NEGATIVE offsets from the chunk's first row, which a
        // shifted descriptor base would turn into out-of-range = zero)
        const uintptr_t p = (uintptr_t)(a.x + (size_t)b * a.Lin * a.Cin);
        const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, xbytes, 0x00020000);
        gn_buffer_load_lds(srd, (lptr_t)(stg + (k * NT + wv64) * 4), 16, xoff[k] + cb * 32 * a.Cin * 4, 0, 0, 0);
      }
    } else {
      constexpr int it = k - X_ITEMS;
      // dy: the chunk's row offset goes into the descriptor's BASE (and out of its size), so the per-lane offset is loop-invariant (no vector add per
      // piece); the range check still zero-fills the rows past the end of the batch element
      const int cbo = cb * 32 * a.Cout * 4;
      const uintptr_t p = (uintptr_t)(a.dy + (size_t)b * a.M * a.Cout) + (uintptr_t)(unsigned)cbo;
      const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, ybytes - cbo, 0x00020000);
      gn_buffer_load_lds(srd, (lptr_t)(stg + XS + (it * NT + wv64) * 4), 16, yoff[it], 0, 0, 0);
    }
  };
  auto dma_all = [&]() {
    dma_piece(std::integral_constant<int, 0>{}); dma_piece(std::integral_constant<int, 1>{}); dma_piece(std::integral_constant<int, 2>{});
    dma_piece(std::integral_constant<int, 3>{}); dma_piece(std::integral_constant<int, 4>{});
  };

  // this lane's operand addresses in stage 0: x row pairs (three row offsets; plane and half of the chunk are instruction offsets), dy pairs
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  unsigned base_a[3];
#pragma unroll
  for (int j2 = 0; j2 < 3; ++j2) base_a[j2] = lds0 + wave * 2560 + ((kq + j2) * 16 + n16) * 4;
  const unsigned base_b = lds0 + XS * 4 + (kq * 16 + n16) * 4;
  const unsigned long long k15 = 0x3fc000003fc00000ull, km15 = 0xbfc00000bfc00000ull;       // 1.5, -1.5 on both halves

  if (q_begin < q_end) {
    q_next = q_begin; st_next = 0; dma_all();
    q_next = min(q_begin + 1, q_end - 1); st_next = 1; dma_all();
    __syncthreads();                                // drains the LDS-DMA (vmcnt(0)) in front of the barrier

    f32x2 V0[6], V1[6], d[6], D[2][6];
    WinoT tt;
    {
      const char* sb = reinterpret_cast<const char*>(smem);
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const float* pp = reinterpret_cast<const float*>(sb + (base_a[j >> 1] - lds0) + (j & 1) * 1280);
        d[j][0] = pp[0]; d[j][1] = pp[64];
      }
      wino_bt_all(d, V0, tt, k15, km15);
      const float* pb = reinterpret_cast<const float*>(sb + (base_b - lds0));
      D[0][0][0] = pb[0]; D[0][0][1] = pb[64]; D[0][5][0] = pb[256]; D[0][5][1] = pb[320];
      wino_a_piece<0>(D[0][0], D[0][5], D[0][1], D[0][2], D[0][3], D[0][4]); wino_a_piece<1>(D[0][0], D[0][5], D[0][1], D[0][2], D[0][3], D[0][4]);
      wino_a_piece<2>(D[0][0], D[0][5], D[0][1], D[0][2], D[0][3], D[0][4]); wino_a_piece<3>(D[0][0], D[0][5], D[0][1], D[0][2], D[0][3], D[0][4]);
    }
    int st = 0;
    in_loop = true;
    for (int q = q_begin; q < q_end; ++q) {
      const int st1 = st == 2 ? 0 : st + 1, st2 = st1 == 2 ? 0 : st1 + 1;
      q_next = min(q + 2, q_end - 1);               // stage st2 held chunk - 1: every wave is past its last read of it
      st_next = st2;
      unsigned addr_a[3], addr_a_next[3];
#pragma unroll
      for (int j2 = 0; j2 < 3; ++j2) { addr_a[j2] = base_a[j2] + st * STAGE_BYTES; addr_a_next[j2] = base_a[j2] + st1 * STAGE_BYTES; }
      const unsigned addr_b = base_b + st * STAGE_BYTES, addr_b_next = base_b + st1 * STAGE_BYTES;
      WgPair<0, ABL>::template run<3, 2>(acc, V0, V1, d, tt, D, addr_b, addr_b_next, addr_a, addr_a_next, k15, km15, dma_piece);
      WgPair<1, ABL>::template run<3, 2>(acc, V1, V0, d, tt, D, addr_b, addr_b_next, addr_a, addr_a_next, k15, km15, dma_piece);
      if constexpr (!(ABL & 4)) __syncthreads();
      st = st1;
    }
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[5][0]), "+v"(acc[5][1]), "+v"(acc[5][2]), "+v"(acc[5][3]));
  }

  // partial slabs [split][point][Cin][Cout]; lane (n16, kq) holds rows ci = 4 kq + r of column n16 of each 16 x 16 tile
  const size_t cc = (size_t)a.Cin * a.Cout;
#pragma unroll
  for (int p = 0; p < 6; ++p) {
    float* dst = a.part + ((size_t)split * 6 + p) * cc + (size_t)(ci0 + wave * 16 + 4 * kq) * a.Cout + co0 + n16;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[(size_t)r * a.Cout + ct * 16] = acc[p][ct][r];
  }
#endif
}

// dW_q = sum_p G[p][q] * (sum over splits of Q_p), fp64, rounded once.  G: conv_wino.hip's wino_u_kernel.
__global__ void wgrad_wino_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, size_t cc, int splits) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cc) return;
  double Q[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll 4
  for (int s = 0; s < splits; ++s)                  // (four splits' loads in flight: the pass is latency-bound, not byte-bound)
#pragma unroll
    for (int p = 0; p < 6; ++p) Q[p] += (double)part[((size_t)s * 6 + p) * cc + i];
  // (the partial sums of points 0, 1, 2, 5 are half of the integer form's: wino_piece scales those rows by 1/2)
  const double e = (Q[1] + Q[2]) * (1.0 / 3.0), o = (Q[1] - Q[2]) * (1.0 / 3.0);
  dw[i] = (float)(Q[0] + e + Q[3] * (16.0 / 15.0) + Q[4] * (1.0 / 30.0));
  dw[cc + i] = (float)(o + Q[3] * (8.0 / 15.0) - Q[4] * (2.0 / 30.0));
  dw[2 * cc + i] = (float)(e + Q[3] * (4.0 / 15.0) + Q[4] * (4.0 / 30.0));
  dw[3 * cc + i] = (float)(o + Q[3] * (2.0 / 15.0) - Q[4] * (8.0 / 30.0));
  dw[4 * cc + i] = (float)(e + Q[3] * (1.0 / 15.0) + Q[4] * (16.0 / 30.0) + Q[5]);
}

bool wgrad_wino_supported(const WgradArgs& a) {
  if (a.ntaps != 5 || a.in_stride != 1 || a.Cin % 64 || a.Cout % 64) return false;
  for (int j = 0; j < 5; ++j)
    if (a.off[j] != a.off[0] + j) return false;
  return (size_t)a.Lin * a.Cin * 4 < 0x40000000ull && (size_t)a.M * a.Cout * 4 < 0x40000000ull;
}

size_t wgrad_wino_workspace_bytes(int B, int M, int Cin, int Cout) {
  int s, cps;
  wgrad_split_plan(B, M, Cin, Cout, 64, 64, &s, &cps);
  return (size_t)s * 6 * Cin * Cout * sizeof(float);
}

int wgrad_wino_run(WgradArgs& a, float* dw, size_t ws_bytes, hipStream_t s) {
  if (!wgrad_wino_supported(a)) {
    set_error("wgrad_wino: unsupported shape");
    return GN_EINVAL;
  }
  int splits;
  wgrad_split_plan(a.B, a.M, a.Cin, a.Cout, 64, 64, &splits, &a.chunks_per_split);
  if (ws_bytes < (size_t)splits * 6 * a.Cin * a.Cout * sizeof(float)) {
    set_error("wgrad_wino: workspace too small");
    return GN_EWORKSPACE;
  }
  const int cpb = (a.M + 31) / 32;
  constexpr size_t lds = 3 * sizeof(float) * (4 * 2 * 20 * 16 + 4 * 2 * 16 * 16);
  dim3 grid(a.Cin / 64, a.Cout / 64, splits);
  prof_begin(s);
#ifdef GN_ABLATION
  static const int abl = getenv("GN_WGWINO_ABL") ? atoi(getenv("GN_WGWINO_ABL")) : 0;
  if (abl == 1) hipLaunchKernelGGL(wgrad_wino_kernel<1>, grid, dim3(256), lds, s, a, a.off[0], cpb);
  else if (abl == 2) hipLaunchKernelGGL(wgrad_wino_kernel<2>, grid, dim3(256), lds, s, a, a.off[0], cpb);
  else if (abl == 8) hipLaunchKernelGGL(wgrad_wino_kernel<8>, grid, dim3(256), lds, s, a, a.off[0], cpb);
  else if (abl == 15) hipLaunchKernelGGL(wgrad_wino_kernel<15>, grid, dim3(256), lds, s, a, a.off[0], cpb);
  else
#endif
  hipLaunchKernelGGL(wgrad_wino_kernel<0>, grid, dim3(256), lds, s, a, a.off[0], cpb);
  // flop = what the kernel EXECUTES (0.6 of the algorithmic count)
  prof_end(s, 0.6 * 2.0 * a.B * (double)a.M * 5 * a.Cin * a.Cout, 6, 4.0 * ((double)a.B * a.Lin * a.Cin + (double)a.B * a.M * a.Cout + 5.0 * a.Cin * a.Cout));
  int rc = check_launch("wgrad_wino");
  if (rc) return rc;
  const size_t cc = (size_t)a.Cin * a.Cout;
  hipLaunchKernelGGL(wgrad_wino_reduce_kernel, dim3(cdiv(cc, 256)), dim3(256), 0, s, a.part, dw, cc, splits);
  return check_launch("wgrad_wino_reduce");
}

}  // namespace gn
