// Transform-domain weight gradient of the STRIDE-2 5-tap layers: the transposed form of conv_wino_s2.hip's F(2,3) + F(2,2),
//
//   Q_p[ci, co] = sum_{b, tau} V_p[b, tau, ci] * D_p[b, tau, co]      (seven points per output pair instead of ten taps)
//   V = the forward kernel's input transform of the seven x rows 4 tau + off0 .. + 6:  a0 - a2, a1 + a2, a2 - a1, a1 - a3 | b0 - b1, b1, b2 - b1
//   D = A^T of the dy pair (e0, e1) = (dy[2 tau], dy[2 tau + 1]):  e0, e0 + e1, e0 - e1, (-)e1 | e0, e0 + e1, e1   -- TWO packed additions; the operands repeat
//   dW (reduce pass, fp64): w0 = Q0 + (Q1 + Q2) / 2, w2 = (Q1 - Q2) / 2, w4 = (Q1 + Q2) / 2 - Q3 | w1 = Q4 + Q5, w3 = Q5 + Q6      (Q3 carries +e1: sign here)
//
// Kernel = wgrad_wino.hip's: v_mfma_f32_16x16x4_f32, wave = 16 ci x 64 co x 7 points, block = 4 waves (64 x 64), operands of two consecutive k-steps (tiles kq and
// kq + 4) as register pairs out of one ds_read2st64_b32, transforms packed, the dy pair one column tile ahead of its 14 MFMAs, three LDS stages.  The x image has
// FOUR row planes (row mod 4) of 12 rows x 16 channels per wave; a K-chunk is ONE pair-step (8 tiles = 16 output rows): 16 KiB per stage, 56 MFMAs per barrier.
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include "common.h"
#include "wino_common.h"

namespace gn {

__device__ __forceinline__ void wg2_slot(f32x4& c, float a, float b) {
  asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b) : "memory");
}
template <int O>
__device__ __forceinline__ void wg2_slot_e(f32x4& c, float a, float b, f32x2& e0, f32x2& e1, unsigned addr_b) {
  asm volatile(
      "ds_read2st64_b32 %1, %5 offset0:%6 offset1:%7\n\t"
      "ds_read2st64_b32 %2, %5 offset0:%8 offset1:%9\n\t"
      "v_mfma_f32_16x16x4_f32 %0, %3, %4, %0"
      : "+v"(c), "=&v"(e0), "=&v"(e1)
      : "v"(a), "v"(b), "v"(addr_b), "i"(O), "i"(O + 1), "i"(O + 2), "i"(O + 3)
      : "memory");
}
// pair-step slot 0: also the seven raw x row pairs of the NEXT chunk: row j in plane j & 3 (3 units of 256 bytes each) at plane row + (j >> 2)
template <int O>
__device__ __forceinline__ void wg2_slot_ea(f32x4& c, float a, float b, f32x2& e0, f32x2& e1, unsigned addr_b, f32x2 (&d)[7], const unsigned (&addr_a)[2]) {
  asm volatile(
      "ds_read2st64_b32 %1, %12 offset0:%15 offset1:%16\n\t"
      "ds_read2st64_b32 %2, %12 offset0:%17 offset1:%18\n\t"
      "ds_read2st64_b32 %3, %13 offset0:0 offset1:1\n\t"
      "ds_read2st64_b32 %4, %13 offset0:3 offset1:4\n\t"
      "ds_read2st64_b32 %5, %13 offset0:6 offset1:7\n\t"
      "ds_read2st64_b32 %6, %13 offset0:9 offset1:10\n\t"
      "ds_read2st64_b32 %7, %14 offset0:0 offset1:1\n\t"
      "ds_read2st64_b32 %8, %14 offset0:3 offset1:4\n\t"
      "ds_read2st64_b32 %9, %14 offset0:6 offset1:7\n\t"
      "v_mfma_f32_16x16x4_f32 %0, %10, %11, %0"
      : "+v"(c), "=&v"(e0), "=&v"(e1), "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6])
      : "v"(a), "v"(b), "v"(addr_b), "v"(addr_a[0]), "v"(addr_a[1]), "i"(O), "i"(O + 1), "i"(O + 2), "i"(O + 3)
      : "memory");
}
// slot 14: its reads go behind the raw x rows, so a wait for all but its own two retires them; they are operands so that their readers depend on it
template <int O>
__device__ __forceinline__ void wg2_slot_e_wd(f32x4& c, float a, float b, f32x2& e0, f32x2& e1, unsigned addr_b, f32x2 (&d)[7]) {
  asm volatile(
      "ds_read2st64_b32 %1, %12 offset0:%13 offset1:%14\n\t"
      "ds_read2st64_b32 %2, %12 offset0:%15 offset1:%16\n\t"
      "s_waitcnt lgkmcnt(2)\n\t"
      "v_mfma_f32_16x16x4_f32 %0, %10, %11, %0"
      : "+v"(c), "=&v"(e0), "=&v"(e1), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6])
      : "v"(a), "v"(b), "v"(addr_b), "i"(O), "i"(O + 1), "i"(O + 2), "i"(O + 3)
      : "memory");
}
template <int WAIT>
__device__ __forceinline__ void wg2_slot_w(f32x4& c, float a, float b, f32x2& e0, f32x2& e1) {
  asm volatile("s_waitcnt lgkmcnt(%5)\n\tv_mfma_f32_16x16x4_f32 %0, %3, %4, %0" : "+v"(c), "+v"(e0), "+v"(e1) : "v"(a), "v"(b), "i"(WAIT) : "memory");
}

#define GN_PK_ADD2(o, x, y) asm volatile("v_pk_add_f32 %0, %1, %2" : "=&v"(o) : "v"(x), "v"(y))
#define GN_PK_COPY2(o, x) asm volatile("v_pk_mul_f32 %0, %1, 1.0 op_sel_hi:[1,0]" : "=&v"(o) : "v"(x))
template <int K>
__device__ __forceinline__ void wg2_x_piece(const f32x2 (&d)[7], f32x2 (&v)[7]) {
  if constexpr (K == 0) GN_PK_SUB(v[0], d[0], d[4]);
  else if constexpr (K == 1) GN_PK_ADD2(v[1], d[2], d[4]);
  else if constexpr (K == 2) GN_PK_SUB(v[2], d[4], d[2]);
  else if constexpr (K == 3) GN_PK_SUB(v[3], d[2], d[6]);
  else if constexpr (K == 4) GN_PK_SUB(v[4], d[1], d[3]);
  else if constexpr (K == 5) GN_PK_COPY2(v[5], d[3]);
  else GN_PK_SUB(v[6], d[5], d[3]);
}

// One chunk (= one pair-step: 8 tiles, two k-steps) of a wave: 56 MFMA slots ordered column tile (14 each) > k-step > point.
// D[buffer][0..3] = e0, e1, e0 + e1, e0 - e1 of a column tile; the B operand of point p is D[.][BI[p]].
struct Wg2Chunk {
  template <int NPIECES, int I = 0, class DMA>
  static __device__ __forceinline__ void run(f32x4 (&acc)[7][4], const f32x2 (&v)[7], f32x2 (&vn)[7], f32x2 (&d)[7], f32x2 (&D)[2][4], unsigned addr_b,
                                             unsigned addr_b_next, const unsigned (&addr_a_next)[2], DMA& dma) {
    if constexpr (I < 56) {
      constexpr int CT = I / 14, G = I % 14, S = G / 7, P = G % 7;
      constexpr int BI = (P == 0 || P == 4) ? 0 : ((P == 1 || P == 5) ? 2 : (P == 2 ? 3 : 1));
      f32x2(&dc)[4] = D[CT & 1];
      f32x2(&dn)[4] = D[(CT + 1) & 1];
      const float av = v[P][S], bv = dc[BI][S];
      constexpr int OE = (CT < 3) ? (CT + 1) * 4 : 0;                         // the next column tile's dy pair (behind the last one: the next chunk's first)
      if constexpr (I == 0) wg2_slot_ea<OE>(acc[P][CT], av, bv, dn[0], dn[1], addr_b, d, addr_a_next);
      else if constexpr (I == 14) wg2_slot_e_wd<OE>(acc[P][CT], av, bv, dn[0], dn[1], addr_b, d);
      else if constexpr (G == 0) {
        if constexpr (CT == 3) wg2_slot_e<OE>(acc[P][CT], av, bv, dn[0], dn[1], addr_b_next);
        else wg2_slot_e<OE>(acc[P][CT], av, bv, dn[0], dn[1], addr_b);
      } else if constexpr (G == 4) wg2_slot_w<(I == 4 ? 7 : 0)>(acc[P][CT], av, bv, dn[0], dn[1]);
      else wg2_slot(acc[P][CT], av, bv);
      if constexpr (G == 4 && CT < NPIECES) dma(std::integral_constant<int, CT>{});
      // vector instructions in RUNS (alone between two MFMAs of a wave one costs 16 cycles, in a run 7: scripts/valu_rate.hip): the dy pair's two behind
      // slot 5 of every column tile, the next chunk's x transform as 4 + 3 behind slot 9 of column tiles 1 and 2
      if constexpr (G == 5) {
        GN_PK_ADD2(dn[2], dn[0], dn[1]);
        GN_PK_SUB(dn[3], dn[0], dn[1]);
      }
      if constexpr (CT == 1 && G == 9) { wg2_x_piece<0>(d, vn); wg2_x_piece<1>(d, vn); wg2_x_piece<2>(d, vn); wg2_x_piece<3>(d, vn); }
      if constexpr (CT == 2 && G == 9) { wg2_x_piece<4>(d, vn); wg2_x_piece<5>(d, vn); wg2_x_piece<6>(d, vn); }
      run<NPIECES, I + 1>(acc, v, vn, d, D, addr_b, addr_b_next, addr_a_next, dma);
    }
  }
};

__global__ __launch_bounds__(256, 2) void wgrad_wino_s2_kernel(WgradArgs a, int off0, int cpb) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NT = 256;
  constexpr int XS = 4 * 4 * 12 * 16;              // floats: 4 sub-slabs (16 ci each) x 4 row planes x 12 rows (9 used) x 16 channels
  constexpr int YS = 4 * 2 * 8 * 16;               // 4 column tiles x 2 row planes x 8 tiles x 16 columns
  constexpr int BUF = XS + YS;
  constexpr int STAGE_BYTES = BUF * 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  typedef __attribute__((address_space(3))) void* lptr_t;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n16 = lane & 15, kq = lane >> 4;
  const int ci0 = blockIdx.x * 64, co0 = blockIdx.y * 64, split = blockIdx.z;
  const int total = a.B * cpb;
  const int q_begin = split * a.chunks_per_split, q_end = min(q_begin + a.chunks_per_split, total);

  f32x4 acc[7][4];
#pragma unroll
  for (int p = 0; p < 7; ++p)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[p][ct][r] = 0.f;

  constexpr int X_ITEMS = 3;                 // + one piece of dy
  int xoff[X_ITEMS], yoff;
#pragma unroll
  for (int it = 0; it < X_ITEMS; ++it) {
    const int g = tid + it * NT;
    const int sub = g / 192, rem = g % 192, plane = rem / 48, prow = (rem % 48) >> 2, c4 = rem & 3;
    xoff[it] = ((4 * prow + plane + off0) * a.Cin + ci0 + sub * 16 + 4 * c4) * 4;
  }
  {
    const int g = tid;
    const int ct = g >> 6, rem = g & 63, plane = rem >> 5, tile = (rem & 31) >> 2, c4 = rem & 3;
    yoff = ((2 * tile + plane) * a.Cout + co0 + ct * 16 + 4 * c4) * 4;
  }
  const int xbytes = __builtin_amdgcn_readfirstlane(a.Lin * a.Cin * 4), ybytes = __builtin_amdgcn_readfirstlane(a.M * a.Cout * 4);
  int q_next = 0, st_next = 0;
  const int wv64 = __builtin_amdgcn_readfirstlane(tid & ~63);       // the wave's first thread, in a scalar register: the LDS-DMA destination (M0) is then scalar arithmetic
  auto dma_piece = [&](auto kc) {
    constexpr int k = decltype(kc)::value;
    const int b = __builtin_amdgcn_readfirstlane(q_next / cpb), cb = __builtin_amdgcn_readfirstlane(q_next % cpb);
    float* stg = smem + st_next * BUF;
    if constexpr (k < X_ITEMS) {
      const uintptr_t p = (uintptr_t)(a.x + (size_t)b * a.Lin * a.Cin);
      const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, xbytes, 0x00020000);
      gn_buffer_load_lds(srd, (lptr_t)(stg + (k * NT + wv64) * 4), 16, xoff[k] + cb * 32 * a.Cin * 4, 0, 0, 0);
    } else {
      // dy: the chunk's row offset in the descriptor's base and size, as in wgrad_wino.hip (x cannot: its halo rows sit at negative offsets)
      const int cbo = cb * 16 * a.Cout * 4;
      const uintptr_t p = (uintptr_t)(a.dy + (size_t)b * a.M * a.Cout) + (uintptr_t)(unsigned)cbo;
      const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, ybytes - cbo, 0x00020000);
      gn_buffer_load_lds(srd, (lptr_t)(stg + XS + wv64 * 4), 16, yoff, 0, 0, 0);
    }
  };
  auto dma_all = [&]() {
    dma_piece(std::integral_constant<int, 0>{}); dma_piece(std::integral_constant<int, 1>{}); dma_piece(std::integral_constant<int, 2>{});
    dma_piece(std::integral_constant<int, 3>{});
  };

  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  unsigned base_a[2];
#pragma unroll
  for (int j4 = 0; j4 < 2; ++j4) base_a[j4] = lds0 + wave * 3072 + ((kq + j4) * 16 + n16) * 4;
  const unsigned base_b = lds0 + XS * 4 + (kq * 16 + n16) * 4;

  if (q_begin < q_end) {
    q_next = q_begin; st_next = 0; dma_all();
    q_next = min(q_begin + 1, q_end - 1); st_next = 1; dma_all();
    __syncthreads();

    f32x2 V0[7], V1[7], d[7], D[2][4];
    {
      const char* sb = reinterpret_cast<const char*>(smem);
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const float* pp = reinterpret_cast<const float*>(sb + (base_a[j >> 2] - lds0) + (j & 3) * 768);
        d[j][0] = pp[0]; d[j][1] = pp[64];
      }
      wg2_x_piece<0>(d, V0); wg2_x_piece<1>(d, V0); wg2_x_piece<2>(d, V0); wg2_x_piece<3>(d, V0); wg2_x_piece<4>(d, V0); wg2_x_piece<5>(d, V0); wg2_x_piece<6>(d, V0);
      const float* pb = reinterpret_cast<const float*>(sb + (base_b - lds0));
      D[0][0][0] = pb[0]; D[0][0][1] = pb[64]; D[0][1][0] = pb[128]; D[0][1][1] = pb[192];
      GN_PK_ADD2(D[0][2], D[0][0], D[0][1]);
      GN_PK_SUB(D[0][3], D[0][0], D[0][1]);
    }
    int st = 0;
    for (int q = q_begin; q < q_end; q += 2) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if (half == 1 && q + 1 >= q_end) break;
        const int st1 = st == 2 ? 0 : st + 1, st2 = st1 == 2 ? 0 : st1 + 1;
        q_next = min(q + half + 2, q_end - 1);
        st_next = st2;
        unsigned addr_a_next[2];
#pragma unroll
        for (int j4 = 0; j4 < 2; ++j4) addr_a_next[j4] = base_a[j4] + st1 * STAGE_BYTES;
        const unsigned addr_b = base_b + st * STAGE_BYTES, addr_b_next = base_b + st1 * STAGE_BYTES;
        if (half == 0) Wg2Chunk::run<4>(acc, V0, V1, d, D, addr_b, addr_b_next, addr_a_next, dma_piece);
        else Wg2Chunk::run<4>(acc, V1, V0, d, D, addr_b, addr_b_next, addr_a_next, dma_piece);
        __syncthreads();
        st = st1;
      }
    }
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[6][0]), "+v"(acc[6][1]), "+v"(acc[6][2]), "+v"(acc[6][3]));
  }

  const size_t cc = (size_t)a.Cin * a.Cout;
#pragma unroll
  for (int p = 0; p < 7; ++p) {
    float* dst = a.part + ((size_t)split * 7 + p) * cc + (size_t)(ci0 + wave * 16 + 4 * kq) * a.Cout + co0 + n16;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[(size_t)r * a.Cout + ct * 16] = acc[p][ct][r];
  }
#endif
}

__global__ void wgrad_wino_s2_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, size_t cc, int splits) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cc) return;
  double Q[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll 4
  for (int s = 0; s < splits; ++s)                  // (four splits' loads in flight: the pass is latency-bound, not byte-bound)
#pragma unroll
    for (int p = 0; p < 7; ++p) Q[p] += (double)part[((size_t)s * 7 + p) * cc + i];
  dw[i] = (float)(Q[0] + 0.5 * (Q[1] + Q[2]));
  dw[cc + i] = (float)(Q[4] + Q[5]);
  dw[2 * cc + i] = (float)(0.5 * (Q[1] - Q[2]));
  dw[3 * cc + i] = (float)(Q[5] + Q[6]);
  dw[4 * cc + i] = (float)(0.5 * (Q[1] + Q[2]) - Q[3]);
}

static void s2_plan(int B, int M, int Cin, int Cout, int* splits, int* cps16, int* cpb16) {
  int s, cps32;
  wgrad_split_plan(B, M, Cin, Cout, 64, 64, &s, &cps32);        // the direct kernel's plan in 32-row chunks; a chunk here is 16 output rows
  *cpb16 = (M + 15) / 16;
  const int cpb32 = (M + 31) / 32;
  // keep the same partition of every batch element: 2 chunks here per chunk there (the last one of an element may be short)
  if (cps32 % cpb32 == 0) *cps16 = (cps32 / cpb32) * *cpb16;    // whole batch elements per split
  else *cps16 = 2 * cps32;                                       // ranges inside an element
  const long total = (long)B * *cpb16;
  *splits = (int)((total + *cps16 - 1) / *cps16);
}

bool wgrad_wino_s2_supported(const WgradArgs& a) {
  if (a.ntaps != 5 || a.in_stride != 2 || a.Cin % 64 || a.Cout % 64) return false;
  for (int j = 0; j < 5; ++j)
    if (a.off[j] != a.off[0] + j) return false;
  return (size_t)a.Lin * a.Cin * 4 < 0x40000000ull && (size_t)a.M * a.Cout * 4 < 0x40000000ull;
}

size_t wgrad_wino_s2_workspace_bytes(int B, int M, int Cin, int Cout) {
  int s, cps, cpb;
  s2_plan(B, M, Cin, Cout, &s, &cps, &cpb);
  return (size_t)s * 7 * Cin * Cout * sizeof(float);
}

int wgrad_wino_s2_run(WgradArgs& a, float* dw, size_t ws_bytes, hipStream_t s) {
  if (!wgrad_wino_s2_supported(a)) {
    set_error("wgrad_wino_s2: unsupported shape");
    return GN_EINVAL;
  }
  int splits, cpb;
  s2_plan(a.B, a.M, a.Cin, a.Cout, &splits, &a.chunks_per_split, &cpb);
  if (ws_bytes < (size_t)splits * 7 * a.Cin * a.Cout * sizeof(float)) {
    set_error("wgrad_wino_s2: workspace too small");
    return GN_EWORKSPACE;
  }
  constexpr size_t lds = 3 * sizeof(float) * (4 * 4 * 12 * 16 + 4 * 2 * 8 * 16);
  dim3 grid(a.Cin / 64, a.Cout / 64, splits);
  prof_begin(s);
  hipLaunchKernelGGL(wgrad_wino_s2_kernel, grid, dim3(256), lds, s, a, a.off[0], cpb);
  prof_end(s, 0.7 * 2.0 * a.B * (double)a.M * 5 * a.Cin * a.Cout, 8, 4.0 * ((double)a.B * a.Lin * a.Cin + (double)a.B * a.M * a.Cout + 5.0 * a.Cin * a.Cout));
  int rc = check_launch("wgrad_wino_s2");
  if (rc) return rc;
  const size_t cc = (size_t)a.Cin * a.Cout;
  hipLaunchKernelGGL(wgrad_wino_s2_reduce_kernel, dim3(cdiv(cc, 256)), dim3(256), 0, s, a.part, dw, cc, splits);
  return check_launch("wgrad_wino_s2_reduce");
}

}  // namespace gn
