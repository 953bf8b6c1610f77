// EXPERIMENTAL, opt-in (never on the default path): the Conv1D weight gradient on the bf16 matrix cores by operand splitting -- the counterpart of
// conv_bf16x3.hip (same three-piece split, same six products, smallest first, fp32 accumulation) for
//     dW[j][ci][co] = sum over (b, m) of x[b, IS * m + off_j, ci] * dy[b, m, co].
//
// The reduction index of this GEMM is the ROW (b, m), and v_mfma_f32_32x32x16_bf16 wants every lane to hold 8 consecutive reduction indices of one
// matrix row: both operands are needed row-contiguous per channel, the transpose of how the tensors lie in HBM.  So the split pass also transposes:
//     x   -> planes[3][B][IS][chunk][Cin ][40]   row r of chunk k, parity class par = x row t = IS * (32 k + r - S) + par, zero outside [0, L)
//     dy  -> planes[3][B]    [chunk][Cout][32]   row r of chunk k = dy row 32 k + r, zero from row M on
// in records of one K-chunk x one channel (x: 32 output rows + 8 rows of halo duplicated from the next chunk = 80 bytes; dy: 64 bytes), so that what a
// block stages per chunk is ONE contiguous run per plane (128 channels x 80 bytes, 64 x 64): every LDS-DMA instruction moves 1 KiB of whole cache lines.  (A plain
// [channel][row] plane was built first: 16-byte granules at a 4-KiB stride, 203 TFLOP/s against 286 with the staging switched off.)  S is chosen so
// that tap j of output row m reads row m + shift_j of parity class par_j, shift_j in 0 .. 4 (stride 1: shift_j = j; stride 2: the even / odd input
// rows are two classes and shift_j <= 2).  A tap is then a SHIFT of the x operand along the
// reduction index by 0 .. 4 bf16 -- not 16-byte aligned, which is what kept this kernel unwritten until round 4: the shift is done in registers.  A
// lane reads the 16 aligned rows that cover all five taps (two ds_read_b128) ONCE per 16-row step and forms the five fragments from those eight dwords:
// even shifts are dword selections, odd ones four v_alignbit_b32 each.  Per 16-row step a wave (64 ci x 32 co x 5 taps) issues 60 MFMAs against
// 15 ds_read_b128 and ~50 VALU instructions: the kernel is matrix-core bound with the LDS nearly idle (the forward kernel is LDS-bound at 12 reads
// per 24 MFMAs).
//
// Block = 4 waves (2 x 2) = 128 ci x 64 co x 5 taps; K-chunk = 32 rows of one batch element (the exact kernel's chunking and K-split plan, so the
// partial slabs, the fixed-order reduce pass and the workspace size are the exact path's own); LDS image per stage: [plane][class][channel][40 rows]
// bf16, 80 bytes per channel -- consecutive channels 20 banks apart: the 16-byte fragment reads of 16 lanes are conflict-free -- filled by LDS-DMA
// (a lane's 16 bytes = 8 rows of one channel; 5 lanes per channel).  Two stages; the staging of chunk c + 2 is issued behind the barrier in the
// second half of chunk c and has a whole chunk (1.6 us) to land.
#include <algorithm>
#include <stdlib.h>
#include "common.h"

namespace gn {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ inline void wsplit3(float a, unsigned& h, unsigned& m, unsigned& l) {       // conv_bf16x3.hip's split3, pieces as the low 16 bits
  const __bf16 bh = (__bf16)a;
  float r1 = a - (float)bh;
  if (!(fabsf(a) < INFINITY)) r1 = 0.f;
  const __bf16 bm = (__bf16)r1;
  const float r2 = r1 - (float)bm;
  const __bf16 bl = (__bf16)r2;
  h = __builtin_bit_cast(unsigned short, bh);
  m = __builtin_bit_cast(unsigned short, bm);
  l = __builtin_bit_cast(unsigned short, bl);
}

// src (B, L, C) fp32 -> planes[3][B][IS][C][LP] bf16 (see the header).  Block = 128 rows x 64 channels of one (batch element, parity class): rows are
// read as 256-byte runs, split, transposed through LDS (row pairs packed into dwords; 65-dword channel stride: writes and reads conflict-free) and
// written as 256-byte runs (128 rows of one channel).
__global__ __launch_bounds__(256) void split_t_kernel(const float* __restrict__ src, unsigned short* __restrict__ planes, int B, int L, int C, int cpb, int S, int IS,
                                                      int rec) {
  const int LP = 32 * cpb + 8;
  __shared__ unsigned T[3][64][65];
  const int tid = threadIdx.x;
  // channel tiles fastest: the blocks in flight together read whole rows of src (one 256-byte piece each) and write adjacent 5-KiB runs
  const int p_tile = blockIdx.y * 128, c0 = blockIdx.x * 64;
  const int b = blockIdx.z / IS, par = blockIdx.z % IS;
  const int c4 = tid & 15, r2 = tid >> 4;
  const float* sb = src + (size_t)b * L * C + c0 + 4 * c4;
  // all eight row loads of the thread in flight before the first is used (one round trip per block instead of four)
  float4 v[4][2];
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int p0 = p_tile + 2 * (pass * 16 + r2);
    const int t0 = IS * (p0 - S) + par, t1 = t0 + IS;
    v[pass][0] = *reinterpret_cast<const float4*>(sb + (size_t)min(max(t0, 0), L - 1) * C);      // branch-free: clamped address, zeroed below
    v[pass][1] = *reinterpret_cast<const float4*>(sb + (size_t)min(max(t1, 0), L - 1) * C);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int p0 = p_tile + 2 * (pass * 16 + r2);
    const int t0 = IS * (p0 - S) + par, t1 = t0 + IS;
    if (t0 < 0 || t0 >= L) v[pass][0] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t1 < 0 || t1 >= L) v[pass][1] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int pr = pass * 16 + r2;
    const float e0[4] = {v[pass][0].x, v[pass][0].y, v[pass][0].z, v[pass][0].w}, e1[4] = {v[pass][1].x, v[pass][1].y, v[pass][1].z, v[pass][1].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsigned h0, m0, l0, h1, m1, l1;
      wsplit3(e0[e], h0, m0, l0);
      wsplit3(e1[e], h1, m1, l1);
      T[0][4 * c4 + e][pr] = h0 | (h1 << 16);
      T[1][4 * c4 + e][pr] = m0 | (m1 << 16);
      T[2][4 * c4 + e][pr] = l0 | (l1 << 16);
    }
  }
  __syncthreads();
  // rec = 40: records of 32 rows + 8 rows of halo (x).  rec = 32: no halo, and the four 16-byte granules of a channel are stored at position
  // granule ^ ((channel >> 2) & 3): with 64-byte records that is what keeps the kernel's fragment reads bank-conflict free (dy)
  const size_t plane = (size_t)B * IS * cpb * C * rec;
  unsigned short* out = planes + (size_t)(b * IS + par) * cpb * C * rec;
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    const int idx = tid + k * 256;
    const int pl = idx >> 10, rem = idx & 1023, blk = rem & 3, c = (rem >> 2) & 63, kc = rem >> 8;      // 8-row block blk of chunk kc of the tile, channel c
    const int p = p_tile + 32 * kc + 8 * blk;
    if (p < LP) {
      const int chunk = p >> 5;
      uint4 v;
      v.x = T[pl][c][16 * kc + 4 * blk]; v.y = T[pl][c][16 * kc + 4 * blk + 1]; v.z = T[pl][c][16 * kc + 4 * blk + 2]; v.w = T[pl][c][16 * kc + 4 * blk + 3];
      const int pos = rec == 40 ? blk : (blk ^ ((c >> 2) & 3));
      if (chunk < cpb) *reinterpret_cast<uint4*>(out + pl * plane + ((size_t)chunk * C + c0 + c) * rec + 8 * pos) = v;
      if (rec == 40 && blk == 0 && chunk >= 1) *reinterpret_cast<uint4*>(out + pl * plane + ((size_t)(chunk - 1) * C + c0 + c) * 40 + 32) = v;      // the halo of the chunk before
    }
  }
}

// tap j (slab row offset j = off - minoff) -> (parity class, shift in rows of that class); PODD = pad_left & 1 (stride 2 only)
template <int IS, int PODD> __host__ __device__ constexpr int tap_par(int j) { return IS == 1 ? 0 : ((j + PODD) & 1); }
template <int IS, int PODD> __host__ __device__ constexpr int tap_shift(int j) { return IS == 1 ? j : ((j + PODD) >> 1); }

// rows SH .. SH + 7 of the 12 rows held in (lo, hi), two rows per dword
template <int SH>
__device__ __forceinline__ bf16x8 shifted_rows(const u32x4& lo, const uint2& hi) {
  static_assert(SH >= 0 && SH <= 4, "12 rows are read");
  const unsigned d[6] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y};
  u32x4 r;
  if constexpr (SH % 2 == 0) {
    r.x = d[SH / 2]; r.y = d[SH / 2 + 1]; r.z = d[SH / 2 + 2]; r.w = d[SH / 2 + 3];
  } else {
    r.x = __builtin_amdgcn_alignbit(d[(SH + 1) / 2], d[(SH - 1) / 2], 16);
    r.y = __builtin_amdgcn_alignbit(d[(SH + 1) / 2 + 1], d[(SH - 1) / 2 + 1], 16);
    r.z = __builtin_amdgcn_alignbit(d[(SH + 1) / 2 + 2], d[(SH - 1) / 2 + 2], 16);
    r.w = __builtin_amdgcn_alignbit(d[(SH + 1) / 2 + 3], d[(SH - 1) / 2 + 3], 16);
  }
  return __builtin_bit_cast(bf16x8, r);
}

// IB = 32-channel blocks of x per wave: 2 -> 4 waves of 64 ci x 32 co (one per SIMD), 1 -> 8 waves of 32 x 32 (two per SIMD: one wave's barrier,
// operand and VMEM-issue waits are covered by the other's MFMAs)
// WNN = waves across the output channels: 2 -> 128 ci x 64 co blocks; 4 (stride 2, IB = 1) -> 64 ci x 128 co: at stride 2 the x tile is staged twice (even and
// odd input rows), so the narrower x tile stages 54 instead of 73 KiB per chunk for the same MFMAs
template <int IS, int PODD, int ABL = 0, int IB = 2, int WNN = 2>
__global__ __launch_bounds__(512 / IB, 1) void wgrad_bf16x3_kernel(WgradArgs a, const unsigned short* __restrict__ xt, const unsigned short* __restrict__ dyt,
                                                              size_t x_plane, size_t dy_plane, int xcd_order) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NTAPS = 5, KC = 32;
  constexpr int TN = 32 * WNN, TC = 32 * IB * ((8 / IB) / WNN);
  static_assert(TC * TN == 128 * 64, "eight (four) waves of 32 (64) x 32");
  constexpr int CH_BYTES = 80;                              // 40 rows of one channel
  constexpr int A_PLANE = IS * TC * CH_BYTES, B_PLANE = TN * 64;        // dy: 32 rows per channel, granules swizzled (see split_t_kernel)
  constexpr int A_BYTES = 3 * A_PLANE, B_BYTES = 3 * B_PLANE, STAGE = A_BYTES + B_BYTES;
  static_assert(2 * STAGE <= 160 * 1024, "two stages must fit the CU's LDS");
  constexpr int QA_PLANE = IS * TC * 5 / 64, QB_PLANE = TN * 4 / 64;        // wave-level DMA instructions per plane: 10 (20 at stride 2) and 4
  static_assert((IS * TC * 5) % 64 == 0 && (TN * 4) % 64 == 0, "whole 1-KiB pieces");
  constexpr int QA = 3 * QA_PLANE, Q_TOTAL = QA + 3 * QB_PLANE;
  constexpr int NW = 8 / IB;                                // waves per block
  constexpr int Q_WAVE = (Q_TOTAL + NW - 1) / NW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave / WNN, wn = wave % WNN;
  const int i32 = lane & 31, h = lane >> 5;
  // block -> (Cin tile, Cout tile, K-split).  xcd_order: block i runs on XCD i mod 8; every XCD takes whole K-splits (the splits k, k + 8, ...) with all
  // their tiles one after the other, so each x / dy chunk crosses the fabric once and its reuse by the tiles is served by that XCD's L2
  int ct = blockIdx.x, nt_ = blockIdx.y, split = blockIdx.z;
  if (xcd_order) {
    const int tiles = gridDim.x * gridDim.y;
    const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int q = lin >> 3, tile = q % tiles;
    split = (q / tiles) * 8 + (lin & 7);
    ct = tile % (int)gridDim.x;
    nt_ = tile / (int)gridDim.x;
  }
  ct = __builtin_amdgcn_readfirstlane(ct); nt_ = __builtin_amdgcn_readfirstlane(nt_); split = __builtin_amdgcn_readfirstlane(split);
  const int c0 = ct * TC, n0 = nt_ * TN;

  int minoff = a.off[0];
#pragma unroll
  for (int j = 1; j < NTAPS; ++j) minoff = min(minoff, a.off[j]);

  f32x16 acc[NTAPS][IB];
#pragma unroll
  for (int j = 0; j < NTAPS; ++j)
#pragma unroll
    for (int ib = 0; ib < IB; ++ib)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][ib][r] = 0.f;

  const int cpb = (a.M + KC - 1) / KC;
  const int c_lo = split * a.chunks_per_split, c_hi = min(a.B * cpb, c_lo + a.chunks_per_split);
  const int n_chunks = max(c_hi - c_lo, 0);

  // staging: per plane a block needs ONE contiguous run of the split planes per chunk (and parity class): 128 channels x 80 bytes of x, 64 x 80 of dy.
  // Instruction q of a chunk (wave w issues q = w, w + 4, ...) moves 1 KiB of it; address = wave-uniform base + lane * 16.
  const char* srcp[Q_WAVE];
  int dst[Q_WAVE];
  int spar[Q_WAVE];                                         // x: parity class; dy: -1
#pragma unroll
  for (int i = 0; i < Q_WAVE; ++i) {
    int q = wave + NW * i;
    if (q >= Q_TOTAL) q = Q_TOTAL - 1;                      // padding instruction: repeats the last one
    if (q < QA) {
      const int pl = q / QA_PLANE, f = q % QA_PLANE, par = f / (QA_PLANE / IS), piece = f % (QA_PLANE / IS);
      srcp[i] = (const char*)(xt + pl * x_plane) + ((size_t)c0 * 40) * 2 + piece * 1024 + lane * 16;
      dst[i] = pl * A_PLANE + f * 1024;
      spar[i] = par;
    } else {
      const int qb = q - QA;
      const int pl = qb / QB_PLANE, piece = qb % QB_PLANE;
      srcp[i] = (const char*)(dyt + pl * dy_plane) + ((size_t)n0 * 32) * 2 + piece * 1024 + lane * 16;
      dst[i] = A_BYTES + pl * B_PLANE + piece * 1024;
      spar[i] = -1;
    }
  }
  bool in_loop = false;
  auto dma_chunk = [&](int ch, unsigned char* stage) {
    if ((ABL & 2) && in_loop) return;
    const int cid = c_lo + ch;
    const int b = __builtin_amdgcn_readfirstlane(cid / cpb), kc = __builtin_amdgcn_readfirstlane(cid % cpb);
    const size_t yb = ((size_t)b * cpb + kc) * a.Cout * 64;
#pragma unroll
    for (int i = 0; i < Q_WAVE; ++i) {
      const size_t xb = ((size_t)(b * IS + max(spar[i], 0)) * cpb + kc) * a.Cin * 80;
      gn_global_load_lds((gptr_t)(srcp[i] + (spar[i] >= 0 ? xb : yb)), (lptr_t)(stage + dst[i]), 16, 0, 0);
    }
  };

  // Fragment reads of one 16-row step, per plane: for each 32-channel block (and parity class) the aligned rows 16 * ks + 8 * h ... + 11 (a 16- and an
  // 8-byte load: the largest shift is 4 rows), and the 8 dy rows.
  struct Raw {
    u32x4 alo[3][IS][IB];
    uint2 ahi[3][IS][IB];
    u32x4 b[3];
  };
  auto read_plane = [&](const unsigned char* st, int ks, int p, Raw& r) {
    if ((ABL & 1) && in_loop) return;
    const int rowb = 32 * ks + 16 * h;
#pragma unroll
    for (int par = 0; par < IS; ++par)
#pragma unroll
      for (int ib = 0; ib < IB; ++ib) {
        const unsigned char* q = st + p * A_PLANE + (par * TC + wc * (32 * IB) + ib * 32 + i32) * CH_BYTES + rowb;
        r.alo[p][par][ib] = *reinterpret_cast<const u32x4*>(q);
        r.ahi[p][par][ib] = *reinterpret_cast<const uint2*>(q + 16);
      }
    r.b[p] = *reinterpret_cast<const u32x4*>(st + A_BYTES + p * B_PLANE + (wn * 32 + i32) * 64 + (((2 * ks + h) ^ ((i32 >> 2) & 3)) << 4));
  };
  // the five tap fragments of plane p: tap j = rows shift_j .. shift_j + 7 of its parity class
  auto prep = [&](const Raw& r, int p, bf16x8 (&f)[NTAPS][IB]) {
#pragma unroll
    for (int ib = 0; ib < IB; ++ib) {
      constexpr int P0 = tap_par<IS, PODD>(0), P1 = tap_par<IS, PODD>(1), P2 = tap_par<IS, PODD>(2), P3 = tap_par<IS, PODD>(3), P4 = tap_par<IS, PODD>(4);
      f[0][ib] = shifted_rows<tap_shift<IS, PODD>(0)>(r.alo[p][P0][ib], r.ahi[p][P0][ib]);
      f[1][ib] = shifted_rows<tap_shift<IS, PODD>(1)>(r.alo[p][P1][ib], r.ahi[p][P1][ib]);
      f[2][ib] = shifted_rows<tap_shift<IS, PODD>(2)>(r.alo[p][P2][ib], r.ahi[p][P2][ib]);
      f[3][ib] = shifted_rows<tap_shift<IS, PODD>(3)>(r.alo[p][P3][ib], r.ahi[p][P3][ib]);
      f[4][ib] = shifted_rows<tap_shift<IS, PODD>(4)>(r.alo[p][P4][ib], r.ahi[p][P4][ib]);
    }
  };
  auto mma = [&](const bf16x8 (&f)[NTAPS][IB], const u32x4& braw) {
    const bf16x8 bv = __builtin_bit_cast(bf16x8, braw);
#pragma unroll
    for (int j = 0; j < NTAPS; ++j)
#pragma unroll
      for (int ib = 0; ib < IB; ++ib) acc[j][ib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[j][ib], bv, acc[j][ib], 0, 0, 0);
  };
  // One 16-row step in three stages, products grouped by the x plane (the accumulators persist over all steps of a split, so the order of the six
  // products inside a step does not matter for the rounding): lo * hi | mid * (mid, hi) | hi * (lo, mid, hi) = 10 + 20 + 30 MFMAs.  The fragments of
  // the NEXT stage's plane are formed (VALU) under the current stage's MFMAs, the next step's operands are read (DS) under the first two stages, the
  // staging DMAs (VMEM) go out under the third: sched_group_barrier pins that interleave, one wave per SIMD has nobody else to hide behind.
  auto step = [&](const Raw& cur, Raw& nxt, const unsigned char* st_next, int ks_next, bf16x8 (&X)[NTAPS][IB], bf16x8 (&Y)[NTAPS][IB], bool barrier, int dma_ch,
                  unsigned char* dma_stage) {
    // stage 0: X = lo fragments of cur
    read_plane(st_next, ks_next, 2, nxt);
    read_plane(st_next, ks_next, 1, nxt);
    prep(cur, 1, Y);
    mma(X, cur.b[0]);
#pragma unroll
    for (int k = 0; k < 5 * IB; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, IS, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // stage 1: Y = mid fragments
    read_plane(st_next, ks_next, 0, nxt);
    prep(cur, 0, X);
    mma(Y, cur.b[1]);
    mma(Y, cur.b[0]);
#pragma unroll
    for (int k = 0; k < 10 * IB; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (k < 5 * IS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);                     // lgkmcnt(0): the next step's operands are in registers
    if (barrier) {
      // every wave has read all it needs of this chunk's stage: the chunk after next goes there; the next chunk has landed
      if (!(ABL & 4)) {
        __builtin_amdgcn_s_waitcnt(0x0F70);
        asm volatile("s_barrier" ::: "memory");             // (the builtin is IntrNoMem: LDS loads may move across it)
      }
      dma_chunk(dma_ch, dma_stage);
    }
    // stage 2: X = hi fragments; Y <- the next step's lo fragments
    prep(nxt, 2, Y);
    mma(X, cur.b[2]);
    mma(X, cur.b[1]);
    mma(X, cur.b[0]);
#pragma unroll
    for (int k = 0; k < 15 * IB; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (2 * Q_WAVE <= 15 * IB ? (k % 2 == 0 && k / 2 < Q_WAVE) : k < Q_WAVE) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);      // staging DMAs spread under the MFMAs
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  if (n_chunks > 0) {
    dma_chunk(0, smem_b);
    dma_chunk(min(1, n_chunks - 1), smem_b + STAGE);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0)
  asm volatile("s_barrier" ::: "memory");
  Raw r0, r1;
  bf16x8 X[NTAPS][IB], Y[NTAPS][IB];
  read_plane(smem_b, 0, 0, r0); read_plane(smem_b, 0, 1, r0); read_plane(smem_b, 0, 2, r0);
  __builtin_amdgcn_s_waitcnt(0xC07F);
  prep(r0, 2, X);
  in_loop = true;
  for (int ch = 0; ch < n_chunks; ++ch) {
    unsigned char* sa = smem_b + (ch & 1) * STAGE;
    const unsigned char* sb = smem_b + ((ch + 1) & 1) * STAGE;
    // past the end the staging repeats the last chunk into a free stage and the reads fetch operands nobody uses: no branches in the loop
    step(r0, r1, sa, 1, X, Y, true, min(ch + 2, n_chunks - 1), sa);
    step(r1, r0, sb, 0, Y, X, false, 0, nullptr);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);

  const int n = n0 + wn * 32 + i32;
#pragma unroll
  for (int j = 0; j < NTAPS; ++j) {
    int tap = 0;
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
      if (a.off[t] - minoff == j) tap = t;
    float* pj = a.part + ((size_t)split * NTAPS + tap) * a.Cin * a.Cout;
#pragma unroll
    for (int ib = 0; ib < IB; ++ib)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = c0 + wc * (32 * IB) + ib * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        pj[(size_t)c * a.Cout + n] = acc[j][ib][r];
      }
  }
#endif
}

static int wgrad_cpb(int M) { return (M + 31) / 32; }

size_t wgrad_bf16x3_workspace_bytes(int B, int M, int Cin, int Cout, int in_stride) {
  return (size_t)6 * B * wgrad_cpb(M) * ((size_t)40 * in_stride * Cin + (size_t)32 * Cout) + 256;
}

bool wgrad_bf16x3_supported(const WgradArgs& a) {
  if (a.ntaps != 5 || (a.in_stride != 1 && a.in_stride != 2) || a.Cin % 128 || a.Cout % 64 || a.M < 1) return false;
  // stride 2 stages (and splits) two input rows per output row: measured 187 TFLOP/s for the kernel, 157 with the split passes against the exact
  // kernel's 142 on 512 -> 1024, but 130 against 141 on 256 -> 512 -- only the wide layers take it
  if (a.in_stride == 2 && a.Cin < 512) return false;
  int minoff = a.off[0], maxoff = a.off[0];
  for (int j = 1; j < 5; ++j) {
    minoff = std::min(minoff, a.off[j]);
    maxoff = std::max(maxoff, a.off[j]);
  }
  return maxoff - minoff == 4 && minoff <= 0 && minoff >= -4;
}

// The split passes and the kernel; the caller (launch_wgrad, conv_mfma.hip) owns the K-split plan (a.chunks_per_split, splits) and the reduce pass.
int wgrad_bf16x3_run(const WgradArgs& a_in, int splits, void* ws, size_t ws_bytes, hipStream_t s) {
  WgradArgs a = a_in;
  const int xcd_order = splits % 8 == 0;
  if (ws_bytes < wgrad_bf16x3_workspace_bytes(a.B, a.M, a.Cin, a.Cout, a.in_stride)) {
    set_error("wgrad_bf16x3: workspace too small");
    return GN_EWORKSPACE;
  }
  const int IS = a.in_stride, cpb = wgrad_cpb(a.M), LP = 32 * cpb + 8;
  int minoff = a.off[0];
  for (int j = 1; j < 5; ++j) minoff = std::min(minoff, a.off[j]);
  const int pl = -minoff;
  const int S = IS == 1 ? pl : (pl + 1) / 2;
  const size_t x_plane = (size_t)a.B * IS * cpb * a.Cin * 40, dy_plane = (size_t)a.B * cpb * a.Cout * 32;
  unsigned short* xt = (unsigned short*)ws;
  unsigned short* dyt = xt + 3 * x_plane;
  if ((size_t)a.B * IS > 65535 || a.Cin / 64 > 65535 || a.Cout / 64 > 65535) {
    set_error("wgrad_bf16x3: bad grid");
    return GN_EINVAL;
  }
  hipLaunchKernelGGL(split_t_kernel, dim3(a.Cin / 64, cdiv(LP, 128), a.B * IS), dim3(256), 0, s, a.x, xt, a.B, a.Lin, a.Cin, cpb, S, IS, 40);
  hipLaunchKernelGGL(split_t_kernel, dim3(a.Cout / 64, cdiv(LP, 128), a.B), dim3(256), 0, s, a.dy, dyt, a.B, a.M, a.Cout, cpb, 0, 1, 32);
  int rc = check_launch("split_t");
  if (rc) return rc;
  dim3 grid(a.Cin / 128, a.Cout / 64, splits);
  const unsigned short* xc = xt;
  const unsigned short* dc = dyt;
  const bool wide_n = a.Cout % 128 == 0;            // stride 2: 64 ci x 128 co blocks stage 54 instead of 73 KiB per chunk (196.7 against 190.1 TFLOP/s)
#ifdef GN_ABLATION
  static const int abl = getenv("GN_WGBF_ABL") ? atoi(getenv("GN_WGBF_ABL")) : 0;       // timing ablations, stride 1 (results are wrong with any of them)
#else
  constexpr int abl = 0;
#endif
  prof_begin(s);
  if (IS == 1) {
    constexpr size_t lds = 2 * (size_t)(3 * 128 * 80 + 3 * 64 * 64);
#ifdef GN_ABLATION
    static unsigned long long d1 = 0, d2 = 0, d4 = 0;
    if (abl == 1) {
      allow_big_lds((const void*)wgrad_bf16x3_kernel<1, 0, 1>, &d1);
      hipLaunchKernelGGL((wgrad_bf16x3_kernel<1, 0, 1>), grid, dim3(256), lds, s, a, xc, dc, x_plane, dy_plane, xcd_order);
    } else if (abl == 2) {
      allow_big_lds((const void*)wgrad_bf16x3_kernel<1, 0, 2>, &d2);
      hipLaunchKernelGGL((wgrad_bf16x3_kernel<1, 0, 2>), grid, dim3(256), lds, s, a, xc, dc, x_plane, dy_plane, xcd_order);
    } else if (abl == 4) {
      allow_big_lds((const void*)wgrad_bf16x3_kernel<1, 0, 4>, &d4);
      hipLaunchKernelGGL((wgrad_bf16x3_kernel<1, 0, 4>), grid, dim3(256), lds, s, a, xc, dc, x_plane, dy_plane, xcd_order);
    } else
#endif
    {
      (void)abl;
      // eight waves of 32 x 32 x 5 taps, two per SIMD (four waves of 64 x 32: 229-239 against 248 TFLOP/s; removed in round 5)
      static unsigned long long d8 = 0;
      allow_big_lds((const void*)wgrad_bf16x3_kernel<1, 0, 0, 1>, &d8);
      hipLaunchKernelGGL((wgrad_bf16x3_kernel<1, 0, 0, 1>), grid, dim3(512), lds, s, a, xc, dc, x_plane, dy_plane, xcd_order);
    }
  } else {
    constexpr size_t lds = 2 * (size_t)(3 * 2 * 128 * 80 + 3 * 64 * 64);
    static unsigned long long f0 = 0, f1 = 0;
    if (wide_n) {
      // 64 ci x 128 co blocks
      constexpr size_t lds2 = 2 * (size_t)(3 * 2 * 64 * 80 + 3 * 128 * 64);
      dim3 grid2(a.Cin / 64, a.Cout / 128, splits);
      static unsigned long long g0 = 0, g1 = 0;
      if (pl & 1) {
        allow_big_lds((const void*)wgrad_bf16x3_kernel<2, 1, 0, 1, 4>, &g1);
        hipLaunchKernelGGL((wgrad_bf16x3_kernel<2, 1, 0, 1, 4>), grid2, dim3(512), lds2, s, a, xc, dc, x_plane, dy_plane, xcd_order);
      } else {
        allow_big_lds((const void*)wgrad_bf16x3_kernel<2, 0, 0, 1, 4>, &g0);
        hipLaunchKernelGGL((wgrad_bf16x3_kernel<2, 0, 0, 1, 4>), grid2, dim3(512), lds2, s, a, xc, dc, x_plane, dy_plane, xcd_order);
      }
    } else if (pl & 1) {
      allow_big_lds((const void*)wgrad_bf16x3_kernel<2, 1, 0, 1>, &f1);
      hipLaunchKernelGGL((wgrad_bf16x3_kernel<2, 1, 0, 1>), grid, dim3(512), lds, s, a, xc, dc, x_plane, dy_plane, xcd_order);
    } else {
      allow_big_lds((const void*)wgrad_bf16x3_kernel<2, 0, 0, 1>, &f0);
      hipLaunchKernelGGL((wgrad_bf16x3_kernel<2, 0, 0, 1>), grid, dim3(512), lds, s, a, xc, dc, x_plane, dy_plane, xcd_order);
    }
  }
  prof_end(s, 2.0 * a.B * (double)a.M * 5 * a.Cin * a.Cout, 2);
  return check_launch("wgrad_bf16x3");
}

}  // namespace gn
