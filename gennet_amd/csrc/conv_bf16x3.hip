// EXPERIMENTAL, opt-in (never on the default path): fp32 convolution on the bf16 matrix cores by operand splitting.
//
// Every fp32 operand is written as a = hi + mid + lo with three bf16 pieces (8 significant bits each; hi = bf16(a),
// mid = bf16(a - hi), lo = bf16(a - hi - mid), the subtractions are exact in fp32), so a*b = sum of nine bf16 x bf16 products,
// each EXACT in fp32.  The six largest (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid) are accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16; the three dropped ones are <= 2^-24 |a||b| together -- the size of one fp32 rounding of the
// product itself.  Six MFMAs at 1/16 of the cycles-per-flop of v_mfma_f32_32x32x2_f32 = 2.67x fewer matrix-core cycles.
// Measured error against fp64 (scripts/bf16x3_check.py): equal to or below the exact-fp32 MFMA path's.
//
// Structure: the split is a separate elementwise pass that also BLOCKS the operands by 16-channel K chunk, in 16-byte granules
// of 8 consecutive channels (= one lane's MFMA fragment):
//     activations  planes[3][B][Cin/16][2][1 + L + 1][8]  (a zero guard row on either side of every run)
//     weights      planes[3][tap][Cin/16][2][Cout][8]
// so everything a block needs for one chunk is contiguous runs of granules: a wave's LDS-DMA instruction reads 1 KiB contiguous
// (with channels-last planes every lane touches its own 128-byte line and the texture path, not the matrix core, sets the pace:
// measured 139 -> 186 TFLOP/s-equivalent from this alone).  The convolution is then pure LDS-DMA + ds_read_b128 + MFMA:
//   * 128 x 64 output tile, 4 waves (2 x 2), each wave 64 x 32 (two 32 x 32 MFMA tiles); K chunk = 16 channels = one MFMA K;
//   * LDS image per stage: A [plane][k-half][row] and B [plane][tap][k-half][n] in 16-byte granules, so DMA writes and fragment
//     reads are lane-linear and bank-conflict free; rows outside [0, L) are redirected to the zero guard rows;
//   * THREE stages (3 x 48 KiB) at one block per CU: an LDS-DMA lands ~1.1 us after issue, longer than one chunk of MFMAs, so the
//     DMA of chunk c+2 is issued during chunk c and retired by a COUNTED s_waitcnt vmcnt(N) + raw s_barrier at the end of chunk
//     c+1 (every wave issues the same number of DMA instructions per chunk, which is what makes the count valid);
//   * the fragment reads of tap j+1 are interleaved between the MFMAs of tap j (sched_group_barrier), the chunk's barrier sits
//     in front of the last tap's MFMAs.
//
// Measured (MI355X, Conv1D(512 -> 1024, k5) on (64, 2048, 512), random data): 188-190 TFLOP/s fp32-equivalent for the
// convolution, 183 with the split pass, against 134 for the exact-fp32 MFMA kernel (1.4x; 1.2x on 256 -> 512).  Ablations on
// the same launch: without the fragment reads 237, without the DMA 224, with neither 266 -- the bare MFMA loop itself holds only
// ~64 % of the nominal 2.5 PFLOP/s because the chip clocks down under dense bf16 MFMA on random data, and what remains is LDS
// throughput: per fp32-equivalent flop this scheme moves 1.5x the LDS bytes of the fp32 kernel in 2.67x less matrix-core
// time, ~4x its LDS load (which was 31 %); the 160 KiB of LDS rules out the larger wave tiles that would cut it.
// Limits (round 1's 64 x 32-tile kernel described above; the 64 x 64-wave-tile kernels further down added the stride-2 forward and the 2- / 3-tap phases,
// wgrad_bf16x3.hip the weight gradient): an infinite input comes out as NaN (inf * 0 in a cross term).
#include <algorithm>
#include <type_traits>
#include <stdlib.h>
#include "common.h"
#include "conv_epilogue.h"

namespace gn {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__device__ inline void split3(float a, unsigned short& h, unsigned short& m, unsigned short& l) {
  const __bf16 bh = (__bf16)a;
  float r1 = a - (float)bh;
  if (!(fabsf(a) < INFINITY)) r1 = 0.f;      // inf / nan stay in the hi piece alone
  const __bf16 bm = (__bf16)r1;
  const float r2 = r1 - (float)bm;
  const __bf16 bl = (__bf16)r2;
  h = __builtin_bit_cast(unsigned short, bh);
  m = __builtin_bit_cast(unsigned short, bm);
  l = __builtin_bit_cast(unsigned short, bl);
}

__device__ inline uint4 pack8(const unsigned short* p) {
  uint4 r;
  r.x = p[0] | ((unsigned)p[1] << 16); r.y = p[2] | ((unsigned)p[3] << 16);
  r.z = p[4] | ((unsigned)p[5] << 16); r.w = p[6] | ((unsigned)p[7] << 16);
  return r;
}

// x (B, L, C) fp32 -> planes[3][B][C/16][2][L + 2][8] bf16 (row 0 and row L + 1 of every run are zeros).
// Block = 64 rows x 64 channels of one batch element.  Rows are read as 256-byte runs (all four loads of a thread in flight before the first is used),
// split, regrouped through LDS into the 16-byte granules of 8 channels and written as 1-KiB runs (64 consecutive rows of one channel group) per plane.
// (Until round 4 a thread read its own 32 bytes of a row and wrote its three granules: 64-byte pieces per wave and row, 3.9 TB/s.)
__global__ __launch_bounds__(256) void split_x_kernel(const float* __restrict__ x, unsigned short* __restrict__ planes, int B, int L, int C) {
  constexpr int GSTRIDE = 64 * 16 + 64;                     // bytes per (plane, channel group) in LDS: 64 granules + a pad that staggers the groups' banks
  __shared__ __attribute__((aligned(16))) unsigned char T[3 * 8 * GSTRIDE];
  const int tid = threadIdx.x;
  const int t0 = blockIdx.x * 64, c0 = blockIdx.y * 64, b = blockIdx.z;
  const int c4 = tid & 15, r16 = tid >> 4;
  const float* sb = x + (size_t)b * L * C + min(c0 + 4 * c4, C - 4);      // (C % 16 == 0: the last channel tile may be partial; its surplus granules are not written)
  float4 v[4];
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) v[pass] = *reinterpret_cast<const float4*>(sb + (size_t)min(t0 + pass * 16 + r16, L - 1) * C);      // clamped, masked below
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int row = pass * 16 + r16;
    const float e[4] = {v[pass].x, v[pass].y, v[pass].z, v[pass].w};
    unsigned short h[4], m[4], l[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) split3(e[k], h[k], m[k], l[k]);
    unsigned char* q = T + (c4 >> 1) * GSTRIDE + row * 16 + (c4 & 1) * 8;
    *reinterpret_cast<uint2*>(q) = make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));
    *reinterpret_cast<uint2*>(q + 8 * GSTRIDE) = make_uint2(m[0] | ((unsigned)m[1] << 16), m[2] | ((unsigned)m[3] << 16));
    *reinterpret_cast<uint2*>(q + 16 * GSTRIDE) = make_uint2(l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16));
  }
  __syncthreads();
  const size_t plane = (size_t)B * C * (L + 2);
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int idx = tid + k * 256;
    const int row = idx & 63, g = (idx >> 6) & 7, pl = idx >> 9;
    const int t = t0 + row;
    if (t < L && c0 + 8 * g < C) {
      const uint4 val = *reinterpret_cast<const uint4*>(T + (pl * 8 + g) * GSTRIDE + row * 16);
      const size_t run = ((size_t)b * (C >> 3) + (c0 >> 3) + g) * (L + 2);
      unsigned short* o = planes + pl * plane + (run + 1 + t) * 8;
      *reinterpret_cast<uint4*>(o) = val;
      const uint4 z = make_uint4(0, 0, 0, 0);
      if (t == 0) *reinterpret_cast<uint4*>(planes + pl * plane + run * 8) = z;
      if (t == L - 1) *reinterpret_cast<uint4*>(planes + pl * plane + (run + L + 1) * 8) = z;
    }
  }
}

// W[tap][K][N] fp32 (the layout the fp32 conv kernels take) -> planes[3][tap][K/16][2][N][8] bf16; thread = (tap, 8-k group, n)
__global__ __launch_bounds__(256) void split_w_kernel(const float* __restrict__ w, unsigned short* __restrict__ planes, int taps, int K, int N) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int K8 = K >> 3;
  const size_t total = (size_t)taps * K8 * N;
  if (i >= total) return;
  const int n = (int)(i % N);
  const int k8 = (int)((i / N) % K8);
  const int tap = (int)(i / ((size_t)N * K8));
  unsigned short h[8], m[8], l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) split3(w[((size_t)tap * K + 8 * k8 + e) * N + n], h[e], m[e], l[e]);
  const size_t plane = (size_t)taps * K * N;
  const size_t o = (((size_t)tap * K8 + k8) * N + n) * 8;
  *reinterpret_cast<uint4*>(planes + o) = pack8(h);
  *reinterpret_cast<uint4*>(planes + plane + o) = pack8(m);
  *reinterpret_cast<uint4*>(planes + 2 * plane + o) = pack8(l);
}

#define GN_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)      /* lgkmcnt(0), vmcnt / expcnt untouched */

template <int NTAPS, int ASEG>   // ASEG = 64-row DMA segments of one (plane, k-half) region: 3 for stride 1, 5 for stride 2
__global__ __launch_bounds__(256, 1) void conv_bf16x3_kernel(ConvArgs a, const unsigned short* __restrict__ xs, const unsigned short* __restrict__ ws,
                                                             size_t x_plane, size_t w_plane, int m_tiles, int n_tiles) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int TM = 128, TN = 64;
  constexpr int A_BYTES = 6 * ASEG * 1024, B_BYTES = 6 * NTAPS * 1024, STAGE = A_BYTES + B_BYTES;
  constexpr int Q_TOTAL = 6 * ASEG + 6 * NTAPS;            // wave-level DMA instructions per chunk
  constexpr int Q_WAVE = (Q_TOTAL + 3) / 4;                // ... per wave (the same for every wave: counted vmcnt)
  constexpr int VMCNT_Q = 0x0F70 | (Q_WAVE & 15) | ((Q_WAVE >> 4) << 14);          // s_waitcnt vmcnt(Q_WAVE), lgkmcnt / expcnt untouched
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
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
  const int Rtot = is * Rper;
  const int n_chunks = a.Cin >> 4;
  const int Lg = a.Lin + 2;                                 // rows of one guarded run
  const int t_base = is * m0 + minoff;

  f32x16 acc[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

  // DMA instruction table of this wave: instruction q = wave + 4 i.  q < 6 ASEG: A region (plane, half), 64-row segment;
  // else B region (plane, tap, half) = 64 columns.  Source pointers advance by a constant per chunk.
  const unsigned short* src[Q_WAVE];
  int dst[Q_WAVE];          // LDS byte offset inside a stage
  size_t step[Q_WAVE];      // elements per chunk
#pragma unroll
  for (int i = 0; i < Q_WAVE; ++i) {
    int q = wave + 4 * i;
    if (q >= Q_TOTAL) q = Q_TOTAL - 1;                      // padding instruction: repeats the last one (same bytes, same place)
    if (q < 6 * ASEG) {
      const int ph = q / ASEG, seg = q % ASEG;
      const int lr = seg * 64 + lane;
      const int r = (is == 1) ? lr : (lr < Rper ? 2 * lr : 2 * (lr - Rper) + 1);
      const int t = t_base + r;
      const int row = (lr < Rtot && r < R && t >= 0 && t < a.Lin) ? t + 1 : 0;          // 0 = the leading zero guard row
      src[i] = xs + (ph >> 1) * x_plane + ((((size_t)b * n_chunks) * 2 + (ph & 1)) * Lg + row) * 8;
      dst[i] = q * 1024;
      step[i] = (size_t)2 * Lg * 8;
    } else {
      const int qb = q - 6 * ASEG;
      const int pt = qb >> 1, hh = qb & 1;
      const int p = pt / NTAPS, tap = pt % NTAPS;
      src[i] = ws + p * w_plane + ((((size_t)a.t.widx[tap] * n_chunks) * 2 + hh) * a.Cout + n0 + lane) * 8;
      dst[i] = A_BYTES + qb * 1024;
      step[i] = (size_t)2 * a.Cout * 8;
    }
  }
  auto dma_one = [&](int i, int c, unsigned char* stage) {
    gn_global_load_lds((gptr_t)(src[i] + (size_t)c * step[i]), (lptr_t)(stage + dst[i]), 16, 0, 0);
  };

  auto read_tap = [&](const unsigned char* sa, int j, bf16x8 (&av)[3][2], bf16x8 (&bv)[3]) {
    const int d = a.t.off[j] - minoff;
    const int rowbase = (is == 1) ? d : ((d & 1) * Rper + (d >> 1));
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        av[p][mt] = *reinterpret_cast<const bf16x8*>(sa + ((p * 2 + h) * (ASEG * 64) + rowbase + wm * 64 + mt * 32 + i32) * 16);
      bv[p] = *reinterpret_cast<const bf16x8*>(sa + A_BYTES + (((p * NTAPS + j) * 2 + h) * 64 + wn * 32 + i32) * 16);
    }
  };
  auto mma_tap = [&](const bf16x8 (&av)[3][2], const bf16x8 (&bv)[3]) {
    // smallest terms first: hi*lo, lo*hi, mid*mid, then hi*mid, mid*hi, then hi*hi
    constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[PA[q]][mt], bv[PB[q]], acc[mt], 0, 0, 0);
  };

  // ---- prologue: chunks 0 and 1 in flight, chunk 0 retired
#pragma unroll
  for (int i = 0; i < Q_WAVE; ++i) dma_one(i, 0, smem_b);
#pragma unroll
  for (int i = 0; i < Q_WAVE; ++i) dma_one(i, min(1, n_chunks - 1), smem_b + STAGE);
  __builtin_amdgcn_s_waitcnt(VMCNT_Q);                      // chunk 0 has landed (chunk 1 may still fly)
  asm volatile("s_barrier" ::: "memory");       // (not the builtin: it is IntrNoMem, LDS loads may move across it -- ADVICE r4)

  // fragment buffers: tap 0 -> [2], tap j >= 1 -> [(j - 1) & 1]
  bf16x8 fa[3][3][2], fb[3][3];
  read_tap(smem_b, 0, fa[2], fb[2]);
  int s_cur = 0;                                            // stage of chunk ch; chunk ch+2 goes to (s_cur + 2) % 3
  constexpr int DMA_TAPS = NTAPS > 1 ? NTAPS - 1 : 1;       // taps of a chunk that carry DMA issue (all but the last)
  constexpr int DMA_PER_TAP = (Q_WAVE + DMA_TAPS - 1) / DMA_TAPS;
  for (int ch = 0; ch < n_chunks; ++ch) {
    const unsigned char* sa = smem_b + s_cur * STAGE;
    const int s_nxt = s_cur == 2 ? 0 : s_cur + 1;
    unsigned char* s_dma = smem_b + (s_nxt == 2 ? 0 : s_nxt + 1) * STAGE;
    const int c_dma = min(ch + 2, n_chunks - 1);            // past the end: reloads the last chunk into a free stage (keeps the count)
#pragma unroll
    for (int j = 0; j < NTAPS; ++j) {
      const int cur = j == 0 ? 2 : ((j - 1) & 1);
      GN_WAIT_LGKM0();                                      // this tap's fragments are in registers
      __builtin_amdgcn_sched_barrier(0);
      if (j + 1 < NTAPS) {
#pragma unroll
        for (int i = j * DMA_PER_TAP; i < (j + 1) * DMA_PER_TAP && i < Q_WAVE; ++i) dma_one(i, c_dma, s_dma);
        read_tap(sa, j + 1, fa[j & 1], fb[j & 1]);
      } else {
        if (NTAPS == 1) {
#pragma unroll
          for (int i = 0; i < Q_WAVE; ++i) dma_one(i, c_dma, s_dma);
        }
        // every read of this stage has landed; chunk ch+1 must have landed: at most this chunk's own Q_WAVE DMAs stay in flight
        __builtin_amdgcn_s_waitcnt(VMCNT_Q);
        asm volatile("s_barrier" ::: "memory");       // (not the builtin: it is IntrNoMem, LDS loads may move across it -- ADVICE r4)
        if (ch + 1 < n_chunks) read_tap(smem_b + s_nxt * STAGE, 0, fa[2], fb[2]);
      }
      mma_tap(fa[cur], fb[cur]);
      // 12 MFMAs with the 9 fragment reads of the next tap (and this tap's DMA issue) between them
#pragma unroll
      // front-loaded: the reads are out after 5 MFMAs, so 7 more MFMAs (~220 cycles) cover their latency before the next wait
      for (int k = 0; k < 4; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        if (k < DMA_PER_TAP) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 7, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    s_cur = s_nxt;
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0): drain the padding DMAs before the block may end

  float* yb = a.y + (size_t)b * a.Ly * a.Cout;
  const uint8_t* mb = a.mask ? a.mask + (size_t)b * a.Ly * a.Cout : nullptr;
  const float* gyb = a.gy ? a.gy + (size_t)b * a.Ly * a.Cout : nullptr;
  const uint8_t* gmb = a.gmask ? a.gmask + (size_t)b * a.Ly * a.Cout : nullptr;
  const int n = n0 + wn * 32 + i32;
  const float bias = a.bias ? a.bias[n] : 0.f;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
      const int m = m0 + wm * 64 + mt * 32 + row;
      if (m < a.M) {
        const size_t o = (size_t)(a.t.out_stride * m + a.t.out_off) * a.Cout + n;
        float v = act_apply(acc[mt][r] + bias, a.act, a.act_param);
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
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Round 4: the same arithmetic on 64 x 64 WAVE tiles (256 x 64 block, four waves stacked in M, unit input stride, 4-5 taps).
// The kernel above is LDS-bound: per tap a 64 x 32 wave issues 9 fragment reads (ds_read_b128) for 12 MFMAs -- with the staging
// writes ~98 % of the CU's 128 B/cycle.  A 64 x 64 wave tile issues 12 reads for 24 MFMAs (A: 3 planes x 2 row tiles, B: 3 planes x
// 2 column tiles): half the LDS bytes per MFMA, ~65 % with the staging.  One chunk is then 5 x 24 MFMAs = 3840 matrix cycles
// (1.6 us), longer than the ~1.1 us an LDS-DMA takes to land, so TWO stages of 60 KiB suffice: chunk c+2 is issued right behind the
// barrier that frees chunk c's stage (in front of chunk c's last tap) and retired by vmcnt(0) at the next barrier.
// Block -> tile: XCD k (block i -> XCD i mod 8) takes a contiguous eighth of the (slab, column tile) list, so an input slab -- 6 bytes
// per element here -- is fetched through one L2 and serves all its column tiles from there.
// Ceiling (scripts/mfma_bf16_peak.hip): the bare six-product stream sustains 307-313 TFLOP/s fp32-equivalent on live data.
// ---------------------------------------------------------------------------------------------------------------------------------
// ABL (timing experiments only, GN_BF16X3_ABL; results are wrong for ABL != 0): bit 0 = no fragment reads in the loop, bit 1 = no staging in the
// loop, bit 2 = no epilogue
// IS = input stride (1, or 2 for the stride-2 forward: the slab is staged de-interleaved, even input rows then odd ones, so that every tap reads a
// unit-stride run -- 258 + 258 rows per (plane, k-half) region, 80 256 bytes per stage: two stages just fit the 160 KiB).
// MERGE (1, 2): BOTH output phases of a stride-2 data gradient in one launch -- five taps in kernel-tap order, even index -> accumulator set 0 (rows
// out_stride * m + out_off), odd index -> set 1 (out_off_odd), offsets spanning three rows.  Two pairs of neighbouring taps read the SAME input rows
// (pad_left odd, MERGE 1: taps (1, 2) and (3, 4); even, MERGE 2: (0, 1) and (2, 3)): their x fragments are read once -- 18 + 30 fragment reads per
// chunk instead of the 30 + 30 of five distinct taps, in a kernel that is bound by exactly those reads -- and the slab is staged once, not per phase.
template <int MERGE> __host__ __device__ constexpr bool tap_shares_a(int j) { return MERGE == 1 ? (j == 2 || j == 4) : MERGE == 2 ? (j == 1 || j == 3) : false; }
template <int MERGE> __host__ __device__ constexpr int a_reads_upto(int j) {       // x-fragment reads among taps 1 .. j
  int n = 0;
  for (int t = 1; t <= j; ++t) n += tap_shares_a<MERGE>(t) ? 0 : 1;
  return n;
}
template <int NTAPS, int ABL = 0, int IS = 1, int MERGE = 0>
__global__ __launch_bounds__(256, 1) void conv_bf16x3_wide_kernel(ConvArgs a, const unsigned short* __restrict__ xs, const unsigned short* __restrict__ ws,
                                                                  size_t x_plane, size_t w_plane, int m_tiles, int n_tiles) {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(NTAPS >= 4 && NTAPS <= 5, "a chunk must outlast the staging latency: 4 or 5 taps");
  static_assert(IS == 1 || (IS == 2 && NTAPS == 5), "stride 2: the 5-tap forward only");
  static_assert(MERGE == 0 || (NTAPS == 5 && IS == 1), "merged phases: five taps, unit input stride");
  constexpr int NSETS = MERGE ? 2 : 1;
  constexpr int TM = 256, TN = 64;
  constexpr int RPER = IS == 1 ? 320 : 258;                // rows of one parity class of a region (stride 1: one class, padded to whole 64-row DMA segments)
  constexpr int RTOT = IS * RPER;                          // LDS rows of one (plane, k-half) region
  constexpr int ASEG = (RTOT + 63) / 64;                   // 64-row DMA segments per region (the last one partial at stride 2: lanes past the region are masked)
  constexpr int REGION = RTOT * 16;
  constexpr int A_BYTES = 6 * REGION, B_BYTES = 6 * NTAPS * 1024, STAGE = A_BYTES + B_BYTES;
  static_assert(2 * STAGE <= 160 * 1024, "two stages must fit the CU's LDS");
  constexpr int Q_TOTAL = 6 * ASEG + 6 * NTAPS;            // wave-level DMA instructions per chunk
  constexpr int Q_WAVE = (Q_TOTAL + 3) / 4;                // ... per wave
  constexpr int Q_FIRST = (Q_WAVE + 1) / 2;                // issued behind the barrier (last tap of chunk c); the rest in tap 0 of chunk c + 1
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i32 = lane & 31, h = lane >> 5;
  int lin = blockIdx.x;
  const int nb = gridDim.x;
  if (lin < (nb & ~7)) lin = (lin & 7) * (nb >> 3) + (lin >> 3);
  const int n_tile = lin % n_tiles;
  const int rest = lin / n_tiles;
  const int m_tile = rest % m_tiles;
  const int b = rest / m_tiles;
  const int m0 = m_tile * TM, n0 = n_tile * TN;

  int minoff = a.t.off[0], maxoff = a.t.off[0];
#pragma unroll
  for (int j = 1; j < NTAPS; ++j) {
    minoff = min(minoff, a.t.off[j]);
    maxoff = max(maxoff, a.t.off[j]);
  }
  const int R = IS * (TM - 1) + (maxoff - minoff) + 1;     // staged input rows (the launcher checks the taps span <= 5 rows)
  const int n_chunks = a.Cin >> 4;
  const int Lg = a.Lin + 2;
  const int t_base = IS * m0 + minoff;

  f32x16 acc[NSETS][2][2];
#pragma unroll
  for (int st = 0; st < NSETS; ++st)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[st][mt][nt][r] = 0.f;

  // Staging plan.  Wave wm stages the slab segments wm, wm + 4, ... (64 rows of all six (plane, k-half) regions each); the LAST segment of the regions
  // (ASEG = 4 NREG + 1) is shared out by region: wave wm takes regions wm and wm + 4; a share of the 6 * NTAPS weight pieces evens the count out to
  // Q_WAVE instructions per wave (slab rows are gathers, weight pieces contiguous: the slab work has to be even, the barrier waits for the slowest wave).
  // What differs between the lanes of one instruction is then only the row offset of the wave's NREG + 1 segments or the column (one register each);
  // everything else is wave-uniform and lives in scalar registers -- the per-instruction pointer / stride / destination arrays this replaces took ~6
  // registers per instruction.
  static_assert((ASEG - 1) % 4 == 0, "segments per region: 4 NREG + 1");
  constexpr int NREG = (ASEG - 1) / 4;
  constexpr int NB_PURE = Q_WAVE - 6 * NREG - 2;           // instructions that are weight pieces on every wave
  constexpr int QB = 6 * NTAPS;
  constexpr bool MASKED = (RTOT % 64) != 0;                // the last segment of a region is partial: lanes past it must not write into the next region
  static_assert(NB_PURE >= 0, "a wave's slab share must fit its instruction count");
  const bool x1_inv = wm + 4 >= 6;                         // waves 2, 3 have no second region of the last segment: that instruction carries a weight piece
  int b_first = 0;                                         // first weight piece of this wave
#pragma unroll
  for (int w2 = 0; w2 < 3; ++w2)
    if (w2 < wm) b_first += NB_PURE + (w2 + 4 >= 6 ? 1 : 0);
  auto row_offset = [&](int seg) {                         // byte offset of the lane's row inside a (plane, k-half) run; -1: the lane is past the region
    const int lr = seg * 64 + lane;                                                          // LDS row inside the region
    const int r = (IS == 1) ? lr : (lr < RPER ? 2 * lr : 2 * (lr - RPER) + 1);               // input row it holds (stride 2: even rows, then odd rows)
    const int t = t_base + r;
    const int row = (r < R && t >= 0 && t < a.Lin) ? t + 1 : 0;                             // 0 = the leading zero guard row
    return (MASKED && lr >= RTOT) ? -1 : row * 16;
  };
  int rowoff[NREG + 1];
#pragma unroll
  for (int sl = 0; sl < NREG; ++sl) rowoff[sl] = row_offset(wm + 4 * sl);
  rowoff[NREG] = row_offset(ASEG - 1);
  const int coloff = lane * 16;
  auto a_run = [&](int ph) { return ((size_t)(ph >> 1) * x_plane + (((size_t)b * n_chunks) * 2 + (ph & 1)) * Lg * 8) * 2; };      // (plane, k-half) run of chunk 0, bytes
  size_t a_base[6];
#pragma unroll
  for (int ph = 0; ph < 6; ++ph) a_base[ph] = a_run(ph);
  const size_t a_step = (size_t)2 * Lg * 8 * 2, b_step = (size_t)2 * a.Cout * 8 * 2;      // bytes per chunk
  auto piece = [&](int ord, size_t& off, int& dst_) {      // weight piece number b_first + ord (clamped: the padding instructions repeat the last piece)
    const int qb = min(b_first + ord, QB - 1);
    const int pt = qb >> 1, hh = qb & 1;
    const int pp = pt / NTAPS, tap = pt % NTAPS;
    int wi = a.t.widx[0];
#pragma unroll
    for (int t2 = 1; t2 < NTAPS; ++t2) wi = (tap == t2) ? a.t.widx[t2] : wi;                // (no indexed load of the argument block)
    off = ((size_t)pp * w_plane + (((size_t)wi * n_chunks) * 2 + hh) * a.Cout * 8 + (size_t)n0 * 8) * 2;
    dst_ = A_BYTES + qb * 1024;
  };
  // the two instructions of the last segment: regions wm and wm + 4 (the second one a weight piece on waves 2, 3)
  const size_t x0_off = a_run(wm);
  const int x0_dst = wm * REGION + (ASEG - 1) * 1024;
  size_t x1_off;
  int x1_dst;
  piece(0, x1_off, x1_dst);
  if (!x1_inv) { x1_off = a_run(wm + 4); x1_dst = (wm + 4) * REGION + (ASEG - 1) * 1024; }
  size_t pure_off[NB_PURE > 0 ? NB_PURE : 1];
  int pure_dst[NB_PURE > 0 ? NB_PURE : 1];
#pragma unroll
  for (int k = 0; k < NB_PURE; ++k) piece((x1_inv ? 1 : 0) + k, pure_off[k], pure_dst[k]);
  bool in_loop = false;
  auto dma_one = [&](int i, int c, unsigned char* stage) {
    if ((ABL & 2) && in_loop) return;
    const char* base;
    int voff, d;
    if (i < 6 * NREG) {
      const int sl = i / 6, ph = i % 6;
      base = (const char*)xs + a_base[ph] + (size_t)c * a_step;
      voff = rowoff[sl];
      d = ph * REGION + (wm + 4 * sl) * 1024;
    } else if (i == 6 * NREG) {
      base = (const char*)xs + x0_off + (size_t)c * a_step;
      voff = rowoff[NREG];
      d = x0_dst;
    } else if (i == 6 * NREG + 1) {
      base = (x1_inv ? (const char*)ws : (const char*)xs) + x1_off + (size_t)c * (x1_inv ? b_step : a_step);
      voff = x1_inv ? coloff : rowoff[NREG];
      d = x1_dst;
    } else {
      base = (const char*)ws + pure_off[i - 6 * NREG - 2] + (size_t)c * b_step;
      voff = coloff;
      d = pure_dst[i - 6 * NREG - 2];
    }
    // only the two instructions of the (partial) last segment can have lanes past the region
    if (!MASKED || (i != 6 * NREG && i != 6 * NREG + 1) || voff >= 0) gn_global_load_lds((gptr_t)(base + voff), (lptr_t)(stage + d), 16, 0, 0);
  };
  auto read_tap = [&](const unsigned char* sa, int j, bf16x8 (&av)[3][2], bf16x8 (&bv)[3][2], bool with_a = true) {
    if ((ABL & 1) && in_loop) return;
    const int d = a.t.off[j] - minoff;
    const int rowbase = (IS == 1) ? d : ((d & 1) * RPER + (d >> 1));
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      if (with_a) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          av[p][mt] = *reinterpret_cast<const bf16x8*>(sa + ((p * 2 + h) * RTOT + rowbase + wm * 64 + mt * 32 + i32) * 16);
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        bv[p][nt] = *reinterpret_cast<const bf16x8*>(sa + A_BYTES + (((p * NTAPS + j) * 2 + h) * 64 + nt * 32 + i32) * 16);
    }
  };
  auto mma_tap = [&](const bf16x8 (&av)[3][2], const bf16x8 (&bv)[3][2], int st) {
    // smallest terms first: hi*lo, lo*hi, mid*mid, then hi*mid, mid*hi, then hi*hi
    constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[st][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[PA[q]][mt], bv[PB[q]][nt], acc[st][mt][nt], 0, 0, 0);
  };

  // ---- prologue: chunks 0 and 1 staged and landed
#pragma unroll
  for (int i = 0; i < Q_WAVE; ++i) dma_one(i, 0, smem_b);
#pragma unroll
  for (int i = 0; i < Q_WAVE; ++i) dma_one(i, min(1, n_chunks - 1), smem_b + STAGE);
  __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0)
  asm volatile("s_barrier" ::: "memory");       // (not the builtin: it is IntrNoMem, LDS loads may move across it -- ADVICE r4)

  bf16x8 fa[2][3][2], fb[2][3][2];                          // fragment double buffer: the tap with running parity P reads from [P], prefetches into [P ^ 1]
  read_tap(smem_b, 0, fa[0], fb[0]);
  // one chunk; P0 = buffer parity of its tap 0 (compile time: with five taps per chunk the parity alternates from chunk to chunk)
  auto do_chunk = [&](int ch, auto p0) {
    constexpr int P0 = decltype(p0)::value;
    unsigned char* sa = smem_b + (ch & 1) * STAGE;
    unsigned char* sb = smem_b + ((ch + 1) & 1) * STAGE;
    const int c_dma1 = min(ch + 1, n_chunks - 1);           // second half of chunk ch+1's staging (into sb; the first half went out in chunk ch-1's last tap)
    const int c_dma2 = min(ch + 2, n_chunks - 1);           // first half of chunk ch+2's staging (into sa, once the barrier below has freed it)
#pragma unroll
    for (int j = 0; j < NTAPS; ++j) {
      constexpr int dummy = 0; (void)dummy;
      const int P = (P0 + j) & 1;                           // weight fragments: one set per tap
      const int PA_ = (P0 + a_reads_upto<MERGE>(j)) & 1;    // x fragments: a tap that shares its rows with the tap before keeps that tap's set
      const int PAn = (P0 + a_reads_upto<MERGE>(j) + 1) & 1;
      const bool nxt_a = j + 1 < NTAPS ? !tap_shares_a<MERGE>(j + 1) : true;
      __builtin_amdgcn_s_waitcnt(0xC07F);                   // lgkmcnt(0): this tap's fragments are in registers
      __builtin_amdgcn_sched_barrier(0);
      int ndma = 0;
      if (j + 1 < NTAPS) {
        if (j == 0 && ch > 0) {
#pragma unroll
          for (int i = Q_FIRST; i < Q_WAVE; ++i) dma_one(i, c_dma1, sb);
          ndma = Q_WAVE - Q_FIRST;
        }
        read_tap(sa, j + 1, fa[nxt_a ? PAn : PA_], fb[P ^ 1], nxt_a);
      } else {
        // every read of stage sa has landed (lgkmcnt(0) above, on every wave once past the barrier); chunk ch+1 must have landed
        __builtin_amdgcn_s_waitcnt(0x0F70);
        asm volatile("s_barrier" ::: "memory");       // (not the builtin: it is IntrNoMem, LDS loads may move across it -- ADVICE r4)
#pragma unroll
        for (int i = 0; i < Q_FIRST; ++i) dma_one(i, c_dma2, sa);      // (past the end: one harmless re-stage of the last chunk keeps the count uniform)
        ndma = Q_FIRST;
        if (ch + 1 < n_chunks) read_tap(sb, 0, fa[PAn], fb[P ^ 1]);
      }
      mma_tap(fa[PA_], fb[P], MERGE ? (j & 1) : 0);
      // 24 MFMAs with the 12 fragment reads of the next tap (and this tap's share of the staging) between them
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (k < ndma) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  in_loop = true;
  if constexpr (NTAPS % 2 == 0) {
    for (int ch = 0; ch < n_chunks; ++ch) do_chunk(ch, std::integral_constant<int, 0>{});
  } else {
    for (int ch = 0; ch < n_chunks; ch += 2) {
      do_chunk(ch, std::integral_constant<int, 0>{});
      if (ch + 1 < n_chunks) do_chunk(ch + 1, std::integral_constant<int, 1>{});
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0): drain the trailing DMAs before the block may end

  if constexpr ((ABL & 4) != 0) {
    if (acc[0][0][0][0] == 12345.678f) a.y[0] = acc[0][0][0][0] + acc[0][1][1][15] + acc[0][0][1][3] + acc[0][1][0][7];      // keeps the accumulators alive
    return;
  }
  // the lean epilogue of the hand-scheduled fp32 kernel (conv_epilogue.h): activation / fused variants decided once per wave, buffer stores with a
  // scalar row offset, rows past M dropped by the descriptor's range check -- the generic per-element epilogue cost 10 % of this kernel's time at
  // one block per CU (2.92 against 3.25 ms on G 512 -> 1024 at batch 64 without it)
  const int mode = a.gy ? (a.gmask ? 3 : 2) : (a.mask ? 1 : 0);
  pipe_epilogue_dispatch<2>(a, acc[0], b, m0 + wm * 64, n0, i32, h, a.t.out_off, mode);
  if constexpr (MERGE != 0) pipe_epilogue_dispatch<2>(a, acc[1], b, m0 + wm * 64, n0, i32, h, a.t.out_off_odd, mode);
#endif
}

// The same 64 x 64 wave tiles for the 2- / 3-tap launches (the two output phases of a stride-2 data gradient).  A chunk is then 2 or 3 taps x 24
// MFMAs = 0.64 / 0.96 us, shorter than an LDS-DMA takes to land, so chunk c+2 is staged during chunk c -- THREE stages of 42 / 48 KiB -- and retired
// by a counted vmcnt(Q_WAVE) at the barrier of chunk c+1 (every wave issues the same number of DMA instructions per chunk: the count is valid).
template <int NTAPS>
__global__ __launch_bounds__(256, 1) void conv_bf16x3_wide3_kernel(ConvArgs a, const unsigned short* __restrict__ xs, const unsigned short* __restrict__ ws,
                                                                  size_t x_plane, size_t w_plane, int m_tiles, int n_tiles) {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(NTAPS >= 2 && NTAPS <= 3, "the short-tap launches");
  constexpr int IS = 1, ABL = 0;
  constexpr int TM = 256, TN = 64;
  constexpr int RPER = IS == 1 ? 320 : 258;                // rows of one parity class of a region (stride 1: one class, padded to whole 64-row DMA segments)
  constexpr int RTOT = IS * RPER;                          // LDS rows of one (plane, k-half) region
  constexpr int ASEG = (RTOT + 63) / 64;                   // 64-row DMA segments per region (the last one partial at stride 2: lanes past the region are masked)
  constexpr int REGION = RTOT * 16;
  constexpr int A_BYTES = 6 * REGION, B_BYTES = 6 * NTAPS * 1024, STAGE = A_BYTES + B_BYTES;
  static_assert(3 * STAGE <= 160 * 1024, "three stages must fit the CU's LDS");
  constexpr int Q_TOTAL = 6 * ASEG + 6 * NTAPS;            // wave-level DMA instructions per chunk
  constexpr int Q_WAVE = (Q_TOTAL + 3) / 4;                // ... per wave
  constexpr int VMCNT_Q = 0x0F70 | (Q_WAVE & 15) | ((Q_WAVE >> 4) << 14);          // s_waitcnt vmcnt(Q_WAVE), lgkmcnt / expcnt untouched
  constexpr int DMA_PER_TAP = (Q_WAVE + NTAPS - 2) / (NTAPS - 1);                   // chunk c+2 goes out over taps 0 .. NTAPS-2 of chunk c
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i32 = lane & 31, h = lane >> 5;
  int lin = blockIdx.x;
  const int nb = gridDim.x;
  if (lin < (nb & ~7)) lin = (lin & 7) * (nb >> 3) + (lin >> 3);
  const int n_tile = lin % n_tiles;
  const int rest = lin / n_tiles;
  const int m_tile = rest % m_tiles;
  const int b = rest / m_tiles;
  const int m0 = m_tile * TM, n0 = n_tile * TN;

  int minoff = a.t.off[0], maxoff = a.t.off[0];
#pragma unroll
  for (int j = 1; j < NTAPS; ++j) {
    minoff = min(minoff, a.t.off[j]);
    maxoff = max(maxoff, a.t.off[j]);
  }
  const int R = IS * (TM - 1) + (maxoff - minoff) + 1;     // staged input rows (the launcher checks the taps span <= 5 rows)
  const int n_chunks = a.Cin >> 4;
  const int Lg = a.Lin + 2;
  const int t_base = IS * m0 + minoff;

  f32x16 acc[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  const unsigned short* src[Q_WAVE];
  int dst[Q_WAVE];
  bool live[Q_WAVE];
  size_t step[Q_WAVE];
#pragma unroll
  for (int i = 0; i < Q_WAVE; ++i) {
    int q = wm + 4 * i;
    if (q >= Q_TOTAL) q = Q_TOTAL - 1;                      // padding instruction: repeats the last one (same bytes, same place)
    if (q < 6 * ASEG) {
      const int ph = q / ASEG, seg = q % ASEG;
      const int lr = seg * 64 + lane;                                                        // LDS row inside the region
      const int r = (IS == 1) ? lr : (lr < RPER ? 2 * lr : 2 * (lr - RPER) + 1);             // input row it holds (stride 2: even rows, then odd rows)
      const int t = t_base + r;
      const int row = (r < R && t >= 0 && t < a.Lin) ? t + 1 : 0;                           // 0 = the leading zero guard row
      src[i] = xs + (ph >> 1) * x_plane + ((((size_t)b * n_chunks) * 2 + (ph & 1)) * Lg + row) * 8;
      dst[i] = ph * REGION + seg * 1024;
      live[i] = lr < RTOT;                                                                   // the partial last segment must not write into the next region
      step[i] = (size_t)2 * Lg * 8;
    } else {
      const int qb = q - 6 * ASEG;
      const int pt = qb >> 1, hh = qb & 1;
      const int p = pt / NTAPS, tap = pt % NTAPS;
      src[i] = ws + p * w_plane + ((((size_t)a.t.widx[tap] * n_chunks) * 2 + hh) * a.Cout + n0 + lane) * 8;
      dst[i] = A_BYTES + qb * 1024;
      live[i] = true;
      step[i] = (size_t)2 * a.Cout * 8;
    }
  }
  bool in_loop = false;
  auto dma_one = [&](int i, int c, unsigned char* stage) {
    if ((ABL & 2) && in_loop) return;
    if (IS == 1 || live[i]) gn_global_load_lds((gptr_t)(src[i] + (size_t)c * step[i]), (lptr_t)(stage + dst[i]), 16, 0, 0);
  };
  auto read_tap = [&](const unsigned char* sa, int j, bf16x8 (&av)[3][2], bf16x8 (&bv)[3][2]) {
    if ((ABL & 1) && in_loop) return;
    const int d = a.t.off[j] - minoff;
    const int rowbase = (IS == 1) ? d : ((d & 1) * RPER + (d >> 1));
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        av[p][mt] = *reinterpret_cast<const bf16x8*>(sa + ((p * 2 + h) * RTOT + rowbase + wm * 64 + mt * 32 + i32) * 16);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        bv[p][nt] = *reinterpret_cast<const bf16x8*>(sa + A_BYTES + (((p * NTAPS + j) * 2 + h) * 64 + nt * 32 + i32) * 16);
    }
  };
  auto mma_tap = [&](const bf16x8 (&av)[3][2], const bf16x8 (&bv)[3][2]) {
    // smallest terms first: hi*lo, lo*hi, mid*mid, then hi*mid, mid*hi, then hi*hi
    constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[PA[q]][mt], bv[PB[q]][nt], acc[mt][nt], 0, 0, 0);
  };

  // ---- prologue: chunks 0 and 1 staged and landed
#pragma unroll
  for (int i = 0; i < Q_WAVE; ++i) dma_one(i, 0, smem_b);
#pragma unroll
  for (int i = 0; i < Q_WAVE; ++i) dma_one(i, min(1, n_chunks - 1), smem_b + STAGE);
  __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0)
  asm volatile("s_barrier" ::: "memory");       // (not the builtin: it is IntrNoMem, LDS loads may move across it -- ADVICE r4)

  bf16x8 fa[2][3][2], fb[2][3][2];                          // fragment double buffer: the tap with running parity P reads from [P], prefetches into [P ^ 1]
  read_tap(smem_b, 0, fa[0], fb[0]);
  in_loop = true;
  int s_cur = 0;                                            // stage of chunk ch; chunk ch+1 sits in s_cur + 1, chunk ch+2 goes to s_cur + 2 (mod 3)
  auto do_chunk = [&](int ch, auto p0) {
    constexpr int P0 = decltype(p0)::value;
    const int s_nxt = s_cur == 2 ? 0 : s_cur + 1;
    unsigned char* sa = smem_b + s_cur * STAGE;
    unsigned char* sb = smem_b + s_nxt * STAGE;
    unsigned char* s_dma = smem_b + (s_nxt == 2 ? 0 : s_nxt + 1) * STAGE;      // = the stage of chunk ch-1: every wave is past that chunk's barrier
    const int c_dma = min(ch + 2, n_chunks - 1);            // past the end: one harmless re-stage of the last chunk keeps the count uniform
#pragma unroll
    for (int j = 0; j < NTAPS; ++j) {
      const int P = (P0 + j) & 1;
      __builtin_amdgcn_s_waitcnt(0xC07F);                   // lgkmcnt(0): this tap's fragments are in registers
      __builtin_amdgcn_sched_barrier(0);
      int ndma = 0;
      if (j + 1 < NTAPS) {
#pragma unroll
        for (int i = j * DMA_PER_TAP; i < (j + 1) * DMA_PER_TAP && i < Q_WAVE; ++i) { dma_one(i, c_dma, s_dma); ++ndma; }
        read_tap(sa, j + 1, fa[P ^ 1], fb[P ^ 1]);
      } else {
        // every read of stage sa has landed; chunk ch+1 must have landed: at most this chunk's own Q_WAVE DMAs (chunk ch+2) stay in flight
        __builtin_amdgcn_s_waitcnt(VMCNT_Q);
        asm volatile("s_barrier" ::: "memory");       // (not the builtin: it is IntrNoMem, LDS loads may move across it -- ADVICE r4)
        if (ch + 1 < n_chunks) read_tap(sb, 0, fa[P ^ 1], fb[P ^ 1]);
      }
      mma_tap(fa[P], fb[P]);
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (k < ndma) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    s_cur = s_nxt;
  };
  if constexpr (NTAPS % 2 == 0) {
    for (int ch = 0; ch < n_chunks; ++ch) do_chunk(ch, std::integral_constant<int, 0>{});
  } else {
    for (int ch = 0; ch < n_chunks; ch += 2) {
      do_chunk(ch, std::integral_constant<int, 0>{});
      if (ch + 1 < n_chunks) do_chunk(ch + 1, std::integral_constant<int, 1>{});
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0): drain the trailing DMAs before the block may end

  if constexpr ((ABL & 4) != 0) {
    if (acc[0][0][0] == 12345.678f) a.y[0] = acc[0][0][0] + acc[1][1][15] + acc[0][1][3] + acc[1][0][7];      // keeps the accumulators alive
    return;
  }
  // the lean epilogue of the hand-scheduled fp32 kernel (conv_epilogue.h): activation / fused variants decided once per wave, buffer stores with a
  // scalar row offset, rows past M dropped by the descriptor's range check -- the generic per-element epilogue cost 10 % of this kernel's time at
  // one block per CU (2.92 against 3.25 ms on G 512 -> 1024 at batch 64 without it)
  const int mode = a.gy ? (a.gmask ? 3 : 2) : (a.mask ? 1 : 0);
  pipe_epilogue_dispatch<2>(a, acc, b, m0 + wm * 64, n0, i32, h, a.t.out_off, mode);
#endif
}

// Measured and NOT kept (round 4): a wave-specialised form -- a fifth "producer" wave issues all 60 LDS-DMA instructions of a chunk, the four MFMA
// waves only read fragments and multiply.  The ablation above reads as if the 12 % were the cost of issuing the staging from the MFMA waves; it is
// not: with the producer wave the kernel runs at the same 226-228 TFLOP/s, so what the staging costs is LDS bandwidth shared with the fragment
// reads (12 ds_read_b128 per 24 MFMAs and wave, the same B fragments read by all four waves), not instruction issue.  (One trap on the way: a raw
// __builtin_amdgcn_s_barrier() is IntrNoMem, and along a code path with no visible store to LDS -- the MFMA waves, whose stages only LDS-DMA of
// ANOTHER wave writes -- the optimiser treats LDS as invariant and reuses fragment loads from two chunks earlier: deterministic garbage in every
// second row tile until the barrier was written as asm volatile("s_barrier" ::: "memory").)

size_t conv_bf16x3_workspace_bytes(int B, int Lin, int Cin, int Cout, int w_taps) {
  return 6 * ((size_t)B * (Lin + 2) * Cin + (size_t)w_taps * Cin * Cout) + 256;
}

// stride-2 forward (round 4): the 5-tap launches whose rows fill the 256-row blocks of the wide kernel; everything else needs unit input stride
static bool bf16x3_stride2_ok(const ConvArgs& a) {
  if (a.t.in_stride != 2 || a.t.ntaps != 5 || a.M < 192 || a.t.out_stride != 1) return false;
  int minoff = a.t.off[0], maxoff = a.t.off[0];
  for (int j = 1; j < 5; ++j) {
    minoff = std::min(minoff, a.t.off[j]);
    maxoff = std::max(maxoff, a.t.off[j]);
  }
  return maxoff - minoff + 1 == 5;
}

bool conv_bf16x3_supported(const ConvArgs& a) {
  return a.Cin % 16 == 0 && a.Cout % 64 == 0 && a.t.ntaps >= 1 && a.t.ntaps <= 5 && (a.t.in_stride == 1 || bf16x3_stride2_ok(a));
}

// splits: x (B, Lin, Cin) and the conv-layout weights w (w_taps, Cin, Cout) -> blocked bf16 planes in `ws`
int conv_bf16x3_split(const ConvArgs& a, int w_taps, void* ws, size_t ws_bytes, bool split_x, bool split_w, hipStream_t s) {
  if (ws_bytes < conv_bf16x3_workspace_bytes(a.B, a.Lin, a.Cin, a.Cout, w_taps)) {
    set_error("conv_bf16x3: workspace too small");
    return GN_EWORKSPACE;
  }
  const size_t xn = (size_t)a.B * (a.Lin + 2) * a.Cin, wn = (size_t)w_taps * a.Cin * a.Cout;
  unsigned short* xs = (unsigned short*)ws;
  unsigned short* wsp = xs + 3 * xn;
  if (split_x) {
    hipLaunchKernelGGL(split_x_kernel, dim3(cdiv(a.Lin, 64), cdiv(a.Cin, 64), a.B), dim3(256), 0, s, a.x, xs, a.B, a.Lin, a.Cin);
    int rc = check_launch("split_x");
    if (rc) return rc;
  }
  if (split_w) {
    hipLaunchKernelGGL(split_w_kernel, dim3(cdiv(wn / 8, 256)), dim3(256), 0, s, a.w, wsp, w_taps, a.Cin, a.Cout);
    int rc = check_launch("split_w");
    if (rc) return rc;
  }
  return GN_OK;
}

template <int NTAPS, int ASEG>
static int launch_bf16x3(const ConvArgs& a, int w_taps, void* ws, hipStream_t s) {
  const size_t lds = 3 * ((size_t)6 * ASEG * 1024 + (size_t)6 * NTAPS * 1024);
  if (lds > 160 * 1024) {
    set_error("conv_bf16x3: LDS %zu", lds);
    return GN_EINVAL;
  }
  static unsigned long long lds_done = 0;
    allow_big_lds((const void*)conv_bf16x3_kernel<NTAPS, ASEG>, &lds_done);
  const int m_tiles = (a.M + 127) / 128, n_tiles = a.Cout / 64;
  const size_t blocks = (size_t)m_tiles * n_tiles * a.B;
  const size_t xn = (size_t)a.B * (a.Lin + 2) * a.Cin, wn = (size_t)w_taps * a.Cin * a.Cout;
  const unsigned short* xs = (const unsigned short*)ws;
  const unsigned short* wsp = xs + 3 * xn;
  prof_begin(s);
  hipLaunchKernelGGL((conv_bf16x3_kernel<NTAPS, ASEG>), dim3((unsigned)blocks), dim3(256), lds, s, a, xs, wsp, xn, wn, m_tiles, n_tiles);
  prof_end(s, 2.0 * a.B * (double)a.M * a.t.ntaps * a.Cin * a.Cout, 2);
  return check_launch("conv_bf16x3");
}

template <int NTAPS, int ABL = 0, int IS = 1, int MERGE = 0>
static int launch_bf16x3_wide(const ConvArgs& a, int w_taps, void* ws, hipStream_t s) {
  constexpr size_t lds = 2 * ((size_t)6 * (IS == 1 ? 320 : 516) * 16 + (size_t)6 * NTAPS * 1024);
  static_assert(lds <= 160 * 1024, "two stages must fit the CU's LDS");
  static unsigned long long lds_done = 0;
  allow_big_lds((const void*)conv_bf16x3_wide_kernel<NTAPS, ABL, IS, MERGE>, &lds_done);
  const int m_tiles = (a.M + 255) / 256, n_tiles = a.Cout / 64;
  const size_t blocks = (size_t)m_tiles * n_tiles * a.B;
  if (blocks == 0 || blocks > 0x7fffffffull) {
    set_error("conv_bf16x3_wide: bad grid %zu", blocks);
    return GN_EINVAL;
  }
  const size_t xn = (size_t)a.B * (a.Lin + 2) * a.Cin, wn = (size_t)w_taps * a.Cin * a.Cout;
  const unsigned short* xs = (const unsigned short*)ws;
  const unsigned short* wsp = xs + 3 * xn;
  prof_begin(s);
  hipLaunchKernelGGL((conv_bf16x3_wide_kernel<NTAPS, ABL, IS, MERGE>), dim3((unsigned)blocks), dim3(256), lds, s, a, xs, wsp, xn, wn, m_tiles, n_tiles);
  prof_end(s, 2.0 * a.B * (double)a.M * a.t.ntaps * a.Cin * a.Cout, 2);
  return check_launch("conv_bf16x3_wide");
}

template <int NTAPS>
static int launch_bf16x3_wide3(const ConvArgs& a, int w_taps, void* ws, hipStream_t s) {
  constexpr size_t lds = 3 * ((size_t)6 * 320 * 16 + (size_t)6 * NTAPS * 1024);
  static_assert(lds <= 160 * 1024, "three stages must fit the CU's LDS");
  static unsigned long long lds_done = 0;
  allow_big_lds((const void*)conv_bf16x3_wide3_kernel<NTAPS>, &lds_done);
  const int m_tiles = (a.M + 255) / 256, n_tiles = a.Cout / 64;
  const size_t blocks = (size_t)m_tiles * n_tiles * a.B;
  if (blocks == 0 || blocks > 0x7fffffffull) {
    set_error("conv_bf16x3_wide3: bad grid %zu", blocks);
    return GN_EINVAL;
  }
  const size_t xn = (size_t)a.B * (a.Lin + 2) * a.Cin, wn = (size_t)w_taps * a.Cin * a.Cout;
  const unsigned short* xs = (const unsigned short*)ws;
  const unsigned short* wsp = xs + 3 * xn;
  prof_begin(s);
  hipLaunchKernelGGL((conv_bf16x3_wide3_kernel<NTAPS>), dim3((unsigned)blocks), dim3(256), lds, s, a, xs, wsp, xn, wn, m_tiles, n_tiles);
  prof_end(s, 2.0 * a.B * (double)a.M * a.t.ntaps * a.Cin * a.Cout, 2);
  return check_launch("conv_bf16x3_wide3");
}

// Both output phases of a stride-2 data gradient in one launch (a.t as conv_pipe_try_merged takes it: five taps in kernel-tap order, even index -> rows
// out_stride * m + out_off, odd index -> out_off_odd).  Returns 0 when the launch does not have the merged form's shape (the caller runs the phases).
int conv_bf16x3_merged_kind(const ConvArgs& a) {
  if (a.t.ntaps != 5 || a.t.in_stride != 1 || a.t.out_stride != 2 || a.stat_part || a.mask || a.bias || a.M < 192 || a.Cin % 16 || a.Cout % 64) return 0;
  for (int j = 0; j < 5; ++j)
    if (a.t.widx[j] != j) return 0;
  const int* o = a.t.off;
  if (o[1] == o[2] && o[3] == o[4] && o[0] == o[1] + 1 && o[3] == o[2] - 1) return 1;       // pad_left odd
  if (o[0] == o[1] && o[2] == o[3] && o[2] == o[1] - 1 && o[4] == o[3] - 1) return 2;       // pad_left even
  return 0;
}
int conv_bf16x3_run_merged(const ConvArgs& a, void* ws, hipStream_t s) {
  return conv_bf16x3_merged_kind(a) == 1 ? launch_bf16x3_wide<5, 0, 1, 1>(a, 5, ws, s) : launch_bf16x3_wide<5, 0, 1, 2>(a, 5, ws, s);
}

int conv_bf16x3_run(const ConvArgs& a, int w_taps, void* ws, hipStream_t s) {
  if (!conv_bf16x3_supported(a)) {
    set_error("conv_bf16x3: shape not supported (Cin %% 16, Cout %% 64, taps <= 5, unit input stride)");
    return GN_EINVAL;
  }
  if (a.t.in_stride == 2) return launch_bf16x3_wide<5, 0, 2>(a, w_taps, ws, s);
  // 64 x 64 wave tiles in 256-row blocks for the 4- / 5-tap launches whose rows fill them (round 4); shorter launches keep round 1's kernel
  if ((a.t.ntaps == 2 || a.t.ntaps == 3) && a.M >= 192) {
    int minoff = a.t.off[0], maxoff = a.t.off[0];
    for (int j = 1; j < a.t.ntaps; ++j) {
      minoff = std::min(minoff, a.t.off[j]);
      maxoff = std::max(maxoff, a.t.off[j]);
    }
    if (maxoff - minoff + 1 <= 5) return a.t.ntaps == 2 ? launch_bf16x3_wide3<2>(a, w_taps, ws, s) : launch_bf16x3_wide3<3>(a, w_taps, ws, s);
  }
  if (a.t.ntaps >= 4 && a.M >= 192) {
    int minoff = a.t.off[0], maxoff = a.t.off[0];
    for (int j = 1; j < a.t.ntaps; ++j) {
      minoff = std::min(minoff, a.t.off[j]);
      maxoff = std::max(maxoff, a.t.off[j]);
    }
    if (maxoff - minoff + 1 <= 5) {
#ifdef GN_ABLATION                  // timing experiments (wrong results): only in -DGN_ABLATION builds, never in the shipped library
      static const int abl = getenv("GN_BF16X3_ABL") ? atoi(getenv("GN_BF16X3_ABL")) : 0;
      if (abl && a.t.ntaps == 5) {
        switch (abl) {
          case 1: return launch_bf16x3_wide<5, 1>(a, w_taps, ws, s);
          case 2: return launch_bf16x3_wide<5, 2>(a, w_taps, ws, s);
          case 3: return launch_bf16x3_wide<5, 3>(a, w_taps, ws, s);
          case 4: return launch_bf16x3_wide<5, 4>(a, w_taps, ws, s);
          default: return launch_bf16x3_wide<5, 7>(a, w_taps, ws, s);
        }
      }
#endif
      return a.t.ntaps == 4 ? launch_bf16x3_wide<4>(a, w_taps, ws, s) : launch_bf16x3_wide<5>(a, w_taps, ws, s);
    }
  }
  switch (a.t.ntaps) {
    case 1: return launch_bf16x3<1, 3>(a, w_taps, ws, s);
    case 2: return launch_bf16x3<2, 3>(a, w_taps, ws, s);
    case 3: return launch_bf16x3<3, 3>(a, w_taps, ws, s);
    case 4: return launch_bf16x3<4, 3>(a, w_taps, ws, s);
    default: return launch_bf16x3<5, 3>(a, w_taps, ws, s);
  }
}

}  // namespace gn
