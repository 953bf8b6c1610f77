// Shared by the transform-domain kernels (conv_wino.hip: forward / data gradient; wgrad_wino.hip: weight gradient): the F(2,5) input transform on
// register pairs, as packed fp32 asm.
#pragma once
#include "common.h"

namespace gn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------
// BT of F(2,5) on {0, 1, -1, 1/2, -2, inf} on a channel PAIR per lane, rows scaled by (1/2, 1/2, 1/2, 1, 1, 1/2) against the integer form (the inverse
// factors sit in G: wino_u_kernel, wgrad_wino_reduce_kernel; powers of two, so no rounding changes):
//   a = d1 - d3, b = d2 - d4
//   v0 = (d0 - d2) - b - 1.5 a           v1 = s + t,  v2 = s - t  with  s = 1.5 d2 - a,  t = 1.5 d3 - b
//   v3 = 2 a + b                         v4 = a - 2 b                        v5 = (d5 - d3) + a - 1.5 b
// FOURTEEN packed instructions (the first form of the round took 18: every instruction here costs the matrix pipe its full issue time, so the sums were
// refactored until no coefficient needed an instruction of its own: v1 / v2 share s and t, 2 d0 - 4 d2 + 2 d4 = 2 ((d0 - d2) - b)), written as asm: hipcc moves
// plain fma code away from the MFMA slots it is meant to sit between.  1.5 is not an inline constant: SGPR pairs (k15 = 1.5, km15 = -1.5).  Piece K of
// wino_piece is one instruction; the order interleaves the dependency chains; K >= 14 is empty (the weight gradient's slot map has 18 places).
// ---------------------------------------------------------------------------------------------
struct WinoT {
  f32x2 a, b;
};
constexpr int kWinoPieces = 14;
#define GN_PK_SUB(o, x, y) asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=&v"(o) : "v"(x), "v"(y))
#define GN_PK_SUB_SELF(o, y) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(o) : "v"(y))
#define GN_PK_ADD_SELF(o, y) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(o) : "v"(y))
#define GN_PK_FMA_NEW(o, x, c, z) asm volatile("v_pk_fma_f32 %0, %1, " c ", %2 op_sel_hi:[1,0,1]" : "=&v"(o) : "v"(x), "v"(z))
#define GN_PK_FMA_SELF(o, c, z) asm volatile("v_pk_fma_f32 %0, %0, " c ", %1 op_sel_hi:[1,0,1]" : "+v"(o) : "v"(z))
#define GN_PK_FMA_NEWS_NEG(o, x, k, z) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=&v"(o) : "v"(x), "s"(k), "v"(z))
#define GN_PK_FMA_ACCS(o, x, k) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(o) : "v"(x), "s"(k))
template <int K>
__device__ __forceinline__ void wino_piece(const f32x2 (&d)[6], f32x2 (&v)[6], WinoT& t, unsigned long long k15, unsigned long long km15) {
  if constexpr (K == 0) GN_PK_SUB(t.a, d[1], d[3]);
  else if constexpr (K == 1) GN_PK_SUB(t.b, d[2], d[4]);
  else if constexpr (K == 2) GN_PK_SUB(v[0], d[0], d[2]);
  else if constexpr (K == 3) GN_PK_SUB(v[5], d[5], d[3]);
  else if constexpr (K == 4) GN_PK_FMA_NEWS_NEG(v[1], d[2], k15, t.a);          // s = 1.5 d2 - a
  else if constexpr (K == 5) GN_PK_FMA_NEWS_NEG(v[2], d[3], k15, t.b);          // t = 1.5 d3 - b
  else if constexpr (K == 6) GN_PK_FMA_NEW(v[3], t.a, "2.0", t.b);
  else if constexpr (K == 7) GN_PK_FMA_NEW(v[4], t.b, "-2.0", t.a);
  else if constexpr (K == 8) GN_PK_SUB_SELF(v[0], t.b);
  else if constexpr (K == 9) GN_PK_ADD_SELF(v[5], t.a);
  else if constexpr (K == 10) GN_PK_FMA_ACCS(v[0], t.a, km15);
  else if constexpr (K == 11) GN_PK_FMA_ACCS(v[5], t.b, km15);
  else if constexpr (K == 12) GN_PK_ADD_SELF(v[1], v[2]);                       // s + t
  else if constexpr (K == 13) GN_PK_FMA_SELF(v[2], "-2.0", v[1]);               // (s + t) - 2 t = s - t
}
// pieces K0 .. K1-1 back to back
template <int K0, int K1>
__device__ __forceinline__ void wino_run(const f32x2 (&d)[6], f32x2 (&v)[6], WinoT& t, unsigned long long k15, unsigned long long km15) {
  if constexpr (K0 < K1) {
    wino_piece<K0>(d, v, t, k15, km15);
    wino_run<K0 + 1, K1>(d, v, t, k15, km15);
  }
}
template <int K = 0>
__device__ __forceinline__ void wino_bt_all(const f32x2 (&d)[6], f32x2 (&v)[6], WinoT& t, unsigned long long k15, unsigned long long km15) {
  if constexpr (K < kWinoPieces) {
    wino_piece<K>(d, v, t, k15, km15);
    wino_bt_all<K + 1>(d, v, t, k15, km15);
  }
}


// A of F(2,5) (= AT transposed) on a pair of output rows (e0, e1), the dy side of the weight gradient dW = G^T [ sum_tiles (BT x) . (A dy) ]:
//   p0 = e0, p1 = e0 + e1, p2 = e0 - e1, p3 = e0 + e1 / 2, p4 = e0 - 2 e1, p5 = e1        (p0 and p5 are the operands themselves)
template <int K>
__device__ __forceinline__ void wino_a_piece(const f32x2& e0, const f32x2& e1, f32x2& p1, f32x2& p2, f32x2& p3, f32x2& p4) {
  if constexpr (K == 0) asm volatile("v_pk_add_f32 %0, %1, %2" : "=&v"(p1) : "v"(e0), "v"(e1));
  else if constexpr (K == 1) GN_PK_SUB(p2, e0, e1);
  else if constexpr (K == 2) GN_PK_FMA_NEW(p3, e1, "0.5", e0);
  else GN_PK_FMA_NEW(p4, e1, "-2.0", e0);
}

}  // namespace gn
