// Shared by the transform-domain kernels (conv_wino.hip: forward / data gradient; wgrad_wino.hip: weight gradient): the F(2,5) input transform on
// register pairs, as packed fp32 asm.
#pragma once
#include "common.h"

namespace gn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------
// BT of F(2,5) on {0, 1, -1, 1/2, -2, inf}, rows scaled to integers (the inverse factors sit in G above), on a channel PAIR per lane:
//   a = d1 - d3, b = d2 - d4
//   v0 = 2 d0 - 3 a - 4 d2 + 2 d4        v1 = -2 d1 + d2 + 5 d3 + 2 d4       v2 = -2 d1 + 5 d2 - d3 - 2 d4
//   v3 = 2 a + b                         v4 = a - 2 b                        v5 = 2 d5 + 2 a - 2 d3 - 3 b
// 18 packed instructions, written as asm: hipcc moves plain fma code away from the MFMA slots it is meant to sit between.  3 and 5 are not inline
// constants: SGPR pairs.  Piece K of wino_piece is one instruction; the order interleaves the six dependency chains.
// ---------------------------------------------------------------------------------------------
struct WinoT {
  f32x2 a, b;
};
#define GN_PK_SUB(o, x, y) asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=&v"(o) : "v"(x), "v"(y))
#define GN_PK_DBL(o, x) asm volatile("v_pk_add_f32 %0, %1, %1" : "=&v"(o) : "v"(x))
#define GN_PK_FMA_NEW(o, x, c, z) asm volatile("v_pk_fma_f32 %0, %1, " c ", %2 op_sel_hi:[1,0,1]" : "=&v"(o) : "v"(x), "v"(z))
#define GN_PK_FMA_NEWNEG(o, x, c, z) asm volatile("v_pk_fma_f32 %0, %1, " c ", %2 op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=&v"(o) : "v"(x), "v"(z))
#define GN_PK_FMA_ACC(o, x, c) asm volatile("v_pk_fma_f32 %0, %1, " c ", %0 op_sel_hi:[1,0,1]" : "+v"(o) : "v"(x))
#define GN_PK_FMA_ACCS(o, x, k) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(o) : "v"(x), "s"(k))
template <int K>
__device__ __forceinline__ void wino_piece(const f32x2 (&d)[6], f32x2 (&v)[6], WinoT& t, unsigned long long k5, unsigned long long km3) {
  if constexpr (K == 0) GN_PK_SUB(t.a, d[1], d[3]);
  else if constexpr (K == 1) GN_PK_SUB(t.b, d[2], d[4]);
  else if constexpr (K == 2) GN_PK_FMA_NEW(v[1], d[4], "2.0", d[2]);
  else if constexpr (K == 3) GN_PK_FMA_NEWNEG(v[2], d[4], "-2.0", d[3]);
  else if constexpr (K == 4) GN_PK_DBL(v[0], d[0]);
  else if constexpr (K == 5) GN_PK_DBL(v[5], d[5]);
  else if constexpr (K == 6) GN_PK_FMA_NEW(v[3], t.a, "2.0", t.b);
  else if constexpr (K == 7) GN_PK_FMA_NEW(v[4], t.b, "-2.0", t.a);
  else if constexpr (K == 8) GN_PK_FMA_ACCS(v[1], d[3], k5);
  else if constexpr (K == 9) GN_PK_FMA_ACCS(v[2], d[2], k5);
  else if constexpr (K == 10) GN_PK_FMA_ACCS(v[0], t.a, km3);
  else if constexpr (K == 11) GN_PK_FMA_ACC(v[5], t.a, "2.0");
  else if constexpr (K == 12) GN_PK_FMA_ACC(v[1], d[1], "-2.0");
  else if constexpr (K == 13) GN_PK_FMA_ACC(v[2], d[1], "-2.0");
  else if constexpr (K == 14) GN_PK_FMA_ACC(v[0], d[2], "-4.0");
  else if constexpr (K == 15) GN_PK_FMA_ACC(v[5], d[3], "-2.0");
  else if constexpr (K == 16) GN_PK_FMA_ACC(v[0], d[4], "2.0");
  else GN_PK_FMA_ACCS(v[5], t.b, km3);
}
template <int K = 0>
__device__ __forceinline__ void wino_bt_all(const f32x2 (&d)[6], f32x2 (&v)[6], WinoT& t, unsigned long long k5, unsigned long long km3) {
  if constexpr (K < 18) {
    wino_piece<K>(d, v, t, k5, km3);
    wino_bt_all<K + 1>(d, v, t, k5, km3);
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
