// Epilogue shared by the hand-scheduled fp32 conv kernel (conv_pipe.hip) and the wide-tile bf16-split kernel (conv_bf16x3.hip): both leave a
// wave's outputs as 32 x 32 MFMA accumulator tiles acc[WMT][WN] (WMT = 2 row tiles per wave; 1 in the transform-domain kernel, conv_wino.hip) (rows m_base + mt*32 + (r & 3) + 8 (r >> 2) + 4 h, columns n_base + nt*32 + i32).
#pragma once
#include "common.h"
#ifndef GN_STORE_AUX
#define GN_STORE_AUX 0      // cache policy of the output stores.  2 (nt, streaming) was measured: conv +0.5 %, but the weight gradient that reads
                            // the tensor next lost 2.7 % (144.8 -> 140.9 TFLOP/s): the write-back copies in L2 / Infinity Cache are worth keeping
#endif

namespace gn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Epilogue of the pipelined kernel.  The activation kind and the fused variants are dispatched ONCE per wave (template parameters), so
// the 64 outputs of a lane are straight-line code: the generic epilogue of conv_mfma_kernel re-decides the activation per element and,
// fully unrolled, is ~22k instructions of branches and waits (15 us per block against ~1 us here; measured with s_memrealtime stamps).
// Addresses: one buffer descriptor per batch element, a per-lane byte offset (column, row-within-quad) and a scalar row offset per
// register, so no 64-bit address arithmetic; rows past the end of the output (m >= M) fall outside the descriptor's range and are
// dropped by the hardware bounds check (out row = out_stride*m + out_off >= Ly exactly when m >= M).
// MODE 0: y = act(acc + bias);  MODE 1: ... then the fused Dropout keep-mask;  MODE 2: data gradient times the producer's act'(gy);
// MODE 3: MODE 2 through the producer's dropout.
struct EpiSrd {
  __amdgpu_buffer_rsrc_t y, m, g;      // output, u8 mask (MODE 1: a.mask, MODE 3: a.gmask), producer output gy (MODE >= 2) of batch element b
  float ginv;
};
template <int MODE>
__device__ __forceinline__ EpiSrd epi_srd(const ConvArgs& a, int b) {
  EpiSrd s;
  const uintptr_t yp = (uintptr_t)(a.y + (size_t)b * a.Ly * a.Cout);
  const unsigned ylo = __builtin_amdgcn_readfirstlane((unsigned)yp), yhi = __builtin_amdgcn_readfirstlane((unsigned)(yp >> 32));
  const int ybytes = __builtin_amdgcn_readfirstlane(a.Ly * a.Cout * 4);
  s.y = __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)yhi << 32) | ylo), 0, ybytes, 0x00020000);
  s.m = s.y; s.g = s.y;
  if (MODE == 1 || MODE == 3) {                            // u8 masks: same element indexing, one byte per element
    const uintptr_t mp = (uintptr_t)((MODE == 1 ? a.mask : a.gmask) + (size_t)b * a.Ly * a.Cout);
    const unsigned mlo = __builtin_amdgcn_readfirstlane((unsigned)mp), mhi = __builtin_amdgcn_readfirstlane((unsigned)(mp >> 32));
    s.m = __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)mhi << 32) | mlo), 0, ybytes >> 2, 0x00020000);
  }
  if (MODE >= 2) {
    const uintptr_t gp = (uintptr_t)(a.gy + (size_t)b * a.Ly * a.Cout);
    const unsigned glo = __builtin_amdgcn_readfirstlane((unsigned)gp), ghi = __builtin_amdgcn_readfirstlane((unsigned)(gp >> 32));
    s.g = __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)ghi << 32) | glo), 0, ybytes, 0x00020000);
  }
  s.ginv = (MODE == 3) ? 1.0f / a.gscale : 1.0f;
  return s;
}
// one output element: v = accumulator + bias; voff = per-lane element offset, soff = wave-uniform element offset inside the batch element
template <int ACT, int MODE, int GACT>
__device__ __forceinline__ void epi_store(const ConvArgs& a, const EpiSrd& s, float v, int voff, int soff) {
  if (ACT == GN_ACT_RELU) v = fmaxf(v, 0.f);
  else if (ACT == GN_ACT_LEAKY) v = v > 0.f ? v : a.act_param * v;
  else if (ACT == GN_ACT_TANH) v = gn_tanhf(v);
  else if (ACT != GN_ACT_LINEAR) v = act_apply(v, a.act, a.act_param);         // rare kinds: runtime switch
  if (MODE == 1) {
    const unsigned k = __builtin_amdgcn_raw_buffer_load_b8(s.m, voff, soff, 0);
    v = k ? v * a.keep_scale : 0.f;
  }
  if (MODE >= 2) {
    const float gv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(s.g, voff * 4, soff * 4, 0));
    if (MODE == 3) {
      const unsigned k = __builtin_amdgcn_raw_buffer_load_b8(s.m, voff, soff, 0);
      const float y0 = gv * s.ginv;
      const float dg = GACT == GN_ACT_LEAKY ? (y0 > 0.f ? 1.f : a.gparam) : act_grad_from_y(y0, a.gact, a.gparam);
      v = k ? v * a.gscale * dg : 0.f;
    } else {
      v *= GACT == GN_ACT_RELU ? (gv > 0.f ? 1.f : 0.f) : act_grad_from_y(gv, a.gact, a.gparam);
    }
  }
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), s.y, voff * 4, soff * 4, GN_STORE_AUX);
}

template <int ACT, int MODE, int GACT, int WN, int WMT = 2>
__device__ __forceinline__ void pipe_epilogue(const ConvArgs& a, const f32x16 (&acc)[WMT][WN], int b, int m_base, int n_base, int i32, int h, int out_off) {
  const EpiSrd srd = epi_srd<MODE>(a, b);
  const int rowstride = a.t.out_stride * a.Cout;           // elements between consecutive m
#pragma unroll
  for (int nt = 0; nt < WN; ++nt) {
    const int n = n_base + nt * 32 + i32;
    const float bias = a.bias ? a.bias[n] : 0.f;
    const int voff = rowstride * (4 * h) + out_off * a.Cout + n;                   // element offset of (row 4h, column n)
#pragma unroll
    for (int mt = 0; mt < WMT; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int soff = rowstride * (m_base + mt * 32 + (r & 3) + 8 * (r >> 2));   // wave-uniform
        epi_store<ACT, MODE, GACT>(a, srd, acc[mt][nt][r] + bias, voff, soff);
      }
    }
  }
}

// The same for a wave tile of 16 x 16 accumulator tiles (v_mfma_f32_16x16x4_f32: lane (n16 = lane & 15, kq = lane >> 4) holds rows 4 kq + r,
// r = 0..3, of column n16): acc[NT4] = NT4 column tiles of 16 rows starting at m_base (conv_wino.hip).
typedef float f32x4e __attribute__((ext_vector_type(4)));
template <int ACT, int MODE, int GACT, int NT4>
__device__ __forceinline__ void tile16_epilogue(const ConvArgs& a, const f32x4e (&acc)[NT4], int b, int m_base, int n_base, int n16, int kq, int out_off) {
  const EpiSrd srd = epi_srd<MODE>(a, b);
  const int rowstride = a.t.out_stride * a.Cout;
#pragma unroll
  for (int ct = 0; ct < NT4; ++ct) {
    const int n = n_base + ct * 16 + n16;
    const float bias = a.bias ? a.bias[n] : 0.f;
    const int voff = rowstride * (4 * kq) + out_off * a.Cout + n;
#pragma unroll
    for (int r = 0; r < 4; ++r) epi_store<ACT, MODE, GACT>(a, srd, acc[ct][r] + bias, voff, rowstride * (m_base + r));
  }
}
template <int NT4>
__device__ __forceinline__ void tile16_epilogue_dispatch(const ConvArgs& a, const f32x4e (&acc)[NT4], int b, int m_base, int n_base, int n16, int kq, int out_off,
                                                         int mode) {
#define GN_EPI(A_, M_, G_) tile16_epilogue<A_, M_, G_, NT4>(a, acc, b, m_base, n_base, n16, kq, out_off)
  if (mode == 0) {
    switch (a.act) {
      case GN_ACT_LINEAR: GN_EPI(GN_ACT_LINEAR, 0, -1); break;
      case GN_ACT_RELU: GN_EPI(GN_ACT_RELU, 0, -1); break;
      case GN_ACT_LEAKY: GN_EPI(GN_ACT_LEAKY, 0, -1); break;
      case GN_ACT_TANH: GN_EPI(GN_ACT_TANH, 0, -1); break;
      default: GN_EPI(-1, 0, -1); break;
    }
  } else if (mode == 1) {
    if (a.act == GN_ACT_LEAKY) GN_EPI(GN_ACT_LEAKY, 1, -1);
    else GN_EPI(-1, 1, -1);
  } else if (mode == 2) {
    if (a.act == GN_ACT_LINEAR && a.gact == GN_ACT_RELU) GN_EPI(GN_ACT_LINEAR, 2, GN_ACT_RELU);
    else GN_EPI(-1, 2, -1);
  } else {
    if (a.act == GN_ACT_LINEAR && a.gact == GN_ACT_LEAKY) GN_EPI(GN_ACT_LINEAR, 3, GN_ACT_LEAKY);
    else GN_EPI(-1, 3, -1);
  }
#undef GN_EPI
}

// uniform dispatch, decided once per wave; every case is straight-line code.  Specialised: the forms the three networks run (forward
// linear / relu / LeakyReLU / tanh, LeakyReLU + dropout, data gradient through relu, through LeakyReLU + dropout); the rest take the
// variants that decide the activation per element.
template <int WN, int WMT = 2>
__device__ __forceinline__ void pipe_epilogue_dispatch(const ConvArgs& a, const f32x16 (&acc)[WMT][WN], int b, int m_base, int n_base, int i32, int h, int out_off,
                                                       int mode) {
#define GN_EPI(A_, M_, G_) pipe_epilogue<A_, M_, G_, WN, WMT>(a, acc, b, m_base, n_base, i32, h, out_off)
  if (mode == 0) {
    switch (a.act) {
      case GN_ACT_LINEAR: GN_EPI(GN_ACT_LINEAR, 0, -1); break;
      case GN_ACT_RELU: GN_EPI(GN_ACT_RELU, 0, -1); break;
      case GN_ACT_LEAKY: GN_EPI(GN_ACT_LEAKY, 0, -1); break;
      case GN_ACT_TANH: GN_EPI(GN_ACT_TANH, 0, -1); break;
      default: GN_EPI(-1, 0, -1); break;
    }
  } else if (mode == 1) {
    if (a.act == GN_ACT_LEAKY) GN_EPI(GN_ACT_LEAKY, 1, -1);
    else GN_EPI(-1, 1, -1);
  } else if (mode == 2) {
    if (a.act == GN_ACT_LINEAR && a.gact == GN_ACT_RELU) GN_EPI(GN_ACT_LINEAR, 2, GN_ACT_RELU);
    else GN_EPI(-1, 2, -1);
  } else {
    if (a.act == GN_ACT_LINEAR && a.gact == GN_ACT_LEAKY) GN_EPI(GN_ACT_LINEAR, 3, GN_ACT_LEAKY);
    else GN_EPI(-1, 3, -1);
  }
#undef GN_EPI
}

}  // namespace gn
