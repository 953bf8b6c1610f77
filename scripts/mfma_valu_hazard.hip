// Does a run of packed-fp32 VALU instructions right behind an fp32 MFMA disturb that MFMA's result?  (Found in round 5: conv_wino_kernel with its 14 transform
// instructions grouped in runs behind an MFMA produced rare wrong accumulator PAIRS -- registers 0-1 of the MFMA in front of the run -- where the same
// instructions spread one per MFMA slot never did.)  Every wave accumulates ones: after `iters` rounds every accumulator element must equal 4 * iters * (MFMAs per
// round on that accumulator) exactly.   hipcc --offload-arch=gfx950 -O3 scripts/mfma_valu_hazard.hip -o mfma_valu_hazard && ./mfma_valu_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int N, int GUARD>
__global__ __launch_bounds__(64) void hazard_kernel(unsigned* bad, int iters) {
  f32x2 r[8];
  f32x4 acc[4];
  for (int i = 0; i < 8; ++i) r[i] = f32x2{0.5f + i, 1.0f + i};
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float a = 1.f, b = 1.f;
  const unsigned long long k = 0x3f0000003f000000ull;        // 0.5, 0.5
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(a), "v"(b));
      if (u == 0) {
        if (GUARD > 0) asm volatile("s_nop %0" ::"n"(GUARD > 0 ? GUARD - 1 : 0));
#pragma unroll
        for (int g = 0; g < N; ++g)
          asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=&v"(r[(g + 4) & 7]) : "v"(r[g & 3]), "s"(k), "v"(r[(g + 1) & 3]));
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
  const float want = 4.f * iters;
  unsigned nb = 0;
  for (int u = 0; u < 4; ++u)
    for (int e = 0; e < 4; ++e)
      if (acc[u][e] != want) nb |= 1u << (u * 4 + e);
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += r[i][0] + r[i][1];
  if (nb || s == 12345.f) atomicOr(bad, nb | (s == 12345.f ? 0x80000000u : 0));
  if (nb) atomicAdd(bad + 1, 1u);
}

template <int N, int GUARD>
void run(int blocks) {
  unsigned* bad; hipMalloc(&bad, 8); hipMemset(bad, 0, 8);
  hipLaunchKernelGGL((hazard_kernel<N, GUARD>), dim3(blocks), dim3(64), 0, 0, bad, 2048);
  hipDeviceSynchronize();
  unsigned h[2]; hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost);
  printf("run of %d packed fmas behind the MFMA, guard %2d cycles, %5d waves: %u waves with a wrong accumulator, elements mask (acc*4+reg) 0x%04x\n", N, GUARD, blocks, h[1], h[0] & 0xffff);
  hipFree(bad);
}

int main() {
  for (int blocks : {1024, 3072, 8192}) {
    run<0, 0>(blocks); run<1, 0>(blocks); run<2, 0>(blocks); run<3, 0>(blocks); run<5, 0>(blocks); run<8, 0>(blocks);
    run<5, 4>(blocks); run<5, 8>(blocks); run<5, 16>(blocks); run<8, 16>(blocks);
  }
  return 0;
}
