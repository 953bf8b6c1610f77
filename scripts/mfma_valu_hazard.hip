// Does a run of packed-fp32 VALU instructions right behind an fp32 MFMA disturb that MFMA's result?  (Found in round 5: conv_wino_kernel with its 14 transform
// instructions grouped in runs behind an MFMA produced rare wrong accumulator PAIRS -- registers 0-1 of one 16 x 16 tile -- where the same instructions spread
// one per MFMA slot never did.)  Every wave accumulates ones: after `iters` rounds every accumulator element must equal 4 * (its MFMAs) exactly.
//   MODE 0: MFMAs and the run alone.   MODE 1: as in the kernel, every eighth slot also issues two ds_read_b128 whose values are the B operands eight slots
//   later (counted lgkmcnt), so LDS returns land while the run executes.   RUNSLOT: the slot (0..7) behind whose MFMA the run sits.
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_valu_hazard.hip -o mfma_valu_hazard && ./mfma_valu_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int N, int GUARD, int RUNSLOT>
__global__ __launch_bounds__(256) void hazard_kernel(unsigned* bad, int iters) {
  __shared__ __attribute__((aligned(16))) float ones[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) ones[i] = 1.f;
  __syncthreads();
  f32x2 r[8];
  f32x4 acc[8];
  f32x4 B[2][2];
  for (int i = 0; i < 8; ++i) r[i] = f32x2{0.5f + i, 1.0f + i};
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  B[0][0] = B[0][1] = B[1][0] = B[1][1] = f32x4{1.f, 1.f, 1.f, 1.f};
  const float a = 1.f;
  const unsigned long long k = 0x3f0000003f000000ull;        // 0.5, 0.5
  const unsigned addr = (unsigned)(uintptr_t)ones + (threadIdx.x & 63) * 16;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {                            // two "points" per round: the B double buffer alternates
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        f32x4(&bc)[2] = B[p & 1];
        f32x4(&bn)[2] = B[(p + 1) & 1];
        const float bv = bc[(u & 3) >> 1][2 * (u & 1) + (u >> 2)];
        if (MODE == 1 && u == 0)
          asm volatile("ds_read_b128 %1, %5 offset:0\n\tds_read_b128 %2, %5 offset:1024\n\ts_waitcnt lgkmcnt(2)\n\tv_mfma_f32_16x16x4_f32 %0, %3, %4, %0"
                       : "+v"(acc[u & 3]), "=&v"(bn[0]), "=&v"(bn[1]) : "v"(a), "v"(bv), "v"(addr) : "memory");
        else
          asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[u & 3]) : "v"(a), "v"(bv) : "memory");
        if (u == RUNSLOT && p == 0) {
          if (GUARD > 0) asm volatile("s_nop %0" ::"n"(GUARD > 0 ? GUARD - 1 : 0));
#pragma unroll
          for (int g = 0; g < N; ++g)
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=&v"(r[(g + 4) & 7]) : "v"(r[g & 3]), "s"(k), "v"(r[(g + 1) & 3]));
        }
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
  const float want = 4.f * iters * 4;                      // each of the four accumulators takes 2 MFMAs per point, 2 points per round
  unsigned nb = 0;
  for (int u = 0; u < 4; ++u)
    for (int e = 0; e < 4; ++e)
      if (acc[u][e] != want) nb |= 1u << (u * 4 + e);
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += r[i][0] + r[i][1];
  if (nb || s == 12345.f) atomicOr(bad, nb | (s == 12345.f ? 0x80000000u : 0));
  if (nb && (threadIdx.x & 63) == 0) atomicAdd(bad + 1, 1u);
}

template <int MODE, int N, int GUARD, int RUNSLOT>
void run(int blocks) {
  unsigned* bad; hipMalloc(&bad, 8); hipMemset(bad, 0, 8);
  hipLaunchKernelGGL((hazard_kernel<MODE, N, GUARD, RUNSLOT>), dim3(blocks), dim3(256), 0, 0, bad, 1024);
  hipDeviceSynchronize();
  unsigned h[2]; hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost);
  printf("%s, run of %d packed fmas behind slot %d, guard %2d cycles, %5d blocks of 4 waves: %u waves with a wrong accumulator, elements mask (acc*4+reg) 0x%04x\n",
         MODE ? "with LDS reads" : "MFMA + VALU only", N, RUNSLOT, GUARD, blocks, h[1], h[0] & 0xffff);
  hipFree(bad);
}

int main() {
  for (int blocks : {768, 3072}) {
    run<0, 5, 0, 4>(blocks); run<1, 0, 0, 4>(blocks); run<1, 1, 0, 4>(blocks); run<1, 5, 0, 4>(blocks); run<1, 5, 0, 0>(blocks); run<1, 5, 0, 2>(blocks); run<1, 5, 0, 6>(blocks);
    run<1, 8, 0, 4>(blocks); run<1, 5, 4, 4>(blocks); run<1, 5, 16, 4>(blocks);
  }
  return 0;
}
