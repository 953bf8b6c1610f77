// Issue cost of the packed-fp32 instructions the transform-domain kernels are built from, alone and beside the fp32 MFMA (one wave per SIMD, wall_clock64
// around an unrolled stream; gfx950).   hipcc --offload-arch=gfx950 -O3 scripts/valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(64) void rate_kernel(float* out, long long* cyc, int iters) {
  f32x2 r[8];
  f32x4 acc[4];
  for (int i = 0; i < 8; ++i) r[i] = f32x2{(float)threadIdx.x + i, 1.0f + i};
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x, b = 1.f;
  long long t0 = wall_clock64();
  long long c0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      f32x2& x = r[u & 7];
      f32x2& y = r[(u + 3) & 7];
      if (KIND == 0) asm volatile("v_pk_fma_f32 %0, %1, 2.0, %0 op_sel_hi:[1,0,1]" : "+v"(x) : "v"(y));
      if (KIND == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
      if (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(x) : "v"(y));
      if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(y));
      if (KIND == 4) asm volatile("v_fma_f32 %0, %1, 2.0, %0" : "+v"(x[0]) : "v"(y[0]));
      if (KIND == 5) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[0]) : "v"(y[0]));
      if (KIND == 6) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[u & 3]) : "v"(a), "v"(b));
      if (KIND == 7) {          // one MFMA + one packed fma, alternating: do they overlap?
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[u & 3]) : "v"(a), "v"(b));
        asm volatile("v_pk_fma_f32 %0, %1, 2.0, %0 op_sel_hi:[1,0,1]" : "+v"(x) : "v"(y));
      }
      if (KIND == 8) {          // one MFMA + one packed add
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[u & 3]) : "v"(a), "v"(b));
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
      }
      if (KIND == 10 || KIND == 11 || KIND == 12) {          // GROUPED: G MFMAs, then G packed fmas (per 16-step unroll: 16 of each, in groups of G = 2, 4, 16)
        constexpr int G = KIND == 10 ? 2 : KIND == 11 ? 4 : 16;
        if (u % G == 0) {
#pragma unroll
          for (int g = 0; g < G; ++g) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[(u + g) & 3]) : "v"(a), "v"(b));
#pragma unroll
          for (int g = 0; g < G; ++g) asm volatile("v_pk_fma_f32 %0, %1, 2.0, %0 op_sel_hi:[1,0,1]" : "+v"(r[(u + g) & 7]) : "v"(r[(u + g + 3) & 7]));
        }
      }
      if (KIND == 13) {         // 3 MFMAs per packed fma (the kernels' ratio), the fma alone between MFMAs
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[u & 3]) : "v"(a), "v"(b));
        if (u % 3 == 0) asm volatile("v_pk_fma_f32 %0, %1, 2.0, %0 op_sel_hi:[1,0,1]" : "+v"(x) : "v"(y));
      }
      if (KIND == 14) {         // the same 16 MFMAs and 6 packed fmas, the fmas in two groups of three
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[u & 3]) : "v"(a), "v"(b));
        if (u == 4 || u == 12) {
#pragma unroll
          for (int g = 0; g < 3; ++g) asm volatile("v_pk_fma_f32 %0, %1, 2.0, %0 op_sel_hi:[1,0,1]" : "+v"(r[(u + g) & 7]) : "v"(r[(u + g + 3) & 7]));
        }
      }
      if (KIND == 9) {          // one MFMA + one v_add_u32
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[u & 3]) : "v"(a), "v"(b));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[0]) : "v"(y[0]));
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15");
  long long c1 = clock64();
  long long t1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += r[i][0] + r[i][1];
  for (int i = 0; i < 4; ++i) s += acc[i][0];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = c1 - c0; cyc[1] = t1 - t0; }
}

template <int KIND>
void run(const char* name, int per_iter) {
  float* out; long long* cyc;
  hipMalloc(&out, 64 * 1024 * sizeof(float)); hipMalloc(&cyc, 16);
  const int iters = 4096;
  for (int blocks : {1, 1024}) {          // one wave on the chip / one wave per SIMD on every CU
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    printf("%-34s blocks %5d: %7.2f shader clocks per %s (s_memtime), %.3f ms\n", name, blocks, (double)h[0] / ((double)iters * 16), per_iter == 2 ? "PAIR" : "instruction", ms);
  }
}

int main() {
  run<0>("v_pk_fma_f32", 1); run<1>("v_pk_add_f32", 1); run<2>("v_pk_add_f32 neg", 1); run<3>("v_pk_mul_f32", 1); run<4>("v_fma_f32", 1); run<5>("v_add_u32", 1);
  run<6>("v_mfma_f32_16x16x4_f32", 1); run<7>("mfma + v_pk_fma_f32", 2); run<8>("mfma + v_pk_add_f32", 2); run<9>("mfma + v_add_u32", 2);
  run<10>("2 mfma, 2 pk_fma grouped", 2); run<11>("4 mfma, 4 pk_fma grouped", 2); run<12>("16 mfma, 16 pk_fma grouped", 2);
  run<13>("16 mfma + 6 pk_fma spread", 2); run<14>("16 mfma + 6 pk_fma in 2 groups", 2);
  return 0;
}
