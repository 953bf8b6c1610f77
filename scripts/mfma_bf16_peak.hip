// What the bf16 matrix pipe of THIS card sustains, and therefore what a six-product bf16 split of an fp32 convolution (csrc/conv_bf16x3.hip, VERDICT r3
// next-round item 5) can reach at most: waves that issue v_mfma_f32_32x32x16_bf16 back to back from registers (no LDS, no memory) for ~50 ms per
// measurement, 1 / 2 / 4 waves per SIMD, operands that are (a) zero, (b) one constant, (c) random bf16 values spanning the exponent range an
// activation / weight split has (hi piece ~ N(0,1), mid and lo pieces 2^-8 and 2^-16 of it).  Datasheet: 256 CUs x 4 SIMDs x 1024 flop/cycle x 2.4 GHz
// = 2516.6 TFLOP/s dense bf16 (MI355X_MICROARCH guide: ~2.5 PFLOP/s); fp32-equivalent of the split = bf16 rate / 6.
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_bf16_peak.hip -o /tmp/mfma_bf16_peak && /tmp/mfma_bf16_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <cstring>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// 24 MFMAs per iteration = one 16-channel K step of a 64 x 64 wave tile under the six-product split (4 accumulator tiles x 6 products):
// A pieces a[2 row tiles][3], B pieces b[2 column tiles][3], products (0,0) (0,1) (1,0) (0,2) (2,0) (1,1).
__global__ __launch_bounds__(256) void bf16_loop_kernel(float* out, const u32x4* __restrict__ operands, int iters) {
#if defined(__HIP_DEVICE_COMPILE__)
  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  u32x4 a[2][3], b[2][3];
  const size_t lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      a[t][p] = operands[(lane * 12 + t * 3 + p) % (1u << 20)];
      b[t][p] = operands[(lane * 12 + 6 + t * 3 + p) % (1u << 20)];
    }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        f32x16& c = acc[mt * 2 + nt];
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a[mt][0]), "v"(b[nt][0]));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a[mt][0]), "v"(b[nt][1]));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a[mt][1]), "v"(b[nt][0]));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a[mt][0]), "v"(b[nt][2]));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a[mt][2]), "v"(b[nt][0]));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a[mt][1]), "v"(b[nt][1]));
      }
  }
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][15];
  out[lane] = s;
#endif
}

static unsigned short bf16_of(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("%s: %d CUs, clock %d MHz\n", prop.gcnArchName, cus, prop.clockRate / 1000);
  const double peak = cus * 4.0 * 1024.0 * 2.4e9 / 1e12;
  const size_t NOP = 1u << 20;
  std::vector<unsigned short> h(NOP * 8);
  u32x4* d_ops;
  float* d_out;
  CHECK(hipMalloc(&d_ops, NOP * 16));
  CHECK(hipMalloc(&d_out, (size_t)cus * 8 * 256 * 4 * sizeof(float)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const char* names[3] = {"zero operands", "one constant (1.0)", "random split pieces"};
  for (int data = 0; data < 3; ++data) {
    srand(1);
    for (size_t i = 0; i < NOP * 8; ++i) {
      float v = 0.f;
      if (data == 1) v = 1.0f;
      if (data == 2) {
        float g = 0.f;
        for (int k = 0; k < 12; ++k) g += (float)rand() / RAND_MAX;
        g -= 6.f;                                                    // ~N(0,1)
        const int piece = (int)((i / 8) % 3);                       // consecutive granules: hi, mid, lo pieces
        v = g * (piece == 0 ? 1.f : piece == 1 ? 1.f / 256.f : 1.f / 65536.f);
      }
      h[i] = bf16_of(v);
    }
    CHECK(hipMemcpy(d_ops, h.data(), NOP * 16, hipMemcpyHostToDevice));
    for (int wps : {1, 2, 4}) {
      const int blocks = cus * wps;                                  // 256 threads = 4 waves = one per SIMD; wps blocks per CU
      int iters = 100000;
      hipLaunchKernelGGL(bf16_loop_kernel, dim3(blocks), dim3(256), 0, 0, d_out, d_ops, 2000);
      CHECK(hipDeviceSynchronize());
      for (int rep = 0; rep < 3; ++rep) {                            // three back-to-back launches: the last one runs on a warm, power-limited chip
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(bf16_loop_kernel, dim3(blocks), dim3(256), 0, 0, d_out, d_ops, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double flop = (double)blocks * 4 * iters * 24 * 32768.0;
        const double tf = flop / (ms * 1e-3) / 1e12;
        printf("v_mfma_f32_32x32x16_bf16  %-22s waves/SIMD %d  run %d  %8.3f ms  %8.1f TFLOP/s  %.3f of %.0f   six-product fp32-equivalent %6.1f TFLOP/s\n",
               names[data], wps, rep, ms, tf, tf / peak, peak, tf / 6.0);
      }
    }
  }
  return 0;
}
