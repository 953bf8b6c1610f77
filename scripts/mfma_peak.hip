// What the fp32 matrix pipe of THIS card delivers with nothing else in the way: waves that issue v_mfma_f32_32x32x2_f32 back to back from
// registers (no LDS, no memory), 1 / 2 / 4 / 8 waves per SIMD, 1 / 2 / 4 independent accumulator tiles per wave.  The datasheet figure the
// roofline divides by is 256 CUs x 4 SIMDs x 64 flop/cycle x 2.4 GHz = 157.3 TFLOP/s; this prints what fraction of it the bare instruction
// stream reaches, i.e. the ceiling of MfmaUtil for any kernel built on this instruction.
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void mfma32_kernel(float* out, int iters) {
  f32x16 acc[NACC];
#pragma unroll
  for (int j = 0; j < NACC; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float a = (float)(threadIdx.x & 7) * 0.125f, b = 1.0f / 1024.0f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 32 / NACC; ++u)
#pragma unroll
      for (int j = 0; j < NACC; ++j) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
  }
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][15];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void mfma16_kernel(float* out, int iters) {
  f32x4 acc[NACC];
#pragma unroll
  for (int j = 0; j < NACC; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
  float a = (float)(threadIdx.x & 7) * 0.125f, b = 1.0f / 1024.0f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 32 / NACC; ++u)
#pragma unroll
      for (int j = 0; j < NACC; ++j) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
  }
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][3];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// The conv kernel's chunk loop, piece by piece: 80 MFMAs per wave between two block barriers (4 waves per block, one per SIMD, W blocks per
// CU); + the chunk's LDS operand reads (10 ds_read_b128 + 40 ds_read_b32 per wave, conflict-free, counted waits as in conv_pipe.hip); + the
// chunk's staging traffic as LDS-DMA: NDMA wave loads of SZ bytes per lane, through global_load_lds with a 64-bit address per lane (MODE 1)
// or buffer_load ... lds with a 32-bit lane offset and the chunk offset in an SGPR (MODE 2), all issued after the barrier or (SPREAD) one
// every 80 / NDMA MFMAs.  Addresses are shared by the blocks of a slab group (8 groups), so most loads are L2 hits, as in the kernel; they
// are drained (vmcnt(0)) in front of the barrier.  Says which piece the gap between the bare pipe and the kernel belongs to.
// TAPS (round 4): 5 = the 80-MFMA chunk of the 5-tap kernels; 3 / 2 = the 48- / 32-MFMA chunks of the stride-2 data gradient's phases
template <bool LDS_READS, int MODE, int NDMA, int SZ, bool SPREAD, int TAPS = 5>
__global__ __launch_bounds__(256) void chunk_loop_kernel(float* out, int iters, const float* src, unsigned src_mask) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) float lds[9216];           // 36 KiB: four blocks per CU, as the kernel
  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  for (int i = threadIdx.x; i < 9216; i += 256) lds[i] = 1.0f / 1024.0f;
  __syncthreads();
  const int lane = threadIdx.x & 63, i32 = lane & 31, h = lane >> 5, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row = wave * 64 + i32;
  const unsigned lds0 = (unsigned)(uintptr_t)lds;
  const unsigned addr_a = lds0 + (row * 2 + (h ^ ((row >> 3) & 1))) * 16;
  const unsigned addr_b = lds0 + (2112 + 4 * h * 64 + i32) * 4;
  f32x4 a0 = {0.125f, 0.25f, 0.5f, 1.f}, a1 = a0;
  float b0 = 1.0f / 1024.0f, b1 = b0;
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const unsigned grp = __builtin_amdgcn_readfirstlane((blockIdx.x >> 3) & 7);
  constexpr int WAVE_BYTES = 64 * SZ;                                // bytes one wave load moves
  constexpr int NSLOT = NDMA > 0 ? NDMA : 1;
  const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x40000000, 0x00020000);
  int voff[NSLOT];
#pragma unroll
  for (int k = 0; k < NSLOT; ++k) voff[k] = (wave * NSLOT + k) * WAVE_BYTES + lane * SZ;
  auto dma = [&](int i, int k) {
    float* dst = lds + 3072 + ((wave * NSLOT + k) * WAVE_BYTES) / 4 % 6144;
    const unsigned chunk_bytes = (((unsigned)i * 8u + grp) * 4u * NSLOT * WAVE_BYTES) & (src_mask * 4u) & 0x3ff00000u;      // 1-MiB steps inside the first GiB - 1 MiB: the lane offsets (< 64 KiB) stay inside the buffer
    if constexpr (MODE == 1) {
      if constexpr (SZ == 16) __builtin_amdgcn_global_load_lds((gptr_t)((const char*)src + chunk_bytes + voff[k]), (lptr_t)dst, 16, 0, 0);
      else __builtin_amdgcn_global_load_lds((gptr_t)((const char*)src + chunk_bytes + voff[k]), (lptr_t)dst, 4, 0, 0);
    } else if constexpr (MODE == 2) {
      if constexpr (SZ == 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lptr_t)dst, 16, voff[k], (int)chunk_bytes, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lptr_t)dst, 4, voff[k], (int)chunk_bytes, 0, 0);
    }
  };
  for (int i = 0; i < iters; ++i) {
    if constexpr (MODE != 0 && !SPREAD) {
#pragma unroll
      for (int k = 0; k < NDMA; ++k) dma(i, k);
    }
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        if constexpr (MODE != 0 && SPREAD) {
          constexpr int every = (4 * TAPS) / (NDMA < 4 * TAPS ? NDMA : 4 * TAPS);
          if ((t * 4 + st) % every == 0 && (t * 4 + st) / every < NDMA) dma(i, (t * 4 + st) / every);
        }
        if constexpr (LDS_READS) {
          if (st == 0)
            asm volatile("ds_read_b32 %0, %5 offset:0\n\tds_read_b32 %1, %5 offset:128\n\tds_read_b128 %2, %4 offset:0\n\tds_read_b128 %3, %4 offset:1024\n\ts_waitcnt lgkmcnt(4)"
                         : "=&v"(b0), "=&v"(b1), "=&v"(a0), "=&v"(a1) : "v"(addr_a), "v"(addr_b) : "memory");
          else
            asm volatile("ds_read_b32 %0, %2 offset:256\n\tds_read_b32 %1, %2 offset:384\n\ts_waitcnt lgkmcnt(2)" : "=&v"(b0), "=&v"(b1) : "v"(addr_b) : "memory");
        }
        asm volatile(
            "v_mfma_f32_32x32x2_f32 %0, %4, %6, %0\n\t"
            "v_mfma_f32_32x32x2_f32 %1, %4, %7, %1\n\t"
            "v_mfma_f32_32x32x2_f32 %2, %5, %6, %2\n\t"
            "v_mfma_f32_32x32x2_f32 %3, %5, %7, %3"
            : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
            : "v"(a0[st]), "v"(a1[st]), "v"(b0), "v"(b1));
      }
    }
    __syncthreads();
  }
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][15];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
#endif
}

// Block turnover: the chunk loop above (operand reads + six spread buffer_load ... lds per chunk) as TILES of 64 chunks, as the dominant layer
// runs them (Cin 512).  MODE_T 0: one launch of rounds x 1024 blocks, one tile each, no output (block exit + dispatch of the next one +
// its first staging burst per tile); 1: the same with the tile's epilogue, 64 buffer_store_dword per lane; 2: PERSISTENT blocks, 1024 of them,
// each looping over its tiles with the same epilogue and the next tile's first staging burst issued before it.
template <int MODE_T>
__global__ __launch_bounds__(256) void tile_loop_kernel(float* out, int tiles_per_block, const float* src, unsigned src_mask, float* sink) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) float lds[9216];
  const int lane = threadIdx.x & 63, i32 = lane & 31, h = lane >> 5, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row = wave * 64 + i32;
  const unsigned lds0 = (unsigned)(uintptr_t)lds;
  const unsigned addr_a = lds0 + (row * 2 + (h ^ ((row >> 3) & 1))) * 16;
  const unsigned addr_b = lds0 + (2112 + 4 * h * 64 + i32) * 4;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x40000000, 0x00020000);
  const __amdgpu_buffer_rsrc_t osrd = __builtin_amdgcn_make_buffer_rsrc((void*)sink, 0, 0x40000000, 0x00020000);
  int voff[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) voff[k] = (wave * 6 + k) * 1024 + lane * 16;
  auto dma = [&](unsigned chunk_id, int k) {
    float* dst = lds + 3072 + ((wave * 6 + k) * 256) % 6144;
    const unsigned chunk_bytes = (chunk_id * 24576u) & 0x3ff00000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lptr_t)dst, 16, voff[k], (int)chunk_bytes, 0, 0);
  };
  for (int i = threadIdx.x; i < 3072; i += 256) lds[i] = 1.0f / 1024.0f;
  for (int t = 0; t < tiles_per_block; ++t) {
    const unsigned tile = blockIdx.x + (unsigned)t * gridDim.x;
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    f32x4 a0 = {0.125f, 0.25f, 0.5f, 1.f}, a1 = a0;
    float b0 = 1.0f / 1024.0f, b1 = b0;
    if (MODE_T != 2 || t == 0) {
#pragma unroll
      for (int k = 0; k < 6; ++k) dma(tile * 64u, k);          // the tile's first chunk: a burst, and nothing to overlap it with
    }
    __syncthreads();
    for (int i = 0; i < 64; ++i) {
#pragma unroll
      for (int tp = 0; tp < 5; ++tp) {
#pragma unroll
        for (int st = 0; st < 4; ++st) {
          if ((tp * 4 + st) % 2 == 0 && (tp * 4 + st) / 2 < 6) dma(tile * 64u + i + 1, (tp * 4 + st) / 2);
          if (st == 0)
            asm volatile("ds_read_b32 %0, %5 offset:0\n\tds_read_b32 %1, %5 offset:128\n\tds_read_b128 %2, %4 offset:0\n\tds_read_b128 %3, %4 offset:1024\n\ts_waitcnt lgkmcnt(4)"
                         : "=&v"(b0), "=&v"(b1), "=&v"(a0), "=&v"(a1) : "v"(addr_a), "v"(addr_b) : "memory");
          else
            asm volatile("ds_read_b32 %0, %2 offset:256\n\tds_read_b32 %1, %2 offset:384\n\ts_waitcnt lgkmcnt(2)" : "=&v"(b0), "=&v"(b1) : "v"(addr_b) : "memory");
          asm volatile(
              "v_mfma_f32_32x32x2_f32 %0, %4, %6, %0\n\t"
              "v_mfma_f32_32x32x2_f32 %1, %4, %7, %1\n\t"
              "v_mfma_f32_32x32x2_f32 %2, %5, %6, %2\n\t"
              "v_mfma_f32_32x32x2_f32 %3, %5, %7, %3"
              : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
              : "v"(a0[st]), "v"(a1[st]), "v"(b0), "v"(b1));
        }
      }
      __syncthreads();
    }
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
    if (MODE_T >= 1) {
      // epilogue of the tile: 64 dword stores per lane, 128 contiguous bytes per half-wave and row, rows 4 KiB apart (Cout = 1024)
      const unsigned tile_base = ((tile & 4095u) * 65536u * 4u) & 0x3fc00000u;      // 4-MiB steps: the lane offsets (up to 1 MiB) stay inside the 1-GiB sink
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = wave * 64 + (j >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[j][r] + 1.0f), osrd, (rr * 1024 + (j & 1) * 32 + i32) * 4, (int)tile_base, 0);
        }
    } else {
      float sacc = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) sacc += acc[j][0] + acc[j][15];
      if (sacc == 12345.678f) out[threadIdx.x] = sacc;
    }
  }
#endif
}

static void run_tiles(const char* name, int mode, int cus, float* out, const float* src, unsigned src_mask, float* sink) {
  const int rounds = 8, slots = cus * 4;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0, 0));
    if (mode == 0) hipLaunchKernelGGL(tile_loop_kernel<0>, dim3(slots * rounds), dim3(256), 0, 0, out, 1, src, src_mask, sink);
    else if (mode == 1) hipLaunchKernelGGL(tile_loop_kernel<1>, dim3(slots * rounds), dim3(256), 0, 0, out, 1, src, src_mask, sink);
    else hipLaunchKernelGGL(tile_loop_kernel<2>, dim3(slots), dim3(256), 0, 0, out, rounds, src, src_mask, sink);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  const double tf = (double)slots * rounds * 4 * 64 * 80 * 4096.0 / (best * 1e-3) / 1e12;
  printf("%-60s %8.3f ms  %7.2f TFLOP/s  %.4f of 157.3\n", name, best, tf, tf / 157.3);
}

template <typename K>
static void run_chunk(const char* name, K kernel, int cus, int blocks_per_cu, float* out, const float* src, unsigned src_mask, int mfmas_per_chunk = 80) {
  const int blocks = cus * blocks_per_cu;
  const int iters = 16000 / blocks_per_cu;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, iters / 10, src, src_mask);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, iters, src, src_mask);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double tf = (double)blocks * 4 * iters * mfmas_per_chunk * 4096.0 / (best * 1e-3) / 1e12;
  printf("%-44s blocks/CU %d  %8.3f ms  %7.2f TFLOP/s  %.4f of 157.3\n", name, blocks_per_cu, best, tf, tf / 157.3);
}

template <typename K>
static void run(const char* name, K kernel, int nacc, double flop_per_mfma, int cus, int waves_per_simd, float* out) {
  const int blocks = cus * waves_per_simd;            // 256 threads = one wave per SIMD of a CU
  const int iters = 40000 / waves_per_simd;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, iters / 10);   // warm-up
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, iters);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double flop = (double)blocks * 4 * iters * 32 * flop_per_mfma;
  const double tf = flop / (best * 1e-3) / 1e12;
  printf("%-24s acc tiles %d  waves/SIMD %d  %8.3f ms  %7.2f TFLOP/s  %.4f of 157.3\n", name, nacc, waves_per_simd, best, tf, tf / 157.3);
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  printf("%s: %d CUs, clock %d MHz\n", p.gcnArchName, cus, p.clockRate / 1000);
  float* out;
  CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(float)));
  const int wps[] = {1, 2, 4, 8};
  for (int w : wps) {
    run("v_mfma_f32_32x32x2_f32", mfma32_kernel<1>, 1, 4096.0, cus, w, out);
    run("v_mfma_f32_32x32x2_f32", mfma32_kernel<2>, 2, 4096.0, cus, w, out);
    run("v_mfma_f32_32x32x2_f32", mfma32_kernel<4>, 4, 4096.0, cus, w, out);
  }
  for (int w : wps) {
    run("v_mfma_f32_16x16x4_f32", mfma16_kernel<1>, 1, 2048.0, cus, w, out);
    run("v_mfma_f32_16x16x4_f32", mfma16_kernel<4>, 4, 2048.0, cus, w, out);
  }
  float* src;
  const size_t src_floats = (size_t)1 << 28;                  // 1 GiB
  CHECK(hipMalloc(&src, src_floats * sizeof(float)));
  CHECK(hipMemset(src, 0, src_floats * sizeof(float)));
  const unsigned mask = (unsigned)(src_floats - 1) & ~3u;
  for (int w : {1, 4}) {
    run_chunk("80 MFMAs + barrier", chunk_loop_kernel<false, 0, 0, 16, false>, cus, w, out, src, mask);
    run_chunk("  + operand reads", chunk_loop_kernel<true, 0, 0, 16, false>, cus, w, out, src, mask);
    run_chunk("  + 6 global_load_lds x 16 B", chunk_loop_kernel<true, 1, 6, 16, false>, cus, w, out, src, mask);
    run_chunk("  + 6 buffer_load lds x 16 B", chunk_loop_kernel<true, 2, 6, 16, false>, cus, w, out, src, mask);
    run_chunk("  + 3 buffer_load lds x 16 B", chunk_loop_kernel<true, 2, 3, 16, false>, cus, w, out, src, mask);
    run_chunk("  + 12 buffer_load lds x 16 B", chunk_loop_kernel<true, 2, 12, 16, false>, cus, w, out, src, mask);
    run_chunk("  + 6 buffer_load lds x 4 B", chunk_loop_kernel<true, 2, 6, 4, false>, cus, w, out, src, mask);
    run_chunk("  + 24 buffer_load lds x 4 B", chunk_loop_kernel<true, 2, 24, 4, false>, cus, w, out, src, mask);
    run_chunk("  + 6 buffer_load lds x 16 B, spread", chunk_loop_kernel<true, 2, 6, 16, true>, cus, w, out, src, mask);
    run_chunk("  + 6 buffer x 16 B, no operand reads", chunk_loop_kernel<false, 2, 6, 16, false>, cus, w, out, src, mask);
  }
  // the short-tap chunks of the stride-2 data gradient's two phases (48 / 32 MFMAs per barrier), reads + 5 spread staging pieces, as the kernel runs them
  for (int w : {1, 2, 4}) {
    run_chunk("48 MFMAs + barrier + reads + 5 spread pieces", chunk_loop_kernel<true, 2, 5, 16, true, 3>, cus, w, out, src, mask, 48);
    run_chunk("32 MFMAs + barrier + reads + 4 spread pieces", chunk_loop_kernel<true, 2, 4, 16, true, 2>, cus, w, out, src, mask, 32);
    run_chunk("32 MFMAs + barrier only", chunk_loop_kernel<false, 0, 0, 16, false, 2>, cus, w, out, src, mask, 32);
  }
  float* sink;
  CHECK(hipMalloc(&sink, (size_t)1 << 30));
  run_tiles("tiles of 64 chunks, a block per tile, no output", 0, cus, out, src, mask, sink);
  run_tiles("tiles of 64 chunks, a block per tile, 64 stores per lane", 1, cus, out, src, mask, sink);
  run_tiles("tiles of 64 chunks, persistent blocks, 64 stores per lane", 2, cus, out, src, mask, sink);
  CHECK(hipFree(sink));
  CHECK(hipFree(src));
  CHECK(hipFree(out));
  return 0;
}
