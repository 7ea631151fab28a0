// Micro-benchmark: wave-specialised variant. 12 waves: waves 0-7 do ds_read + MFMA (the GEMM's per-K-tile work), waves
// 8-11 only issue the LDS-DMA (16 pieces of 1 KiB each per K tile) DEPTH tiles ahead. One barrier per K tile for all.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int CM>   // consumer mode: bit0 ds_reads, bit1 MFMA
//   // tiles of DMA in flight (ring of DEPTH+1 slots of 32 KiB halves... here: BK=64 -> 64 KiB per tile, 2 slots => DEPTH 1)
__global__ __launch_bounds__(768, 3) void kws(const uint16_t* A, const uint16_t* W, float* out, int iters, int a_stride, int w_stride) {
  __shared__ __attribute__((aligned(1024))) char smem[131072];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave >= 8) {
    const int lw = wave - 8;
    const uint16_t* a = A + (size_t)(blockIdx.x % 10) * a_stride + lane * 8;
    const uint16_t* w = W + (size_t)(blockIdx.x / 10) * w_stride + lane * 8;
    auto issue = [&](int it) {
      char* base = smem + (it & 1) * 65536;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = j * 4 + lw;   // 32 chunks of 1 KiB per operand
        __builtin_amdgcn_global_load_lds((gptr_t)(a + (size_t)it * 16384 + c * 512), (lptr_t)(base + c * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(w + (size_t)it * 16384 + c * 512), (lptr_t)(base + 32768 + c * 1024), 16, 0, 0);
      }
    };
    issue(0);
    for (int it = 0; it < iters; ++it) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile `it` landed
      __builtin_amdgcn_s_barrier();                        // consumers done with tile it-1, may start tile it
      if (it + 1 < iters) issue(it + 1);                   // into the slot of tile it-1
    }
    return;
  }
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 f[6];
  for (int i = 0; i < 6; ++i) f[i] = bf16x8{};
  const int roff = (lane & 31) * 128 + ((lane >> 5) ^ ((lane >> 1) & 7)) * 16;
  for (int it = 0; it < iters; ++it) {
    __builtin_amdgcn_s_barrier();
    const char* sb = smem + (it & 1) * 65536;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (CM & 1) {
#pragma unroll
        for (int i = 0; i < 6; ++i) f[i] = *(const bf16x8*)(sb + (i & 3) * 4096 + (i >> 2) * 32768 + roff + ks * 32);
      }
      if (CM & 2) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[j * 4 + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[j], f[2 + i], acc[j * 4 + i], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < 6; ++i) asm volatile("" :: "v"(f[i]));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123.456f) out[threadIdx.x] = s;
}

extern "C" int run_ws(int cm, const void* A, const void* W, float* out, int grid, int iters, int a_stride, int w_stride, void* stream) {
  if (cm == 1) hipLaunchKernelGGL(kws<1>, dim3(grid), dim3(768), 0, (hipStream_t)stream, (const uint16_t*)A, (const uint16_t*)W, out, iters, a_stride, w_stride);
  else if (cm == 2) hipLaunchKernelGGL(kws<2>, dim3(grid), dim3(768), 0, (hipStream_t)stream, (const uint16_t*)A, (const uint16_t*)W, out, iters, a_stride, w_stride);
  else hipLaunchKernelGGL(kws<3>, dim3(grid), dim3(768), 0, (hipStream_t)stream, (const uint16_t*)A, (const uint16_t*)W, out, iters, a_stride, w_stride);
  return (int)hipGetLastError();
}
