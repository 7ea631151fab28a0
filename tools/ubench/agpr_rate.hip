// Micro-benchmark: does keeping the MFMA accumulators in AGPRs let LDS-DMA overlap with MFMA? Same loop as dma_rate
// mode 5 (8 DMA pieces + 32 MFMA per wave per K tile); MFMA issued through inline asm with "a" (AGPR) accumulators.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <bool AGPR, bool DMA>
__global__ __launch_bounds__(512) void ka(const uint16_t* A, const uint16_t* W, float* out, int iters, int a_stride, int w_stride) {
  __shared__ __attribute__((aligned(1024))) char smem[131072];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint16_t* a = A + (size_t)(blockIdx.x % 10) * a_stride + wave * 512 + lane * 8;
  const uint16_t* w = W + (size_t)(blockIdx.x / 10) * w_stride + wave * 512 + lane * 8;
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 f[6];
  for (int i = 0; i < 6; ++i) { f[i] = bf16x8{}; asm volatile("" : "+v"(f[i])); }
  for (int it = 0; it < iters; ++it) {
    char* base = smem + (it & 1) * 65536;
    if (DMA) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        __builtin_amdgcn_global_load_lds((gptr_t)(a + (size_t)it * 16384 + j * 4096), (lptr_t)(base + (j * 8 + wave) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(w + (size_t)it * 16384 + j * 4096), (lptr_t)(base + 32768 + (j * 8 + wave) * 1024), 16, 0, 0);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[j * 4 + i]) : "v"(f[j]), "v"(f[2 + i]));
          else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[j * 4 + i]) : "v"(f[j]), "v"(f[2 + i]));
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  asm volatile("s_nop 15\n s_nop 15" ::: "memory");
  float s = 0;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123.456f) out[threadIdx.x] = s;
}
extern "C" int run_a(int mode, const void* A, const void* W, float* out, int grid, int iters, int a_stride, int w_stride, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const uint16_t* a = (const uint16_t*)A; const uint16_t* w = (const uint16_t*)W;
  switch (mode) {
    case 0: hipLaunchKernelGGL((ka<false, false>), dim3(grid), dim3(512), 0, s, a, w, out, iters, a_stride, w_stride); break;
    case 1: hipLaunchKernelGGL((ka<false, true>), dim3(grid), dim3(512), 0, s, a, w, out, iters, a_stride, w_stride); break;
    case 2: hipLaunchKernelGGL((ka<true, false>), dim3(grid), dim3(512), 0, s, a, w, out, iters, a_stride, w_stride); break;
    default: hipLaunchKernelGGL((ka<true, true>), dim3(grid), dim3(512), 0, s, a, w, out, iters, a_stride, w_stride); break;
  }
  return (int)hipGetLastError();
}
