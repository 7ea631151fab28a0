// Micro-benchmark: A via LDS-DMA (4 pieces/wave/K-tile) + W fragments by plain coalesced global_load_dwordx4 straight to
// VGPRs (8 per wave per K tile, consumed one K tile later from a register ring) + 16 ds_reads + 32 MFMA per wave.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int MODE>  // 0: A-DMA + W-reg + ds_read + MFMA ; 1: same without MFMA ; 2: W-reg loads only
__global__ __launch_bounds__(512) void kw(const uint16_t* A, const uint16_t* W, float* out, int iters, int a_stride, int w_stride) {
  __shared__ __attribute__((aligned(1024))) char smem[65536];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wn = wave & 3;
  const uint16_t* a = A + (size_t)(blockIdx.x % 10) * a_stride + wave * 512 + lane * 8;
  // W in fragment order: per K tile 32 KiB = [wn(4)][step(4)][j(2)] blocks of 1 KiB, lane-linear
  const uint16_t* w = W + (size_t)(blockIdx.x / 10) * w_stride + wn * 4096 + lane * 8;
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  u32x4 wr[8];
  for (int i = 0; i < 8; ++i) wr[i] = *(const u32x4*)(w + i * 512);
  bf16x8 xf[4];
  for (int i = 0; i < 4; ++i) xf[i] = bf16x8{};
  const int roff = (lane & 31) * 128 + ((lane >> 5) ^ ((lane >> 1) & 7)) * 16;
  for (int it = 0; it < iters; ++it) {
    char* base = smem + (it & 1) * 32768;
    if (MODE != 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_global_load_lds((gptr_t)(a + (size_t)it * 16384 + j * 4096), (lptr_t)(base + (j * 8 + wave) * 1024), 16, 0, 0);
    }
    const uint16_t* wn_ptr = w + (size_t)(it + 1 < iters ? it + 1 : it) * 16384;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 wf0 = __builtin_bit_cast(bf16x8, wr[ks * 2]), wf1 = __builtin_bit_cast(bf16x8, wr[ks * 2 + 1]);
      if (MODE != 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) xf[i] = *(const bf16x8*)(smem + ((it + 1) & 1) * 32768 + i * 4096 + roff + ks * 32);
      }
      if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf0, xf[i], acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[4 + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf1, xf[i], acc[4 + i], 0, 0, 0);
      } else {
        asm volatile("" :: "v"(wf0), "v"(wf1));
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(xf[i]));
      }
      // refill the two ring entries just consumed with the next K tile's fragments
      wr[ks * 2] = *(const u32x4*)(wn_ptr + (ks * 2) * 512);
      wr[ks * 2 + 1] = *(const u32x4*)(wn_ptr + (ks * 2 + 1) * 512);
    }
    __syncthreads();
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += __uint_as_float(wr[i][0]);
  if (s == 123.456f) out[threadIdx.x] = s;
}
extern "C" int run_w(int mode, const void* A, const void* W, float* out, int grid, int iters, int a_stride, int w_stride, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (mode == 0) hipLaunchKernelGGL(kw<0>, dim3(grid), dim3(512), 0, s, (const uint16_t*)A, (const uint16_t*)W, out, iters, a_stride, w_stride);
  else if (mode == 1) hipLaunchKernelGGL(kw<1>, dim3(grid), dim3(512), 0, s, (const uint16_t*)A, (const uint16_t*)W, out, iters, a_stride, w_stride);
  else hipLaunchKernelGGL(kw<2>, dim3(grid), dim3(512), 0, s, (const uint16_t*)A, (const uint16_t*)W, out, iters, a_stride, w_stride);
  return (int)hipGetLastError();
}
