// Micro-benchmark: cycles per v_mfma_f32_32x32x16_bf16 for ONE wave per SIMD issuing a stream that rotates over NACC
// independent accumulators (dependency distance NACC), and for two waves per SIMD.
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int NACC>
__global__ __launch_bounds__(512) void kchain(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a = {}, b = {};
  asm volatile("" : "+v"(a), "+v"(b));
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 32; ++j) acc[j % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j % NACC], 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (threadIdx.x == 0) out[blockIdx.x] = (float)(t1 - t0) / (32.f * iters);
  if (s == 123.f) out[1000 + threadIdx.x] = s;
}
extern "C" int run_chain(int nacc, int threads, float* out, int iters, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  switch (nacc) {
    case 1: hipLaunchKernelGGL(kchain<1>, dim3(256), dim3(threads), 0, s, out, iters); break;
    case 2: hipLaunchKernelGGL(kchain<2>, dim3(256), dim3(threads), 0, s, out, iters); break;
    case 3: hipLaunchKernelGGL(kchain<3>, dim3(256), dim3(threads), 0, s, out, iters); break;
    case 4: hipLaunchKernelGGL(kchain<4>, dim3(256), dim3(threads), 0, s, out, iters); break;
    case 8: hipLaunchKernelGGL(kchain<8>, dim3(256), dim3(threads), 0, s, out, iters); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}
