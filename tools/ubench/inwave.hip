// Micro-benchmark: ONE wave per SIMD issuing [MFMA, K independent VALU instructions] repeatedly: how many vector
// instructions hide in the shadow of the wave's own 32x32x16 MFMA?  kind 0: v_fma_f32, 1: v_exp_f32, 2: ds_read_b128.
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int K, int KIND>
__global__ __launch_bounds__(256) void kin(float* out, int iters) {
  __shared__ __attribute__((aligned(1024))) char smem[32768];
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.37f * (threadIdx.x % 7) - 1.f); b[i] = (__bf16)(0.11f * (threadIdx.x % 5) - 0.2f); }
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = 0.001f * threadIdx.x + i;
  bf16x8 d[8];
  const unsigned la = (threadIdx.x & 63) * 16;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[j & 3]) : "v"(a), "v"(b));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(j * K + k) & 7]) : "v"(0.999f), "v"(0.001f));
        else if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x[(j * K + k) & 7]));
        else asm volatile("ds_read_b128 %0, %1" : "=v"(d[(j * K + k) & 7]) : "v"(la + ((j * K + k) & 15) * 1024));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += x[i] + (float)d[i][0];
  if (s == 123.f) out[threadIdx.x] = s;
}
#define C(K, KIND) case (KIND) * 100 + (K): hipLaunchKernelGGL((kin<K, KIND>), dim3(256), dim3(256), 0, s, out, iters); break;
extern "C" int run_in(int k, int kind, float* out, int iters, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  switch (kind * 100 + k) {
    C(0, 0) C(1, 0) C(2, 0) C(3, 0) C(4, 0) C(5, 0) C(6, 0) C(8, 0) C(12, 0) C(1, 1) C(2, 1) C(3, 1) C(4, 1) C(1, 2) C(2, 2) C(4, 2)
    default: return -1;
  }
  return (int)hipGetLastError();
}
