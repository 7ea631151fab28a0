// Micro-benchmark: MFMA stream on waves 0-3 (one per SIMD) and a VALU stream on waves 4-7 (their SIMD partners).
//   mode bit0: MFMA waves run, bit1: VALU waves run, bit2: MFMA waves at s_setprio 1, bit3: VALU stream is v_exp_f32
//   (transcendental, quarter rate) instead of v_fma_f32, bit4: roles swapped (VALU on the older waves 0-3)
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ __launch_bounds__(512) void kco(float* out, int iters, int mode, int nvalu) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool swapped = mode & 16;
  const bool mfma_wave = swapped ? wave >= 4 : wave < 4;
  if (mfma_wave) {
    if (!(mode & 1)) return;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.37f * (threadIdx.x % 7) - 1.f); b[i] = (__bf16)(0.11f * (threadIdx.x % 5) - 0.2f); }
    if (mode & 4) __builtin_amdgcn_s_setprio(1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 32; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j & 3], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.f) out[threadIdx.x] = s;
  } else {
    if (!(mode & 2)) return;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = 0.001f * threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
      for (int j = 0; j < nvalu / 8; ++j) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (mode & 8) x[i] = __builtin_amdgcn_exp2f(x[i]) * 0.5f;
          else x[i] = __builtin_fmaf(x[i], 0.999f, 0.001f);
        }
      }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 123.f) out[threadIdx.x] = s;
  }
}
extern "C" int run_co(float* out, int iters, int mode, int nvalu, void* stream) {
  hipLaunchKernelGGL(kco, dim3(256), dim3(512), 0, (hipStream_t)stream, out, iters, mode, nvalu);
  return (int)hipGetLastError();
}
