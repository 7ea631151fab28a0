// What v_cvt_pk_u8_f32 does with fractions, negatives, large values and NaN (the probability-byte producer of lx_attn_fp8_pipe_kernel's
// log-linear form relies on: saturation at 0 and 255, and needs to know the rounding to place its zero point).
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void kcvt(const float* x, uint32_t* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(x[i], 0, 0u) | (__builtin_amdgcn_cvt_pk_u8_f32(x[i], 2, 0xffffffffu) & 0x00ff0000u);
}
extern "C" int run_cvt(const float* x, uint32_t* out, int n, void* stream) {
  hipLaunchKernelGGL(kcvt, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, x, out, n);
  return (int)hipGetLastError();
}
