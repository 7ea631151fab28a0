// gemm_f16.hip -- the 8-wave GEMM kernels (gemm8.h) with fp16 operands (LX_OPERANDS_F16): v_mfma_f32_32x32x16_f16, fp16 16-bit stores.
#include "gemm8.h"

void lx_gemm8_launch_f16(int bm, const GemmArgs& a, hipStream_t s) {
  const int t = a.tile_start[a.n];
  if (bm == 256) hipLaunchKernelGGL((lx_gemm_kernel<256, true>), dim3(t), dim3(NTHREADS), 0, s, a);
  else hipLaunchKernelGGL((lx_gemm_kernel<128, true>), dim3(t), dim3(NTHREADS), 0, s, a);
}
void lx_gemm8_mixed_launch_f16(const GemmArgs& big, const GemmArgs& tail, int n_big_pad, hipStream_t s) {
  hipLaunchKernelGGL(lx_gemm_mixed_kernel<true>, dim3(n_big_pad + tail.tile_start[tail.n]), dim3(NTHREADS), 0, s, big, tail, n_big_pad);
}
