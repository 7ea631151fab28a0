// gemm_modes.hip -- the 8-wave GEMM kernels (gemm8.h) of the two opt-in arithmetic modes: split-bf16 operand pairs (precise mode,
// k_segs = 2 | 3) and OCP e4m3 operands (LX_OPERANDS_FP8, BASELINE configs[4]'s lossy GEMM option).
#include "gemm8.h"

void lx_gemm8_launch_split(int bm, const GemmArgs& a, hipStream_t s) {
  const int t = a.tile_start[a.n];
  if (bm == 256) hipLaunchKernelGGL(lx_gemm_split_kernel<256>, dim3(t), dim3(NTHREADS), 0, s, a);
  else hipLaunchKernelGGL(lx_gemm_split_kernel<128>, dim3(t), dim3(NTHREADS), 0, s, a);
}
void lx_gemm8_launch_fp8(int bm, const GemmArgs& a, hipStream_t s) {
  const int t = a.tile_start[a.n];
  if (bm == 256) hipLaunchKernelGGL(lx_gemm_fp8_kernel<256>, dim3(t), dim3(NTHREADS), 0, s, a);
  else hipLaunchKernelGGL(lx_gemm_fp8_kernel<128>, dim3(t), dim3(NTHREADS), 0, s, a);
}
