// gemm4_f16.hip -- lx_gemm4_kernel (gemm4.h) with fp16 operands (LX_OPERANDS_F16): v_mfma_f32_16x16x32_f16, fp16 16-bit stores.
#include "gemm4.h"

void lx_gemm4_launch_f16(const GemmArgs& a, unsigned grid, int sk_full, int sk_parts, float* slots, int* flags, int* err, hipStream_t s, int np) {
  if (np == 3) hipLaunchKernelGGL((lx_gemm4_kernel<false, true, 3>), dim3(grid), dim3(G4_THREADS), 0, s, a, sk_full, sk_parts, slots, flags, err);
  else hipLaunchKernelGGL((lx_gemm4_kernel<false, true>), dim3(grid), dim3(G4_THREADS), 0, s, a, sk_full, sk_parts, slots, flags, err);
}
