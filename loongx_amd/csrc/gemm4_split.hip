// gemm4_split.hip -- lx_gemm4_kernel (gemm4.h) on split-bf16 operand pairs (precise mode: k_segs = 2 | 3, LX_EPI_SPLIT_BF16).
#include "gemm4.h"

void lx_gemm4_launch_split(const GemmArgs& a, unsigned grid, int sk_full, int sk_parts, float* slots, int* flags, int* err, hipStream_t s) {
  hipLaunchKernelGGL((lx_gemm4_kernel<true, false>), dim3(grid), dim3(G4_THREADS), 0, s, a, sk_full, sk_parts, slots, flags, err);
}
