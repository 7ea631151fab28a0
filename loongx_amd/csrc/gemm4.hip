// gemm4.hip -- lx_gemm4_kernel (gemm4.h) with bf16 operands: the instantiation the default denoise step runs.
#include "gemm4.h"

void lx_gemm4_launch_bf16(const GemmArgs& a, unsigned grid, int sk_full, int sk_parts, float* slots, int* flags, int* err, hipStream_t s, int np) {
  if (np == 3) hipLaunchKernelGGL((lx_gemm4_kernel<false, false, 3>), dim3(grid), dim3(G4_THREADS), 0, s, a, sk_full, sk_parts, slots, flags, err);
  else hipLaunchKernelGGL((lx_gemm4_kernel<false, false>), dim3(grid), dim3(G4_THREADS), 0, s, a, sk_full, sk_parts, slots, flags, err);
}

#ifdef LX_G4_PROBE
extern "C" int lx_g4_probe_read(unsigned long long* host, size_t n_u64) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(lx_g4_probe_buf), n_u64 * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif
