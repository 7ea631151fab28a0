// Smoke test of v_mfma_f32_32x32x64_f8f6f4 (fp8 e4m3 x fp8 e4m3): each lane gives 32 bytes of A (row = lane % 32, k-group =
// lane / 32) and 32 bytes of B (column = lane % 32, same k-group). Checks D = A . B^T for the "lane group g holds 32 k values,
// the same (g, byte) position on both operands is the same k" convention and reports what the scale arguments (E8M0 exponents of
// the MX block scales; both literal 0 selects the unscaled instruction) do.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ void kf8(const uint8_t* A, const uint8_t* B, float* D, int mode, int sa, int sb) {   // A,B: [32][64] bytes (row-major, k contiguous)
  const int lane = threadIdx.x, row = lane & 31, g = lane >> 5;
  v8i a, b;
  for (int i = 0; i < 8; ++i) { a[i] = ((const int*)(A + row * 64 + g * 32))[i]; b[i] = ((const int*)(B + row * 64 + g * 32))[i]; }
  v16f c = {};
  if (mode == 0) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);
  else if (mode == 1) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 127, 0, 127);
  else c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);      // runtime (VGPR) scales, byte 0 of each
  // D[i][j]: j = lane % 32 (B row), i = 8*(r/4) + 4*(lane/32) + r%4 (A row)
  for (int r = 0; r < 16; ++r) D[(8 * (r >> 2) + 4 * g + (r & 3)) * 32 + row] = c[r];
}
extern "C" int run_f8(const void* A, const void* B, float* D, int mode, int sa, int sb, void* stream) {
  hipLaunchKernelGGL(kf8, dim3(1), dim3(64), 0, (hipStream_t)stream, (const uint8_t*)A, (const uint8_t*)B, D, mode, sa, sb);
  return (int)hipGetLastError();
}
