// Fills every CU's LDS with a pattern (junk for the next kernel to find): a kernel that reads LDS it has not written shows up as
// run-to-run differences when the pattern changes. 2 x 256 workgroups of 160 KiB so that every CU is visited.
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ __launch_bounds__(512) void fill(uint32_t seed, uint32_t* sink) {
  __shared__ uint32_t lds[160 * 1024 / 4];
  uint32_t x = seed * 2654435761u + blockIdx.x * 40503u + threadIdx.x;
  for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 512) { x = x * 1664525u + 1013904223u; lds[i] = (seed == 0) ? 0u : ((seed == 1) ? 0x7fc07fc0u : x); }
  __syncthreads();
  if (lds[(threadIdx.x * 97) % (160 * 1024 / 4)] == 0x12345678u && sink) sink[0] = 1;
}
extern "C" int lds_fill(uint32_t seed, void* sink, void* stream) {
  hipLaunchKernelGGL(fill, dim3(1024), dim3(512), 0, (hipStream_t)stream, seed, (uint32_t*)sink);
  return (int)hipGetLastError();
}
