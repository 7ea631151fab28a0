// common.h -- shared device helpers for the gfx950 (CDNA4) kernels. MI355X only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/lx.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define LX_WAVE 64

void lx_set_error(const char* fmt, ...);
#define LX_CHECK_ARG(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      lx_set_error(__VA_ARGS__);           \
      return LX_ERR_INVALID;               \
    }                                      \
  } while (0)
#define LX_LAUNCH_CHECK(name)                                                   \
  do {                                                                          \
    hipError_t e_ = hipGetLastError();                                          \
    if (e_ != hipSuccess) {                                                     \
      lx_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));       \
      return LX_ERR_LAUNCH;                                                     \
    }                                                                           \
  } while (0)

__device__ __forceinline__ float bf16_to_f32(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// fp32 -> bf16, round-to-nearest-even, through the native __bf16 cast: hipcc lowers it to the gfx950 hardware
// converts (v_cvt_pk_bf16_f32) -- branch-free, NaN-safe. (A hand-rolled bit trick with a NaN branch compiles to
// divergent exec-mask code per element and was 10x slower inside the attention loop.)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
// Buffer-addressed LDS-DMA: 16 B per lane straight into LDS, source = SRSRC (wave-uniform origin) + per-lane byte offset +
// scalar offset. Device pass only: the host pass of hipcc must still be able to emit the kernel stubs.
typedef __attribute__((address_space(3))) void* lx_lds_ptr_t;
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t lx_rsrc_t;
__device__ __forceinline__ lx_rsrc_t lx_make_rsrc(const void* p) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7fffffff, 0x00020000); }
__device__ __forceinline__ void lx_buf_to_lds(lx_rsrc_t r, lx_lds_ptr_t l, uint32_t voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, l, 16, voff, soff, 0, 0);
}
#else
typedef int lx_rsrc_t;
__host__ __device__ inline lx_rsrc_t lx_make_rsrc(const void*) { return 0; }
__host__ __device__ inline void lx_buf_to_lds(lx_rsrc_t, lx_lds_ptr_t, uint32_t, int) {}
#endif

__device__ __forceinline__ uint16_t f32_to_bf16(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
// ---- IEEE fp16 operand images (LX_OPERANDS_F16: 11 significand bits on the matrix pipe's A operand instead of bf16's 8) -------------
// fp32 pair -> fp16 pair, round-to-nearest-even (v_cvt_pk_f16_f32), SATURATED to +-65504 (v_med3_f32: fp16 has 5 exponent bits; the
// reference clips its fp16 residual stream the same way, block.py:275-276, 336-337). `mx` collects max |x| of everything converted
// (v_max3_f32 with |.| modifiers: half an instruction per element) so that the producer can REPORT a saturation instead of hiding it.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#define LX_F16_MAX 65504.0f
__device__ __forceinline__ uint32_t pack_f16x2_sat(float lo, float hi, float& mx) {
  mx = fmaxf(fmaxf(fabsf(lo), fabsf(hi)), mx);
  const f16x2 v = {(_Float16)__builtin_amdgcn_fmed3f(lo, -LX_F16_MAX, LX_F16_MAX), (_Float16)__builtin_amdgcn_fmed3f(hi, -LX_F16_MAX, LX_F16_MAX)};
  return __builtin_bit_cast(uint32_t, v);
}
// the 16-bit store of a producer whose consumer is a GEMM: bf16 (the default operand format) or fp16
template <bool F16>
__device__ __forceinline__ uint32_t pack_op16x2(float lo, float hi, float& mx) {
  if constexpr (F16) return pack_f16x2_sat(lo, hi, mx);
  else return pack_bf16x2(lo, hi);
}
// a saturated conversion happened somewhere in this wave's share: count it in the caller's overflow word (NULL: not asked for).
// One atomic per WAVE that saw one, none otherwise -- the word says "how many producer waves clipped", 0 = the images are exact roundings.
__device__ __forceinline__ void report_f16_overflow(float mx, int* ovf) {
  if (ovf != nullptr && __builtin_amdgcn_ballot_w64(mx > LX_F16_MAX) != 0 && (threadIdx.x & 63) == 0) atomicAdd(ovf, 1);
}

__device__ __forceinline__ float gelu_tanh(float x) {
  // 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))) == x * sigmoid(2u)
  // v_exp_f32 + v_rcp_f32 (1 ulp): an IEEE division here compiles to v_div_scale / v_rcp / 4 x v_fma / v_div_fmas / v_div_fixup
  // per element -- ~10 VALU instructions x 128 outputs per lane, 4.4 us per tile round of the MLP GEMMs (measured: 15 us of the
  // 309-us fused single-block launch). The output is rounded to bf16 (or added to an fp32 residual at 2^-8 relative weight).
  // Constants folded so that exp(-2u) is ONE v_exp_f32 of x * (C1 + C2 x^2): C1 = -2 sqrt(2/pi) log2(e), C2 = 0.044715 C1.
  const float p = x * __builtin_fmaf(x * x, -0.10294324221f, -2.30220819813f);
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(p));
}
// Two at a time: the polynomial part on the packed-fp32 pipe (v_pk_mul / v_pk_fma / v_pk_add: two elements per instruction),
// only the exp2 and the reciprocal per element. Same arithmetic as gelu_tanh, element by element.
__device__ __forceinline__ f32x2 gelu_tanh2(f32x2 x) {
  const f32x2 c1 = {-2.30220819813f, -2.30220819813f}, c2 = {-0.10294324221f, -0.10294324221f}, one = {1.0f, 1.0f};
  const f32x2 p = x * __builtin_elementwise_fma(x * x, c2, c1);
  const f32x2 d = f32x2{__builtin_amdgcn_exp2f(p[0]), __builtin_amdgcn_exp2f(p[1])} + one;
  return x * f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}
__device__ __forceinline__ f32x4 gelu_tanh4(f32x4 v) {
  const f32x2 a = gelu_tanh2(f32x2{v[0], v[1]}), b = gelu_tanh2(f32x2{v[2], v[3]});
  return f32x4{a[0], a[1], b[0], b[1]};
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// sum over the 32 lanes of each half wave (lanes 0-31 / 32-63) without touching LDS: quad, half-row and row mirrors, then lane 15 of rows
// 0 / 2 into rows 1 / 3 (DPP row_bcast:15). The totals stand in lanes 16-31 and 48-63. (__shfl_xor is ds_bpermute: five dependent LDS
// round trips per value -- 80 per 32 x 32 block of sigmoids, which is where the gate's second GEMM spent most of its time.)
__device__ __forceinline__ float half_wave_sum_hi16(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));   // row_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));   // row_bcast:15 into rows 1, 3
  return v;
}
// the sum over each row of 16 lanes, in every lane of the row: BIT-identical to v += __shfl_xor(v, 1); ... 2; ... 4; ... 8 (after the two
// quad steps a quad's lanes hold the same bits, so the mirror partners 7 - i / 15 - i carry what the xor partners i ^ 4 / i ^ 8 do), on
// DPP instead of four dependent ds_bpermute round trips
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));   // row_mirror
  return v;
}
// the sum over all 64 lanes as a wave-uniform value: the half-wave sums above, lane 31's into rows 2 / 3 (row_bcast:31), lane 63 read back
// through an SGPR. Seven dependent vector instructions against wave_sum's six ds_bpermute round trips (another summation tree: results
// differ from wave_sum's in the last bit).
__device__ __forceinline__ float wave_total(float v) {
  v = half_wave_sum_hi16(v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
