// gemm_common.h -- what the GEMM translation units share: tile geometry, the by-value launch argument block, the tile map, and the
// launchers each kernel family's translation unit exports to the planner in gemm.hip.
//
//   gemm.hip       planner (lx_gemm_bf16 / lx_gemm_bf16_ws) + the 8-wave bf16 kernels        (gemm8.h)
//   gemm_modes.hip the 8-wave split-bf16 ("precise") and e4m3 kernels                         (gemm8.h)
//   gemm_f16.hip   the 8-wave fp16-operand kernels                                            (gemm8.h)
//   gemm4.hip / gemm4_f16.hip / gemm4_split.hip   lx_gemm4_kernel (one wave per SIMD): bf16 / fp16 / split-bf16 operands   (gemm4.h)
// One kernel family per translation unit so that the build compiles them side by side (gemm.hip alone took 52 of the build's 72 s).
#pragma once
#include "common.h"
#include <type_traits>

constexpr int MAX_SUB = 2 * LX_GEMM_MAX_GROUP;   // a problem may be split into a 256-row-tile part and a 128-row-tile tail

struct GemmArgs {            // (global scope: the launchers below pass it between translation units)
  lx_gemm_desc p[MAX_SUB];
  int tile_start[MAX_SUB + 1];
  int m_base[MAX_SUB];       // row of the original problem at which this (sub)problem starts (for the gate batch index)
  int n;
};
static_assert(2 * sizeof(GemmArgs) + 16 <= 4096, "lx_gemm_mixed_kernel takes two plans by value: the kernarg segment is 4 KiB");

namespace {

constexpr int BN = 256;
constexpr int BK = 64;
constexpr int NTHREADS = 512;
#ifndef LX_GROUP_M              /* measurement builds (tools/gemm_energy.py): the height of an XCD's tile patch */
#define LX_GROUP_M 4
#endif
constexpr int GROUP_M = LX_GROUP_M;   // M-tile rows per column group of the tile order (4 / 8 / 16 measured identical in time, round 1; in joules, round 6)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int PAIR_AUX_SC1 = 16;                    // gfx940+ buffer cache policy: sc1 (agent scope)

__device__ __forceinline__ float clamp_e4m3(float x) { return fminf(fmaxf(x, -448.f), 448.f); }
__device__ __forceinline__ uint32_t pack_fp8x4(float a, float b, float c, float d) {
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(clamp_e4m3(a), clamp_e4m3(b), 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(clamp_e4m3(c), clamp_e4m3(d), w, true);
  return (uint32_t)w;
}

__device__ __forceinline__ int qkv_vt_interleave(int key) {  // within every 16 keys: [0-3, 8-11, 4-7, 12-15] (= rowops.hip)
  return (key & ~15) | (((key >> 2) & 1) << 3) | (((key >> 3) & 1) << 2) | (key & 3);
}

// lid (position in the launch's tile order) -> (sub)problem g and tile (tm, tn): 4-tile-tall column groups inside a problem.
// Two steps, with the descriptor copied BY VALUE in between: indexing the kernarg array lazily, field by field, made the tile
// start a chain of nine dependent scalar-load round trips before the first operand DMA could be issued.
__device__ __forceinline__ int tile_group(const GemmArgs& args, const int lid) {
  int g = 0;                                   // (entries past the last problem hold the total, which no lid reaches: no need for args.n)
#pragma unroll
  for (int i = 1; i < MAX_SUB; ++i)
    if (lid >= args.tile_start[i]) g = i;
  return g;
}

template <int BM>
__device__ __forceinline__ void tile_coords(const lx_gemm_desc& P, const int local, int& tm, int& tn) {
  const int tiles_m = (P.M + BM - 1) / BM;
  const int tiles_n = (P.N + BN - 1) / BN;
  const int gs = GROUP_M * tiles_n;
  const int gi = local / gs, in_g = local - gi * gs;
  const int first_m = gi * GROUP_M;
  const int gm = min(tiles_m - first_m, GROUP_M);
  tm = first_m + in_g % gm;
  tn = in_g / gm;
}

// lx_gemm4_kernel geometry the planner needs (gemm4.h holds the kernel)
constexpr int G4_THREADS = 256;
constexpr int SK_SLOT_FLOATS = 256 * 256;           // one parked 256 x 256 fp32 tile per split workgroup

}  // namespace

// operand / arithmetic variants of a launch (one per kernel instantiation family)
enum { LX_GV_BF16 = 0, LX_GV_F16 = 1, LX_GV_SPLIT = 2, LX_GV_FP8 = 3 };

// ---- launchers: each enqueues exactly one kernel on `s` and returns; the planner checks hipGetLastError() -------------------------
// 8-wave kernels: one plan of `bm`-row tiles (gemm.hip: LX_GV_BF16; gemm_f16.hip: LX_GV_F16; gemm_modes.hip: LX_GV_SPLIT, LX_GV_FP8)
void lx_gemm8_launch_bf16(int bm, const GemmArgs& a, hipStream_t s);
void lx_gemm8_launch_f16(int bm, const GemmArgs& a, hipStream_t s);
void lx_gemm8_launch_split(int bm, const GemmArgs& a, hipStream_t s);
void lx_gemm8_launch_fp8(int bm, const GemmArgs& a, hipStream_t s);
// 8-wave mixed-height plan in one grid: [256-row tiles | pad to 8 | 128-row tail tiles]
void lx_gemm8_mixed_launch_bf16(const GemmArgs& big, const GemmArgs& tail, int n_big_pad, hipStream_t s);
void lx_gemm8_mixed_launch_f16(const GemmArgs& big, const GemmArgs& tail, int n_big_pad, hipStream_t s);
// lx_gemm4_kernel: `grid` workgroups, the first sk_full of them whole tiles, the rest halves of split tiles (sk_parts 1: none | 2 | 3: fault injection)
// (np: the workgroups a split tile is shared by, 2 or 3 -- the bf16 and fp16 kernels; the split-bf16 kernel has the two-way form only)
void lx_gemm4_launch_bf16(const GemmArgs& a, unsigned grid, int sk_full, int sk_parts, float* slots, int* flags, int* err, hipStream_t s, int np = 2);
void lx_gemm4_launch_f16(const GemmArgs& a, unsigned grid, int sk_full, int sk_parts, float* slots, int* flags, int* err, hipStream_t s, int np = 2);
void lx_gemm4_launch_split(const GemmArgs& a, unsigned grid, int sk_full, int sk_parts, float* slots, int* flags, int* err, hipStream_t s);
