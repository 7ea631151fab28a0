// Micro-benchmark: the BM=256 GEMM main loop of loongx_amd/csrc/gemm.hip (same LDS layout, swizzle, k-step software
// pipeline, role-split DMA, one barrier per K tile) with each ingredient switchable, to find which interaction costs time.
//   FLAGS bit0: AGPR accumulators   bit1: LDS-DMA staging   bit2: fragment ds_reads   bit3: MFMAs
//         bit4: no setprio          bit5: DMA issued by all waves right after the barrier (no role split)
//         bits 8-10 (NRD+1): only the first NRD of the six fragment reads per k-step are issued (others keep stale regs)
//         bit11: reads go to scratch registers that no MFMA consumes (no dependency, no wait before the MFMAs)
//         bit7: prime the fragment registers ONCE with real (random) tile data, so an MFMA-only loop multiplies random
//               operands instead of zeros (DVFS: operand toggling costs power, power costs clock)
//         bit12: DMA pieces spread one per MFMA group over the whole K tile (A pieces first, then W; vmcnt(4) at the barrier
//                as a 3-deep W ring allows) instead of a burst of 8 per wave behind the barrier
//         bit13: DMA source as wave-uniform base (SGPR pair) + per-lane 32-bit offset (global saddr form)
//         bit14: DMA through buffer addressing (raw_buffer_load_lds: SRSRC + 32-bit voffset + soffset)
//         bit6: fine interleave: one ds_read after each of the first six MFMAs of a k-step instead of a burst of six
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
constexpr int BM = 256, BN = 256, BK = 64, MI = 4;
constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2, W_BASE = 2 * A_BYTES;

#define PRIO1 if (!(FLAGS & 16)) __builtin_amdgcn_s_setprio(1)
#define PRIO0 if (!(FLAGS & 16)) __builtin_amdgcn_s_setprio(0)
template <int FLAGS>
__global__ __launch_bounds__(512) void kloop(const __bf16* A, const __bf16* W, float* out, int nkt, int K) {
  constexpr bool AGPR = FLAGS & 1, DMA = FLAGS & 2, DSR = FLAGS & 4, MMA = FLAGS & 8, NOPRIO = FLAGS & 16, ALLDMA = FLAGS & 32, FINE = FLAGS & 64, INDEP = FLAGS & 2048;
  constexpr int NRD = ((FLAGS >> 8) & 7) ? ((FLAGS >> 8) & 7) - 1 : 6;
  __shared__ __attribute__((aligned(1024))) char smem[2 * A_BYTES + 3 * W_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3, l31 = lane & 31, lhi = lane >> 5;
  // XCD-aware map of gemm.hip (GROUP_M=4) for an 8 x 32 tile grid: each XCD owns a 4 x 8 patch of tiles
  int tm, tn;
  {
    const int pid = blockIdx.x, lid = (pid & 7) * 32 + (pid >> 3);
    const int gi = lid / 128, in_g = lid % 128;
    tm = gi * 4 + in_g % 4; tn = in_g / 4;
  }
  const __bf16* asrc[MI]; const __bf16* wsrc[4];
  {
    const int rsub = lane >> 3, pslot = lane & 7;
    for (int j = 0; j < MI; ++j) {
      const int row = (j * 8 + wave) * 8 + rsub;
      asrc[j] = A + (size_t)(tm * BM + row) * K + (pslot ^ ((row >> 1) & 7)) * 8;
    }
    for (int j = 0; j < 4; ++j) wsrc[j] = W + ((size_t)tn * (K / BK)) * (BN * BK) + ((j * 8 + wave) * 512 + lane * 8);
  }
  // alternative addressing forms: 32-bit per-lane byte offsets relative to the (wave-uniform) operand base
  uint32_t aoff32[MI], woff32[4];
  for (int j = 0; j < MI; ++j) aoff32[j] = (uint32_t)((const char*)asrc[j] - (const char*)A);
  for (int j = 0; j < 4; ++j) woff32[j] = (uint32_t)((const char*)wsrc[j] - (const char*)W);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 0x7fffffff, 0x00020000);
  bool force_stage = (FLAGS & 128) != 0;
  auto stage_a = [&](int kt, int slot) {
    if (!DMA && !force_stage) return;
    char* base = smem + slot * A_BYTES;
#pragma unroll
    for (int j = 0; j < MI; ++j) {
      if (FLAGS & 16384) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(base + (j * 8 + wave) * 1024), 16, aoff32[j], kt * BK * 2, 0, 0);
      else if (FLAGS & 8192) __builtin_amdgcn_global_load_lds((gptr_t)((const char*)A + kt * BK * 2 + aoff32[j]), (lptr_t)(base + (j * 8 + wave) * 1024), 16, 0, 0);
      else __builtin_amdgcn_global_load_lds((gptr_t)(asrc[j] + kt * BK), (lptr_t)(base + (j * 8 + wave) * 1024), 16, 0, 0);
    }
  };
  auto stage_w = [&](int kt, int slot) {
    if (!DMA && !force_stage) return;
    char* base = smem + W_BASE + slot * W_BYTES;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (FLAGS & 16384) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lptr_t)(base + (j * 8 + wave) * 1024), 16, woff32[j], kt * BN * BK * 2, 0, 0);
      else if (FLAGS & 8192) __builtin_amdgcn_global_load_lds((gptr_t)((const char*)W + (size_t)kt * BN * BK * 2 + woff32[j]), (lptr_t)(base + (j * 8 + wave) * 1024), 16, 0, 0);
      else __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[j] + (size_t)kt * BN * BK), (lptr_t)(base + (j * 8 + wave) * 1024), 16, 0, 0);
    }
  };
  auto piece = [&](int p, int kt, int slot) {   // p 0..3: A pieces, 4..7: W pieces of K tile kt
    if (!DMA || kt >= nkt) return;
    if (FLAGS & 16384) {
      if (p < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(smem + (slot & 1) * A_BYTES + (p * 8 + wave) * 1024), 16, aoff32[p], kt * BK * 2, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lptr_t)(smem + W_BASE + slot * W_BYTES + ((p - 4) * 8 + wave) * 1024), 16, woff32[p - 4], kt * BN * BK * 2, 0, 0);
      return;
    }
    if (p < 4) __builtin_amdgcn_global_load_lds((gptr_t)(asrc[p] + kt * BK), (lptr_t)(smem + (slot & 1) * A_BYTES + (p * 8 + wave) * 1024), 16, 0, 0);
    else __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[p - 4] + (size_t)kt * BN * BK), (lptr_t)(smem + W_BASE + slot * W_BYTES + ((p - 4) * 8 + wave) * 1024), 16, 0, 0);
  };
  const int sw = (l31 >> 1) & 7;
  int slot_off[4];
  for (int ks = 0; ks < 4; ++ks) slot_off[ks] = ((ks * 2 + lhi) ^ sw) * 16;
  const int a_row_off = (wm * (BM / 2) + l31) * 128, w_row_off = (wn * 64 + l31) * 128;
  f32x16 acc[2][MI];
  for (int j = 0; j < 2; ++j) for (int i = 0; i < MI; ++i) for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
  bf16x8 sink[6] = {};
  auto load_frags = [&](int sa, int sw_, int ks, bf16x8 (&wf)[2], bf16x8 (&xf)[MI]) {
    if (!DSR) {
#pragma unroll
      for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(wf[j]));
#pragma unroll
      for (int i = 0; i < MI; ++i) asm volatile("" : "+v"(xf[i]));
      return;
    }
    const char* pa = smem + sa * A_BYTES + a_row_off + slot_off[ks];
    const char* pw = smem + W_BASE + sw_ * W_BYTES + w_row_off + slot_off[ks];
    if (INDEP) {
#pragma unroll
      for (int j = 0; j < 2; ++j) if (j < NRD) sink[j] = *(const bf16x8*)(pw + j * 32 * 128);
#pragma unroll
      for (int i = 0; i < MI; ++i) if (2 + i < NRD) sink[2 + i] = *(const bf16x8*)(pa + i * 32 * 128);
      return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) if (j < NRD) wf[j] = *(const bf16x8*)(pw + j * 32 * 128);
#pragma unroll
    for (int i = 0; i < MI; ++i) if (2 + i < NRD) xf[i] = *(const bf16x8*)(pa + i * 32 * 128);
  };
  auto mma_j = [&](int j, const bf16x8 (&wf)[2], const bf16x8 (&xf)[MI]) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      if (!MMA) { asm volatile("" :: "v"(wf[j]), "v"(xf[i])); continue; }
      if (AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[j][i]) : "v"(wf[j]), "v"(xf[i]));
      else acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], xf[i], acc[j][i], 0, 0, 0);
    }
  };
  auto mma1 = [&](int j, int i, const bf16x8 (&wf)[2], const bf16x8 (&xf)[MI]) {
    if (!MMA) { asm volatile("" :: "v"(wf[j]), "v"(xf[i])); return; }
    if (AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[j][i]) : "v"(wf[j]), "v"(xf[i]));
    else acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], xf[i], acc[j][i], 0, 0, 0);
  };
  auto load1 = [&](int sa, int sw_, int ks, int idx, bf16x8 (&wf)[2], bf16x8 (&xf)[MI]) {   // idx 0,1: W frags; 2..5: A frags
    if (!DSR) { if (idx < 2) asm volatile("" : "+v"(wf[idx])); else asm volatile("" : "+v"(xf[idx - 2])); return; }
    if (idx < 2) wf[idx] = *(const bf16x8*)(smem + W_BASE + sw_ * W_BYTES + w_row_off + slot_off[ks] + idx * 32 * 128);
    else xf[idx - 2] = *(const bf16x8*)(smem + sa * A_BYTES + a_row_off + slot_off[ks] + (idx - 2) * 32 * 128);
  };
  // fine k-step: m(0,0) r0 m(0,1) r1 m(0,2) r2 m(0,3) r3 m(1,0) r4 m(1,1) r5 m(1,2) m(1,3)
  auto fine_step = [&](const bf16x8 (&cw)[2], const bf16x8 (&cx)[MI], bool do_load, int sa, int sw_, int ks, bf16x8 (&nw)[2], bf16x8 (&nx)[MI]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      PRIO1; mma1(q >> 2, q & 3, cw, cx); PRIO0;
      __builtin_amdgcn_sched_barrier(0);
      if (q < 6 && do_load) load1(sa, sw_, ks, q, nw, nx);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  bf16x8 wfA[2] = {}, xfA[MI] = {}, wfB[2] = {}, xfB[MI] = {};
  const long long t_start = __builtin_readcyclecounter();
  stage_a(0, 0); stage_w(0, 0);
  if (nkt > 1) { stage_a(1, 1); stage_w(1, 1); }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  force_stage = false;
  if (FLAGS & 128) {   // prime both register sets from the staged tile
    const char* pa = smem + a_row_off + slot_off[0];
    const char* pw = smem + W_BASE + w_row_off + slot_off[0];
    for (int j = 0; j < 2; ++j) { wfA[j] = *(const bf16x8*)(pw + j * 32 * 128); wfB[j] = *(const bf16x8*)(pw + j * 32 * 128 + 64); }
    for (int i = 0; i < MI; ++i) { xfA[i] = *(const bf16x8*)(pa + i * 32 * 128); xfB[i] = *(const bf16x8*)(pa + i * 32 * 128 + 64); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  load_frags(0, 0, 0, wfA, xfA);
#define PRIO(x) if (!NOPRIO) __builtin_amdgcn_s_setprio(x)
#define LX_STEP(CUR_W, CUR_X, NEXT_STMT) \
  PRIO(1); mma_j(0, CUR_W, CUR_X); PRIO(0); __builtin_amdgcn_sched_barrier(0); NEXT_STMT; __builtin_amdgcn_sched_barrier(0); \
  PRIO(1); mma_j(1, CUR_W, CUR_X); PRIO(0); __builtin_amdgcn_sched_barrier(0);
  int c = 0;
  if (FLAGS & 4096) {
#define LX_STEPP(CUR_W, CUR_X, NEXT_STMT, P0, P1) \
  PRIO(1); mma_j(0, CUR_W, CUR_X); PRIO(0); __builtin_amdgcn_sched_barrier(0); NEXT_STMT; P0; __builtin_amdgcn_sched_barrier(0); \
  PRIO(1); mma_j(1, CUR_W, CUR_X); PRIO(0); __builtin_amdgcn_sched_barrier(0); P1; __builtin_amdgcn_sched_barrier(0);
    int w3 = 0;   // W ring slot (3 deep) being refilled
    for (int kt = 0; kt < nkt; ++kt) {
      const int n = c ^ 1;
      // pieces issued in this K tile period belong to: A of tile kt+1 (slot n... freed at the previous barrier), W of tile kt+2
      LX_STEPP(wfA, xfA, load_frags(c, c, 1, wfB, xfB), piece(2, kt + 1, n), piece(3, kt + 1, n))
      LX_STEPP(wfB, xfB, load_frags(c, c, 2, wfA, xfA), piece(4, kt + 2, w3), piece(5, kt + 2, w3))
      LX_STEPP(wfA, xfA, load_frags(c, c, 3, wfB, xfB), piece(6, kt + 2, w3), piece(7, kt + 2, w3))
      asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      LX_STEPP(wfB, xfB, if (kt + 1 < nkt) load_frags(n, n, 0, wfA, xfA), piece(0, kt + 2, c), piece(1, kt + 2, c))
      c = n; w3 = w3 == 2 ? 0 : w3 + 1;
    }
  } else if (FINE) {
    for (int kt = 0; kt < nkt; ++kt) {
      const int n = c ^ 1;
      fine_step(wfA, xfA, true, c, c, 1, wfB, xfB);
      fine_step(wfB, xfB, true, c, c, 2, wfA, xfA);
      fine_step(wfA, xfA, true, c, c, 3, wfB, xfB);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (ALLDMA || wm == 0) { if (kt + 2 < nkt) { stage_a(kt + 2, c); stage_w(kt + 2, c); } }
      __builtin_amdgcn_sched_barrier(0);
      fine_step(wfB, xfB, kt + 1 < nkt, n, n, 0, wfA, xfA);
      if (!ALLDMA && wm == 1) { if (kt + 2 < nkt) { stage_a(kt + 2, c); stage_w(kt + 2, c); } }
      __builtin_amdgcn_sched_barrier(0);
      c = n;
    }
  } else {
  for (int kt = 0; kt < nkt; ++kt) {
    const int n = c ^ 1;
    LX_STEP(wfA, xfA, load_frags(c, c, 1, wfB, xfB))
    LX_STEP(wfB, xfB, load_frags(c, c, 2, wfA, xfA))
    LX_STEP(wfA, xfA, load_frags(c, c, 3, wfB, xfB))
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (ALLDMA || wm == 0) { if (kt + 2 < nkt) { stage_a(kt + 2, c); stage_w(kt + 2, c); } }
    __builtin_amdgcn_sched_barrier(0);
    LX_STEP(wfB, xfB, if (kt + 1 < nkt) load_frags(n, n, 0, wfA, xfA))
    if (!ALLDMA && wm == 1) { if (kt + 2 < nkt) { stage_a(kt + 2, c); stage_w(kt + 2, c); } }
    __builtin_amdgcn_sched_barrier(0);
    c = n;
  }
  }
  const long long t_end = __builtin_readcyclecounter();
  if (tid == 0) out[1024 + blockIdx.x] = (float)(t_end - t_start);
  if (INDEP) for (int i = 0; i < 6; ++i) asm volatile("" :: "v"(sink[i]));
  if (AGPR) asm volatile("s_nop 15\n s_nop 7" ::: "memory");
  float s = 0;
  for (int j = 0; j < 2; ++j) for (int i = 0; i < MI; ++i) for (int r = 0; r < 16; ++r) s += acc[j][i][r];
  if (s == 123.456f) out[tid] = s;
}
#define CASE(F) case F: hipLaunchKernelGGL((kloop<F>), dim3(grid), dim3(512), 0, s, (const __bf16*)A, (const __bf16*)W, out, nkt, K); break;
extern "C" int run_loop(int flags, const void* A, const void* W, float* out, int grid, int nkt, int K, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  switch (flags) {
    CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15) CASE(6) CASE(2) CASE(4)
    CASE(31) CASE(47) CASE(46) CASE(30) CASE(76) CASE(77) CASE(78) CASE(79) CASE(110) CASE(111) CASE(74) CASE(75) CASE(14+16384+4096) CASE(10+128+8192) CASE(10+128+16384) CASE(14+8192) CASE(14+16384) CASE(14+4096) CASE(10+4096) CASE(10+128+4096) CASE(8+128) CASE(9+128) CASE(10+128) CASE(11+128) CASE(12+0x100) CASE(12+0x300) CASE(12+0x500) CASE(12+0x800) CASE(12+0x800+0x300) CASE(12+0x800+0x500)
    default: return -1;
  }
  return (int)hipGetLastError();
}
