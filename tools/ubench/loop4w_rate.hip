// Micro-benchmark: a GEMM main loop with ONE wave per SIMD -- 256 threads, 2 x 2 waves, 128 x 128 per wave, 256 accumulator registers
// in AGPRs, v_mfma_f32_16x16x32_bf16, every fragment of a 64-deep K tile held in registers (2 k-steps x (8 + 8) fragments = 128 VGPRs),
// the same LDS image / swizzle / LDS-DMA staging as loongx_amd/csrc/gemm.hip -- against tools/ubench/loop_rate (the shipped 8-wave
// loop: two waves per SIMD, 128 x 64 per wave, 32x32x16 MFMAs). The vendor library's fastest kernel on this part has this shape
// (DESIGN 5b item 3a); this file asks whether hipcc can be made to emit such a stream, and what it is worth in cycles and in wall time.
//   FLAGS bit0: LDS-DMA staging   bit1: fragment ds_reads   bit2: MFMAs    bit3: prime the fragment registers once with real data
//         bit4: all 16 DMA pieces of a K tile in one burst behind the first barrier instead of one per five MFMAs
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void* lptr_t;
constexpr int BM = 256, BN = 256, BK = 64;
constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2, STAGE = A_BYTES + W_BYTES;

#define SB() __builtin_amdgcn_sched_barrier(0)
template <int FLAGS>
__global__ __launch_bounds__(256) void kloop4(const __bf16* A, const __bf16* W, float* out, int nkt, int K) {
  constexpr bool DMA = FLAGS & 1, DSR = FLAGS & 2, MMA = FLAGS & 4, PRIME = FLAGS & 8, BURST = FLAGS & 16;
  __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, l15 = lane & 15, lq = lane >> 4;
  int tm, tn;
  {
    const int pid = blockIdx.x, lid = (pid & 7) * 32 + (pid >> 3);
    const int gi = lid / 128, in_g = lid % 128;
    tm = gi * 4 + in_g % 4; tn = in_g / 4;
  }
  // staging: piece p (1 KiB = 8 rows of 128 B) of the A / W tile; this wave moves pieces j * 4 + wave, j = 0..7, of each operand
  uint32_t aoff[8], woff[8];
  {
    const int rsub = lane >> 3, pslot = lane & 7;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = (j * 4 + wave) * 8 + rsub;
      aoff[j] = (uint32_t)(((size_t)(tm * BM + row) * K + (pslot ^ ((row >> 1) & 7)) * 8) * 2);
      woff[j] = (uint32_t)((((size_t)tn * (K / BK)) * (BN * BK) + ((j * 4 + wave) * 512 + lane * 8)) * 2);
    }
  }
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 0x7fffffff, 0x00020000);
  auto piece = [&](int p, int kt, int slot) {        // p 0..7: A pieces, 8..15: W pieces of K tile kt into stage `slot`
    if (!DMA) return;
    if (p < 8) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(smem + slot * STAGE + (p * 4 + wave) * 1024), 16, aoff[p], kt * BK * 2, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lptr_t)(smem + slot * STAGE + A_BYTES + ((p - 8) * 4 + wave) * 1024), 16, woff[p - 8], kt * BN * BK * 2, 0, 0);
  };
  // fragments: 16 rows x 32 k (16 B per lane: row l15, k chunk lq) at k-step ks of rows r0..r0+15 of an operand tile
  const int a_row0 = wm * 128 + l15, w_row0 = wn * 128 + l15;
  auto frag_addr = [&](int row, int ks) { return row * 128 + (((ks * 4 + lq) ^ ((row >> 1) & 7)) * 16); };
  int a_ad[2], w_ad[2];            // byte offsets of block 0 at k-step 0 / 1 (blocks are 16 rows = 2048 B apart: the swizzle term repeats)
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) { a_ad[ks] = frag_addr(a_row0, ks); w_ad[ks] = A_BYTES + frag_addr(w_row0, ks); }
  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 af[2][8], wf[2][8];       // [k-step][block]
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int i = 0; i < 8; ++i) { af[ks][i] = bf16x8{}; wf[ks][i] = bf16x8{}; }
  auto rd = [&](int stage, int ks, int idx) {        // idx 0..7: W block idx, 8..15: A block idx - 8 (16 rows x 128 B = 2048 B apart)
    if (!DSR) { if (idx < 8) asm volatile("" : "+v"(wf[ks][idx])); else asm volatile("" : "+v"(af[ks][idx - 8])); return; }
    const char* base = smem + stage * STAGE;
    if (idx < 8) wf[ks][idx] = *(const bf16x8*)(base + w_ad[ks] + idx * 2048);
    else af[ks][idx - 8] = *(const bf16x8*)(base + a_ad[ks] + (idx - 8) * 2048);
  };
  auto mm = [&](int ks, int n) {                      // MFMA n (0..63) of k-step ks: W block n >> 3 (held for 8 MFMAs) x A block n & 7
    const int j = n >> 3, i = n & 7;
    if (!MMA) { asm volatile("" :: "v"(wf[ks][j]), "v"(af[ks][i])); return; }
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(wf[ks][j]), "v"(af[ks][i]));
  };
  const long long t_start = __builtin_readcyclecounter();
  if (DMA || PRIME) {
    const bool keep = DMA;
    for (int p = 0; p < 16; ++p) {
      if (p < 8) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(smem + (p * 4 + wave) * 1024), 16, aoff[p], 0, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lptr_t)(smem + A_BYTES + ((p - 8) * 4 + wave) * 1024), 16, woff[p - 8], 0, 0, 0);
    }
    if (keep && nkt > 1) for (int p = 0; p < 16; ++p) piece(p, 1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (PRIME || DSR) {
    const char* base = smem;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 8; ++i) { wf[ks][i] = *(const bf16x8*)(base + w_ad[ks] + i * 2048); af[ks][i] = *(const bf16x8*)(base + a_ad[ks] + i * 2048); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  // One K tile (stage c holds it, its k-step-0 fragments are in registers):
  //   k-step 0: 64 MFMAs; the 16 fragment reads of k-step 1 behind the first 16; then (all reads of this stage issued) wait, barrier,
  //             and the DMA of K tile kt + 2 into this stage, A pieces one per five MFMAs
  //   k-step 1: 64 MFMAs; the W pieces one per five MFMAs; wait for the pieces of K tile kt + 1 (issued one iteration ago), barrier,
  //             the 16 fragment reads of k-step 0 of tile kt + 1 behind the last MFMAs
  int c = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    const int n = c ^ 1;
    const int kt2 = min(kt + 2, nkt - 1);      // branch-free tail: the last tiles are staged again (never read)
#pragma unroll
    for (int m = 0; m < 64; ++m) {
      if (m == 16) {
        SB();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        SB();
        if (BURST) for (int p = 0; p < 16; ++p) piece(p, kt2, c);
      }
      mm(0, m); SB();
      if (m < 16) { rd(c, 1, m); SB(); }
      if (!BURST && m >= 16 && (m - 16) % 6 == 0) { piece((m - 16) / 6, kt2, c); SB(); }      // m = 16, 22, ..., 58: pieces 0..7
    }
#pragma unroll
    for (int m = 0; m < 64; ++m) {
      if (m == 43) {
        SB();
        if (DMA) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        SB();
      }
      mm(1, m); SB();
      if (!BURST && m < 40 && m % 5 == 0) { piece(8 + m / 5, kt2, c); SB(); }                 // m = 0, 5, ..., 35: pieces 8..15
      if (m >= 43 && m < 59) { rd(n, 0, m - 43); SB(); }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    c = n;
  }
  const long long t_end = __builtin_readcyclecounter();
  if (tid == 0) out[1024 + blockIdx.x] = (float)(t_end - t_start);
  asm volatile("s_nop 15\n s_nop 7" ::: "memory");
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { f32x4 v; asm volatile("v_accvgpr_read_b32 %0, %4\n v_accvgpr_read_b32 %1, %5\n v_accvgpr_read_b32 %2, %6\n v_accvgpr_read_b32 %3, %7" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]) : "a"(acc[i][j][0]), "a"(acc[i][j][1]), "a"(acc[i][j][2]), "a"(acc[i][j][3])); s += v[0] + v[1] + v[2] + v[3]; }
  if (s == 123.456f) out[tid] = s;
}
#define CASE(F) case F: hipLaunchKernelGGL((kloop4<F>), dim3(grid), dim3(256), 0, s, (const __bf16*)A, (const __bf16*)W, out, nkt, K); break;
extern "C" int run_loop4(int flags, const void* A, const void* W, float* out, int grid, int nkt, int K, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  switch (flags) {
    CASE(4) CASE(12) CASE(14) CASE(13) CASE(15) CASE(7) CASE(31) CASE(6)
    default: return -1;
  }
  return (int)hipGetLastError();
}
