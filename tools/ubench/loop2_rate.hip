// Micro-benchmark 2: BM=256 GEMM main loop with hand-placed ds_read_b128 (inline asm) and counted lgkmcnt waits.
//   DIST = how many k-steps ahead the fragment reads are issued (1: two register sets, 2: three sets).
//   FLAGS bit1: LDS-DMA staging (2-stage ring as gemm.hip, issue after the K-tile barrier, role split)  bit3: MFMAs
// (timing only: results are not checked; DIST=2 reads the next tile's first fragments before its barrier)
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
constexpr int BM = 256, BN = 256, BK = 64, MI = 4;
constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2, W_BASE = 2 * A_BYTES;

struct Frag { bf16x8 w[2]; bf16x8 x[MI]; };

template <int DIST, int FLAGS>
__global__ __launch_bounds__(512) void kloop2(const __bf16* A, const __bf16* W, float* out, int nkt, int K) {
  constexpr bool DMA = FLAGS & 2, MMA = FLAGS & 8;
  __shared__ __attribute__((aligned(1024))) char smem[2 * A_BYTES + 2 * W_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3, l31 = lane & 31, lhi = lane >> 5;
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
  auto stage = [&](int kt, int slot) {
    if (!DMA) return;
    char* ba = smem + slot * A_BYTES; char* bw = smem + W_BASE + slot * W_BYTES;
#pragma unroll
    for (int j = 0; j < MI; ++j) __builtin_amdgcn_global_load_lds((gptr_t)(asrc[j] + kt * BK), (lptr_t)(ba + (j * 8 + wave) * 1024), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[j] + (size_t)kt * BN * BK), (lptr_t)(bw + (j * 8 + wave) * 1024), 16, 0, 0);
  };
  const int sw = (l31 >> 1) & 7;
  uint32_t aoff[4], woff[4];
  const uint32_t sbase = (uint32_t)(uintptr_t)(lptr_t)smem;
  for (int ks = 0; ks < 4; ++ks) {
    const int so = ((ks * 2 + lhi) ^ sw) * 16;
    aoff[ks] = sbase + (wm * (BM / 2) + l31) * 128 + so;
    woff[ks] = sbase + W_BASE + (wn * 64 + l31) * 128 + so;
  }
  f32x16 acc[2][MI];
  for (int j = 0; j < 2; ++j) for (int i = 0; i < MI; ++i) for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
#define DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
  auto reads = [&](Frag& f, int slot, int ks) {     // 6 reads: W first
    const uint32_t aw = woff[ks] + slot * W_BYTES, aa = aoff[ks] + slot * A_BYTES;
    DSR(f.w[0], aw, 0); DSR(f.w[1], aw, 4096);
    DSR(f.x[0], aa, 0); DSR(f.x[1], aa, 4096); DSR(f.x[2], aa, 8192); DSR(f.x[3], aa, 12288);
  };
  auto mmas = [&](const Frag& f) {
    if (!MMA) return;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < MI; ++i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[j][i]) : "v"(f.w[j]), "v"(f.x[i]));
    __builtin_amdgcn_s_setprio(0);
  };
#define SB __builtin_amdgcn_sched_barrier(0)
#define WAITL(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory"); SB
  Frag f0, f1, f2;
  stage(0, 0); if (nkt > 1) stage(1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (DIST == 1) {
    reads(f0, 0, 0); SB;
    int c = 0;
    for (int kt = 0; kt < nkt; ++kt) {
      const int n = c ^ 1;
      reads(f1, c, 1); SB; WAITL(6); mmas(f0); SB;
      reads(f0, c, 2); SB; WAITL(6); mmas(f1); SB;
      reads(f1, c, 3); SB; WAITL(6); mmas(f0); SB;
      // all my reads of tile kt are issued; they complete before the barrier
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier(); SB;
      if (kt + 1 < nkt) reads(f0, n, 0);
      SB;
      if (wm == 0 && kt + 2 < nkt) stage(kt + 2, c);
      SB; mmas(f1); SB;
      if (wm == 1 && kt + 2 < nkt) stage(kt + 2, c);
      SB;
      c = n;
    }
  } else {
    // three sets rotate with period 3 steps; 4 steps per tile -> unroll 3 tiles (12 steps). nkt % 3 == 0 assumed.
    reads(f0, 0, 0); reads(f1, 0, 1); SB;
    int c = 0;
#define STEP(CUR, NXT2, slot2, ks2) reads(NXT2, slot2, ks2); SB; WAITL(12); mmas(CUR); SB;
#define TILE(F0, F1, F2)                                                            \
    {                                                                               \
      const int n = c ^ 1;                                                          \
      STEP(F0, F2, c, 2)                                                            \
      STEP(F1, F0, c, 3)                                                            \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                            \
      __builtin_amdgcn_s_barrier(); SB;                                             \
      STEP(F2, F1, n, 0)                                                            \
      if (wm == 0 && kt + 2 < nkt) stage(kt + 2, c);                                \
      SB;                                                                           \
      STEP(F0, F2, n, 1)                                                            \
      if (wm == 1 && kt + 2 < nkt) stage(kt + 2, c);                                \
      SB;                                                                           \
      c = n; ++kt;                                                                  \
    }
    for (int kt = 0; kt < nkt;) {
      TILE(f0, f1, f2)    // steps 0..3 use f0 f1 f2 f0 ; prefetched into f2 f0 f1 f2
      TILE(f1, f2, f0)
      TILE(f2, f0, f1)
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n s_nop 15\n s_nop 7" ::: "memory");
  float s = 0;
  for (int j = 0; j < 2; ++j) for (int i = 0; i < MI; ++i) for (int r = 0; r < 16; ++r) s += acc[j][i][r];
  if (s == 123.456f) out[tid] = s;
}
#define CASE(D, F) case D * 100 + F: hipLaunchKernelGGL((kloop2<D, F>), dim3(grid), dim3(512), 0, s, (const __bf16*)A, (const __bf16*)W, out, nkt, K); break;
extern "C" int run_loop2(int mode, const void* A, const void* W, float* out, int grid, int nkt, int K, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  switch (mode) {
    CASE(1, 8) CASE(1, 10) CASE(1, 0) CASE(1, 2) CASE(2, 8) CASE(2, 10) CASE(2, 0) CASE(2, 2)
    default: return -1;
  }
  return (int)hipGetLastError();
}
