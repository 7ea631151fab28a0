// gemm4.h -- lx_gemm4_kernel: the 256 x 256 x 64 GEMM tile with ONE wave per SIMD (kernel TEMPLATE; gemm4.hip instantiates the bf16
// form, gemm4_modes.hip the split-bf16 and the fp16-operand forms). See gemm8.h for the operand layouts both families share.
#pragma once
#include "gemm_common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// lx_gemm4_kernel: the same 256 x 256 x 64 tile with ONE wave per SIMD -- 256 threads, 2 x 2 waves of 128 x 128, the 256 accumulator
// registers of a wave in AGPRs, v_mfma_f32_16x16x32_bf16 (a weight fragment is held for eight MFMAs), ALL fragments of a K tile in
// registers (2 k-steps x (8 + 8) x 4 VGPRs), one ds_read_b128 or one LDS-DMA piece per MFMA gap, two barriers per K tile: the reads
// of a stage are finished (registers) before its refill is issued, so two stages of 64 KiB suffice. This is the shape of the vendor
// library's fastest kernel on this part; measured against the 8-wave loop above in tools/ubench/loop4w_rate:
// 1.39 vs 1.58 us per K tile (the fragment bytes read from LDS per MFMA cycle are 2/3, the wave count per SIMD half). Same LDS image,
// swizzle, pre-tiled weights and tile map as gemm_tile. Epilogues so far: bias / GELU / bf16 or fp32 store / gated residual (no LoRA,
// no LX_EPI_QKV: those launches stay on the kernels above).
constexpr int G4_STAGE = 256 * BK * 2 + BN * BK * 2;      // A tile + W tile of one K tile: 64 KiB
constexpr int G4_LDS = 2 * G4_STAGE;
constexpr int G4_PLD = 132;                                // fp32 row stride of the epilogue patch (128 + 4 pad)
static_assert(G4_LDS <= 160 * 1024 && 4 * 2 * 16 * G4_PLD * 4 <= G4_LDS, "LDS budget / epilogue patches");

// (A split-tail form -- the tiles of a partial last round cut into K ranges, one workgroup each, meeting through the workspace -- was
// built and measured in round 3: correct, but its park / fetch code made hipcc spill accumulators on EVERY tile's path (132 VGPRs, 52
// scratch operations behind the main loop), and at K = 3072 it lost to the 8-wave kernels' half-height tail anyway; removed again.)
#ifdef LX_G4_PROBE     /* phase stamps (s_memtime, 100 MHz) of lx_gemm4_kernel per workgroup: tools/g4_probe.py */
__device__ unsigned long long lx_g4_probe_buf[8192 * 8];
#define G4_STAMP(k) if (threadIdx.x == 0 && blockIdx.x < 8192) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); lx_g4_probe_buf[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); }
#else
#define G4_STAMP(k)
#endif

// Split form (sk_parts == 2): the tiles at positions >= sk_full of the tile order -- a partial last round, or every tile of a launch
// with at most 128 of them -- are computed by TWO workgroups, half of K each, next to each other in the grid, and FINISHED by both:
// each parks (plain fp32 rows, sc1 stores into its 256 x 256 slot of the caller's workspace) the four 16-row blocks per wave that the
// other one owns, raises a flag, waits for the partner's (bounded: the workspace's error word reports a time-out, as the pair
// kernel's does), and runs the epilogue on its own four blocks, adding the partner's sums row by row where it reads its own from the
// patch -- the accumulators themselves are never touched outside the main loop and the one place per block that writes them to the
// patch (anything else makes hipcc shuffle and spill them). Round 3's form (one half parks all eight blocks, the other finishes all
// eight) left the epilogue -- a chain of blocks, each a memory round trip long -- on half of the workgroups: section 9 of DESIGN.md.

// SPLIT = the split-bf16 ("precise") problems of lx_gemm_split_kernel: k_segs passes over K in one accumulation (A_hi W_hi, A_lo W_hi,
// A_hi W_lo: the K-tile index of the loop maps to (segment, tile) -> source offsets, once per K tile on the scalar unit) and the
// hi / lo output pair of LX_EPI_SPLIT_BF16. A separate instantiation, so that the bf16 path keeps its exact instruction stream.
// F16 = fp16 operands (LX_OPERANDS_F16): v_mfma_f32_16x16x32_f16 on the same fragments, and the 16-bit store of the bf16-store epilogue
// writes fp16 (saturated, reported through P.f16_ovf); LX_EPI_QKV outputs stay bf16.
// NP = the number of workgroups a split tile is shared by: 2 (above), or 3 (round 5: thirds of K; part 0 owns the 16-row blocks 0-2 of
// every wave, part 1 blocks 3-5, part 2 blocks 6-7; each parks what the other two own, raises its flag as a count of two readers, waits
// for both partners and adds both partners' sums) -- for tails of up to a third of a round, where any two-way tail costs at least half
// a K loop + the exchange (tools/gemm_tail_cost.py: 37 us at K = 3072). Its own instantiation: the two-way kernel keeps its code.
template <bool SPLIT, bool F16 = false, int NP = 2>
__global__ __launch_bounds__(G4_THREADS) void lx_gemm4_kernel(const GemmArgs args, const int sk_full, const int sk_parts, float* __restrict__ sk_slots,
                                                              int* __restrict__ sk_flags, int* __restrict__ sk_err) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(1024))) char smem[G4_LDS];
  constexpr int BM = 256, A_BYTES = BM * BK * 2;
  const int pid = blockIdx.x;
  G4_STAMP(0)
  const int total = args.tile_start[MAX_SUB];
  int lid, part = 0;
  if (pid < sk_full) {                                 // a whole tile: the XCD-aware map over the whole-tile part of the order
    const int q = sk_full >> 3, r = sk_full & 7;
    const int xcd = pid & 7, inx = pid >> 3;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + inx;
  } else {                                             // one half of K of a split tile
    const int r = pid - sk_full;
    lid = sk_full + r / NP;
    part = r % NP;
  }
  (void)total;
  const bool split_tile = pid >= sk_full && sk_parts >= 2;          // (3: fault injection, part 1 never raises its flag -- tools/race_screen_g4.py)
  const int g = tile_group(args, lid);
  lx_gemm_desc P = args.p[g];
  int tm, tn;
  tile_coords<BM>(P, lid - args.tile_start[g], tm, tn);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, l15 = lane & 15, lq = lane >> 4;
  int m0 = tm * BM, n0 = tn * BN;
  int M = P.M, N = P.N;
  const int K = P.K;
  const bool w_tiled = (P.epilogue & LX_W_TILED) != 0;
  const int nk1 = K / BK;                              // K tiles of one pass over K
  const int nkt = SPLIT ? nk1 * max(P.k_segs, 1) : nk1; // K tiles of the tile (all segments); this workgroup's share: [kt_begin, kt_end)
  const int kt_begin = !split_tile ? 0 : NP == 2 ? (part ? nkt >> 1 : 0) : (part * nkt) / NP;
  const int kt_end = !split_tile ? nkt : NP == 2 ? (part ? nkt : nkt >> 1) : ((part + 1) * nkt) / NP;

  // ---- staging: this wave moves pieces j * 4 + wave (j = 0..7; 1 KiB = 8 rows of 128 B each) of both operand tiles ----
  uint32_t aoff[8], woff[8];
  {
    const int rsub = lane >> 3, pslot = lane & 7;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = (j * 4 + wave) * 8 + rsub;
      const int lslot = pslot ^ ((row >> 1) & 7);
      aoff[j] = (uint32_t)((min(m0 + row, M - 1) - m0) * P.lda + lslot * 8) * 2u;
      woff[j] = w_tiled ? (uint32_t)((j * 4 + wave) * 512 + lane * 8) * 2u : (uint32_t)((min(n0 + row, N - 1) - n0) * P.ldw + lslot * 8) * 2u;
    }
  }
  const __bf16* a_org = (const __bf16*)P.A + (size_t)m0 * P.lda;
  const int kw_tiles = SPLIT && P.k_segs == 3 ? 2 * nk1 : nk1;       // K tiles per weight row block ([W_hi | W_lo] with three segments)
  const __bf16* w_org = w_tiled ? (const __bf16*)P.W + ((size_t)tn * kw_tiles) * (BN * BK) : (const __bf16*)P.W + (size_t)n0 * P.ldw;
  const lx_rsrc_t rs_a = lx_make_rsrc(a_org), rs_w = lx_make_rsrc(w_org);
  const int w_kstride_b = (w_tiled ? BN * BK : BK) * 2;
  // K-tile index of the loop -> byte offsets of its A and W tiles (gemm_mainloop's a_soff / w_soff; the identity map for !SPLIT)
  const int a_lo_b = SPLIT ? P.a_lo_off * 2 : 0;
  auto a_soff = [&](int t) -> int {
    if constexpr (!SPLIT) return t * (BK * 2);
    else {
      const int seg = t >= 2 * nk1 ? 2 : (t >= nk1 ? 1 : 0);
      return (t - seg * nk1) * (BK * 2) + (seg == 1 ? a_lo_b : 0);
    }
  };
  auto w_soff = [&](int t) -> int {
    if constexpr (!SPLIT) return t * w_kstride_b;
    else {
      const int seg = t >= 2 * nk1 ? 2 : (t >= nk1 ? 1 : 0);
      return (t - seg * nk1 + (seg == 2 ? nk1 : 0)) * w_kstride_b;
    }
  };
  auto piece = [&](int p_, int a_so, int w_so, int stage) {       // p_ 0..7: A pieces, 8..15: W pieces of the K tile at (a_so, w_so)
    if (p_ < 8) lx_buf_to_lds(rs_a, (lptr_t)(smem + stage * G4_STAGE + (p_ * 4 + wave) * 1024), aoff[p_], a_so);
    else lx_buf_to_lds(rs_w, (lptr_t)(smem + stage * G4_STAGE + A_BYTES + ((p_ - 8) * 4 + wave) * 1024), woff[p_ - 8], w_so);
  };
  // ---- fragments: 16 rows x 32 k = 16 B per lane (row l15, k chunk lq); row blocks are 2 KiB apart and share the swizzle term ----
  int a_ad[2], w_ad[2];
  {
    const int ar = wm * 128 + l15, wr = wn * 128 + l15;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      a_ad[ks] = ar * 128 + (((ks * 4 + lq) ^ ((ar >> 1) & 7)) * 16);
      w_ad[ks] = A_BYTES + wr * 128 + (((ks * 4 + lq) ^ ((wr >> 1) & 7)) * 16);
    }
  }
  f32x4 acc[8][8];                                    // [m block i][n block j]: m = i*16 + l15, n = j*16 + 4*lq + r
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 af[2][8], wf[2][8];                          // [k-step][block]
  auto rd = [&](int stage, int ks, int idx) {         // idx 0..7: W block, 8..15: A block idx - 8
    const char* base = smem + stage * G4_STAGE;
    if (idx < 8) wf[ks][idx] = *(const bf16x8*)(base + w_ad[ks] + idx * 2048);
    else af[ks][idx - 8] = *(const bf16x8*)(base + a_ad[ks] + (idx - 8) * 2048);
  };
  auto mm = [&](int ks, int n_) {                      // MFMA n_ (0..63) of a k-step: W block n_ >> 3 (held for 8 MFMAs) x A block n_ & 7
    const int j = n_ >> 3, i = n_ & 7;
    if constexpr (F16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(wf[ks][j]), "v"(af[ks][i]));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(wf[ks][j]), "v"(af[ks][i]));
  };
#define G4_SB() __builtin_amdgcn_sched_barrier(0)
  // ---- LoRA up-projection (block.py via peft: y += (x A^T) B^T on the adapter's rows): acc starts from t . up^T, as one extra 32-deep
  // MFMA k-step. Lane (row l15, chunk lq) carries ranks 2 lq and 2 lq + 1; each rank fills four k-slots with the bf16 hi / lo cross
  // terms [u_hi, u_hi, u_lo, u_lo] x [t_hi, t_lo, t_hi, t_lo]: fp32-class (2^-16), as lora_apply above does for the 8-wave kernels.
  // The loads go out BEFORE the operand DMA (vmcnt is one in-order queue: the wait that covers K tile 0 then covers them too).
  const bool has_lora = P.lora_t != nullptr && kt_begin == 0;     // (the planner admits rank <= 8, even, 8-byte aligned rows; once per tile: with its first K tiles)
  u32x2 lu[8], lt[8][4];
  if (has_lora) {
    const int R = P.lora_r, nsplit = P.lora_nsplit;
    const int toff = R * min(n0 / max(P.lora_mod_cols, 1), P.lora_toff_max);
    const int rk = min(2 * lq, R - 2);                 // (ranks past R: loaded from a valid address, zeroed below)
#pragma unroll
    for (int j = 0; j < 8; ++j) lu[j] = *(const u32x2*)(P.lora_up + (size_t)min(n0 + wn * 128 + j * 16 + l15, N - 1) * R + rk);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float* tp = P.lora_t + (size_t)min(m0 + wm * 128 + i * 16 + l15, M - 1) * P.lora_ldt + toff + rk;
#pragma unroll
      for (int q = 0; q < 4; ++q) lt[i][q] = *(const u32x2*)(tp + (size_t)min(q, nsplit - 1) * P.lora_split_stride);
    }
  }
  // prologue: K tiles 0 and 1 staged; tile 0 landed; its k-step-0 fragments read
  {
    const int t1 = min(kt_begin + 1, kt_end - 1);
    const int a0 = a_soff(kt_begin), w0 = w_soff(kt_begin), a1 = a_soff(t1), w1 = w_soff(t1);
#pragma unroll
    for (int p_ = 0; p_ < 16; ++p_) piece(p_, a0, w0, 0);
#pragma unroll
    for (int p_ = 0; p_ < 16; ++p_) piece(p_, a1, w1, 1);
  }
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  G4_SB();
  G4_STAMP(1)
#pragma unroll
  for (int x = 0; x < 16; ++x) rd(0, 0, x);
  if (has_lora) {
    const int R = P.lora_r, nsplit = P.lora_nsplit;
    const bool live = 2 * lq < R;                      // this lane's two ranks exist
    auto frag_u = [&](u32x2 v) {
      const float a = live ? __uint_as_float(v[0]) : 0.f, b = live ? __uint_as_float(v[1]) : 0.f;
      const uint16_t ah = f32_to_bf16(a), bh = f32_to_bf16(b);
      const uint16_t al = f32_to_bf16(a - bf16_to_f32(ah)), bl = f32_to_bf16(b - bf16_to_f32(bh));
      const u32x4 w = {(uint32_t)ah * 0x10001u, (uint32_t)al * 0x10001u, (uint32_t)bh * 0x10001u, (uint32_t)bl * 0x10001u};
      return __builtin_bit_cast(bf16x8, w);
    };
    auto frag_t = [&](float a, float b) {
      a = live ? a : 0.f; b = live ? b : 0.f;
      const uint16_t ah = f32_to_bf16(a), bh = f32_to_bf16(b);
      const uint32_t pa = (uint32_t)ah | ((uint32_t)f32_to_bf16(a - bf16_to_f32(ah)) << 16);
      const uint32_t pb = (uint32_t)bh | ((uint32_t)f32_to_bf16(b - bf16_to_f32(bh)) << 16);
      const u32x4 w = {pa, pa, pb, pb};
      return __builtin_bit_cast(bf16x8, w);
    };
    bf16x8 uf[8], tf[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) uf[j] = frag_u(lu[j]);
#pragma unroll
    for (int i = 0; i < 8; ++i) {                      // slabs added in slab order, as lora_sum does
      float a = __uint_as_float(lt[i][0][0]), b = __uint_as_float(lt[i][0][1]);
#pragma unroll
      for (int q = 1; q < 4; ++q)
        if (q < nsplit) { a += __uint_as_float(lt[i][q][0]); b += __uint_as_float(lt[i][q][1]); }
      tf[i] = frag_t(a, b);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(uf[j], tf[i], acc[i][j], 0, 0, 0);
    // (the builtin, not the inline-asm form of the main loop: hipcc moves accumulators between AGPRs around this branch, and only
    //  knows the MFMA -> accvgpr-read wait states of instructions it can see -- with asm statements here the last blocks lost their term)
    asm volatile("s_nop 15\n s_nop 7" ::: "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // One K tile (stage c; its k-step-0 fragments are in registers):
  //   k-step 0: 64 MFMAs; the 16 reads of k-step 1 behind the first 16; then every read of this stage is issued: wait, barrier, and the
  //             refill of this stage with K tile kt + 2 starts -- A pieces one per six MFMAs
  //   k-step 1: 64 MFMAs; W pieces one per five; vmcnt(16) = K tile kt + 1 (issued an iteration ago) has landed, barrier, and its
  //             k-step-0 reads behind the last MFMAs.
  // Branch-free tail: past the last K tile the final tile is staged again (identical bytes over a stage nobody reads any more).
  G4_STAMP(2)
  int c = 0;
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int n = c ^ 1;
    const int kt2 = min(kt + 2, kt_end - 1);
    const int a_so2 = a_soff(kt2), w_so2 = w_soff(kt2);
#pragma unroll
    for (int m = 0; m < 64; ++m) {
      if (m == 16) {
        G4_SB();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        G4_SB();
      }
      mm(0, m); G4_SB();
      if (m < 16) { rd(c, 1, m); G4_SB(); }
      if (m >= 16 && (m - 16) % 6 == 0) { piece((m - 16) / 6, a_so2, w_so2, c); G4_SB(); }
    }
#pragma unroll
    for (int m = 0; m < 64; ++m) {
      if (m == 43) {
        G4_SB();
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        G4_SB();
      }
      mm(1, m); G4_SB();
      if (m < 40 && m % 5 == 0) { piece(8 + m / 5, a_so2, w_so2, c); G4_SB(); }
      if (m >= 43 && m < 59) { rd(n, 0, m - 43); G4_SB(); }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    c = n;
  }
#undef G4_SB
  // the inline-asm MFMAs are opaque to the hazard recogniser (MFMA write -> v_accvgpr_read: 18 wait states); every LDS-DMA piece
  // has landed and every wave is done with the operand stages before the patch below reuses them
  G4_STAMP(3)
  asm volatile("s_nop 15\n s_nop 7\n s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  G4_STAMP(4)

  // ---- split tiles: BOTH halves finish half of the tile. Part p parks (fp32, sc1 stores into its own slot) the four 16-row blocks of
  // every wave that the other part owns -- part 0 owns blocks 0-3, part 1 blocks 4-7 -- raises its flag, waits for the partner's
  // (bounded), and runs the real epilogue on its own four blocks with the partner's sums added. A wave's epilogue is a chain of
  // blocks, each a memory round trip long (tools/g4_probe_split.py: 8 blocks = 36-38 us for the gated residual, on whole tiles too;
  // half of the blocks = half of the time): with one half parking all eight blocks and the other finishing all eight, the launch paid
  // 12.6 us of parking + the wait + a full 38-us epilogue on 120 of the 256 CUs.
  const float* partner = nullptr;                      // the other half's sums (tile-local [256][256] fp32), added in the epilogue
  const float* partner2 = nullptr;                     // NP = 3: the third workgroup's
  float* my_slot = nullptr;
  if (split_tile) {
    my_slot = sk_slots + (size_t)(pid - sk_full) * SK_SLOT_FLOATS;
    if constexpr (NP == 2) {
      partner = sk_slots + (size_t)((pid - sk_full) ^ 1) * SK_SLOT_FLOATS;    // read with sc1 loads (written with sc1 stores): no cache maintenance
    } else {
      const int gb = (pid - sk_full) - part;
      partner = sk_slots + (size_t)(gb + (part + 1) % NP) * SK_SLOT_FLOATS;
      partner2 = sk_slots + (size_t)(gb + (part + 2) % NP) * SK_SLOT_FLOATS;
    }
  }
  auto pld4 = [&](size_t off_floats, const float* from) {    // 16 B of a partner's slot, agent scope (sc1: not from this CU's L1 / a stale L2 line)
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(lx_make_rsrc(from), (int)(off_floats * 4), 0, PAIR_AUX_SC1);
    return f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
  };
  const int lm0 = wm * 128, ln0 = wn * 128;            // this wave's tile-local origin
  // publish my parked blocks, then wait for the partner's: every wave's sc1 stores acknowledged (vmcnt(0)) before the flag; the
  // workgroup-scope fences are for the COMPILER (nothing of the parked sums may sink below the flag, no load of the partner's may rise
  // above it). Bounded: a partner that never shows up sets the workspace's error word (lx_gemm_workspace_status), nothing hangs.
  auto exchange = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (tid == 0) {
      const int me = pid - sk_full, other = NP == 2 ? (me ^ 1) : (me - part + (part + 1) % NP), other2 = me - part + (part + 2) % NP;
      // The error word is STICKY until the host resets the workspace (lx_gemm_workspace_status): after a time-out a partner's flag may be
      // raised late and stay up, and a later launch on the same slot (the engine polls the word asynchronously, graph replays keep coming)
      // would pass its wait on that stale flag and add sums the partner has not written. So a workspace with the word up is refused:
      // no flag is trusted, the tile finishes invalid like the launch that timed out, and the word stays up for the host.
      bool ok = __hip_atomic_load(sk_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
      if (ok) {
        if (!(sk_parts == 3 && part == 1))             // (3: fault injection, part 1 never raises its flag -- tools/race_screen_g4.py)
          __hip_atomic_store(sk_flags + me, NP - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (a count of readers)
        int spins = 0;
        while (__hip_atomic_load(sk_flags + other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1 << 20)) { ok = false; break; }
        }
        if constexpr (NP == 3) {
          while (ok && __hip_atomic_load(sk_flags + other2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 20)) { ok = false; break; }
          }
        }
        if (ok) {                                      // (each flag: raised by its owner, taken down by its readers)
          if constexpr (NP == 2) __hip_atomic_store(sk_flags + other, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else {
            __hip_atomic_fetch_add(sk_flags + other, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(sk_flags + other2, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        } else __hip_atomic_store(sk_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // lx_gemm_workspace_status reports it
      }
    }
    __syncthreads();
    G4_STAMP(6)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  };

  // ---- epilogue: 16-row blocks through a wave-private fp32 patch, so that every global access is a 16-byte row access ----
  const int epi = P.epilogue & 0xff;
  const bool do_gelu = (P.epilogue & LX_EPI_GELU) != 0;
  const int mw0 = m0 + wm * 128, nw0 = n0 + wn * 128;
  float* patch = (float*)smem + wave * (2 * 16 * G4_PLD);          // two 16-row patches per wave
  const bool bf16_out = epi == LX_EPI_STORE_BF16;
  const int c8 = (lane & 15) * 8, c4 = (lane & 31) * 4;
  const int ncol = nw0 + (bf16_out ? c8 : c4);
  const bool col_ok = ncol < N;
  f32x4 bias0 = {0.f, 0.f, 0.f, 0.f}, bias1 = {0.f, 0.f, 0.f, 0.f};
  if (P.bias && col_ok) {
    bias0 = *(const f32x4*)(P.bias + ncol);
    if (bf16_out) bias1 = *(const f32x4*)(P.bias + ncol + 4);
  }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(bias0), "+v"(bias1)::"memory");
  const bool gelu0 = do_gelu && ncol >= P.gelu_col_start;
  const int m_base = args.m_base[g];
  static_assert(!(F16 && SPLIT), "fp16 operands exclude the split-bf16 mode");
  float f16_mx = 0.f;                                  // LX_OPERANDS_F16: max |x| of what this lane rounded to fp16 (pack_f16x2_sat)
  // LX_EPI_QKV tiles (block.py:60-99: attn.norm_q / norm_k + apply_rotary_emb, and the attention kernel's V^T image): a wave's 128
  // columns are exactly one head, so the sum of squares of a row is a 16-lane reduction of the row-access layout (no LDS exchange)
  const bool qkv_tile = (P.epilogue & LX_EPI_QKV) != 0 && n0 < 3 * P.qkv_d;
  const int qkind = qkv_tile ? n0 / P.qkv_d : -1;      // 0: k, 1: v, 2: q (tile-uniform)
  f32x4 nw0v = {1.f, 1.f, 1.f, 1.f}, nw1v = {1.f, 1.f, 1.f, 1.f};
  if (qkv_tile && qkind != 1) {
    const float* nwp = (qkind == 2 ? P.qkv_norm_q : P.qkv_norm_k) + c8;
    nw0v = *(const f32x4*)nwp; nw1v = *(const f32x4*)(nwp + 4);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(nw0v), "+v"(nw1v)::"memory");
  }
  // One 16-row block at a time through TWO patches: with one wave per SIMD nothing else covers the LDS round trips, so block i + 1 is
  // written (accumulators -> patch (i + 1) & 1) right behind the reads of block i, under their latency and block i's arithmetic and
  // stores (a strictly serial write -> wait -> read -> wait -> store chain per block measured 12 us per tile: tools/g4_probe.py).
  // Generic lambdas over an integral constant, not loops: acc[i] must be a compile-time register index (a loop that hipcc declines to
  // unroll sends all 256 accumulators to scratch).
  auto put = [&](auto ic_) {
    constexpr int i = decltype(ic_)::value;
    float* pt = patch + (i & 1) * (16 * G4_PLD);
    // the block's eight accumulators stay in AGPRs up to here (left to itself hipcc moves all 256 to VGPRs at once and spills them)
    asm volatile("" : "+a"(acc[i][0]), "+a"(acc[i][1]), "+a"(acc[i][2]), "+a"(acc[i][3]), "+a"(acc[i][4]), "+a"(acc[i][5]), "+a"(acc[i][6]), "+a"(acc[i][7]));
#pragma unroll
    for (int j = 0; j < 8; ++j) *(f32x4*)(pt + l15 * G4_PLD + j * 16 + 4 * lq) = acc[i][j];
  };
  // a block the OTHER half of a split tile owns: its sums as they are, row layout, into this workgroup's slot -- agent-scope
  // write-through stores (the owner reads them with sc1 loads: no cache maintenance on either side); the next block's accumulators go
  // into the other patch as in block() below
  auto park = [&](auto ic_, auto nx_) {
    constexpr int i = decltype(ic_)::value, nx = decltype(nx_)::value;
    const float* pt = patch + (i & 1) * (16 * G4_PLD);
    __builtin_amdgcn_wave_barrier();
    if constexpr (nx >= 0) { if (mw0 + nx * 16 < M) put(std::integral_constant<int, nx>{}); }
    __builtin_amdgcn_sched_barrier(0);
    if (mw0 + i * 16 >= M) return;
    const lx_rsrc_t rs_slot = lx_make_rsrc(my_slot);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int row = t * 2 + (lane >> 5);
      const f32x4 v = *(const f32x4*)(pt + row * G4_PLD + c4);
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, rs_slot,
                                             (int)(((size_t)(wm * 128 + i * 16 + row) * 256 + wn * 128 + c4) * 4), 0, PAIR_AUX_SC1);
    }
  };
  auto block = [&](auto ic_, auto nx_) {               // block i; nx = the block behind it in this workgroup's order (-1: none)
    constexpr int i = decltype(ic_)::value, nx = decltype(nx_)::value;
    const int mb = mw0 + i * 16;
    if (mb >= M) return;                               // (wave-uniform; the blocks behind it are out of range as well)
    const float* pt = patch + (i & 1) * (16 * G4_PLD);
    f32x4 res[8], gat[8];
    if (epi == LX_EPI_RESID_F32 && !qkv_tile) {        // residual / gate rows of the block, requested first
      const int rpb = P.rows_per_batch;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int m = mb + t * 2 + (lane >> 5);
        if (m < M && col_ok) {
          res[t] = *(const f32x4*)((const float*)P.C + (size_t)m * P.ldc + ncol);
          if (P.gate) gat[t] = *(const f32x4*)(P.gate + (size_t)((m_base + m) / rpb) * P.gate_ld + ncol);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    // the next block goes into the other patch HERE, at one place per block and outside every epilogue branch: with the accumulator reads
    // inside the (tile-uniform) branches hipcc has to reconcile 256 AGPR assignments at every join, through VGPRs and scratch
    if constexpr (nx >= 0) { if (mw0 + nx * 16 < M) put(std::integral_constant<int, nx>{}); }
    __builtin_amdgcn_sched_barrier(0);
    if (partner) {
      // split owner: the other half's sums of this block are added INTO the patch, row layout (two 16-B sc1 loads per lane and four-row
      // pass), before any epilogue path reads it -- one place for all paths (the V^T path reads the patch by columns: per-element
      // partner loads there cost 256 four-byte loads per lane and tile, the q/k/v launch went from 136 to 203 us)
      f32x4 pa[4][2], pb[4][2];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const size_t po = (size_t)(lm0 + i * 16 + t * 4 + (lane >> 4)) * 256 + ln0 + c8;
        pa[t][0] = pld4(po, partner); pa[t][1] = pld4(po + 4, partner);
        if constexpr (NP == 3) { pb[t][0] = pld4(po, partner2); pb[t][1] = pld4(po + 4, partner2); }
      }
      float* pw = patch + (i & 1) * (16 * G4_PLD);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float* q_ = pw + (t * 4 + (lane >> 4)) * G4_PLD + c8;
        f32x4 v0 = *(const f32x4*)q_, v1 = *(const f32x4*)(q_ + 4);
#pragma unroll
        for (int c_ = 0; c_ < 4; ++c_) { v0[c_] += pa[t][0][c_]; v1[c_] += pa[t][1][c_]; }
        if constexpr (NP == 3) {                       // (own + next part's + the part after: a fixed order per owner)
#pragma unroll
          for (int c_ = 0; c_ < 4; ++c_) { v0[c_] += pb[t][0][c_]; v1[c_] += pb[t][1][c_]; }
        }
        *(f32x4*)q_ = v0; *(f32x4*)(q_ + 4) = v1;
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (qkv_tile) {
      const int D = P.qkv_d, L = P.rows_per_batch, H = D >> 7;
      const int gm = m_base + mb, b = gm / L, p0 = gm - b * L;           // (M and L are multiples of 32: a 16-row block is whole, in one batch)
      const int h = (nw0 - qkind * D) >> 7;
      if (qkind == 1) {
        // v: 16 keys x 128 head dims -> V^T rows; lane = (d, group of 8 keys), the 16-key interleave of the attention kernel's image
        const int gk = lane & 1;
        float e[4][8];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int rk = qkv_vt_interleave(gk * 8 + k), dk = t * 32 + (lane >> 1);
            e[t][k] = pt[rk * G4_PLD + dk];
          }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int d = t * 32 + (lane >> 1);
          const float bd = P.bias ? P.bias[nw0 + d] : 0.f;
          u32x4 o = {pack_bf16x2(e[t][0] + bd, e[t][1] + bd), pack_bf16x2(e[t][2] + bd, e[t][3] + bd), pack_bf16x2(e[t][4] + bd, e[t][5] + bd),
                     pack_bf16x2(e[t][6] + bd, e[t][7] + bd)};
          *(u32x4*)((uint16_t*)P.qkv_vt + ((size_t)(b * H + h) * 128 + d) * P.qkv_vt_ld + P.qkv_vt_pos0 + p0 + gk * 8) = o;
        }
      } else {
        uint16_t* const out = (qkind == 0 && P.qkv_k) ? (uint16_t*)P.qkv_k : (uint16_t*)P.C;
        const int out_ld = (qkind == 0 && P.qkv_k) ? P.qkv_k_ld : P.ldc;
        f32x4 cs[4][2], pv[4][2];
#pragma unroll
        for (int t = 0; t < 4; ++t) {                  // the block's RoPE rows: (cos, sin) pairs of this lane's 8 columns
          const float* rp = P.qkv_rope + (size_t)(p0 + t * 4 + (lane >> 4)) * 128 + c8;
          cs[t][0] = *(const f32x4*)rp; cs[t][1] = *(const f32x4*)(rp + 4);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int row = t * 4 + (lane >> 4);
          pv[t][0] = *(const f32x4*)(pt + row * G4_PLD + c8); pv[t][1] = *(const f32x4*)(pt + row * G4_PLD + c8 + 4);
        }
        float yy[4][8];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          f32x4 v0 = pv[t][0], v1 = pv[t][1];
          float ss = 0.f;
#pragma unroll
          for (int c_ = 0; c_ < 4; ++c_) { v0[c_] += bias0[c_]; v1[c_] += bias1[c_]; ss = __builtin_fmaf(v0[c_], v0[c_], ss); ss = __builtin_fmaf(v1[c_], v1[c_], ss); }
#ifdef LX_G4_ROW16_SHFL                                   // (measurement build only, tools/build_variant.sh: the ds_bpermute tree this replaced)
          ss += __shfl_xor(ss, 1, 64); ss += __shfl_xor(ss, 2, 64); ss += __shfl_xor(ss, 4, 64); ss += __shfl_xor(ss, 8, 64);
#else
          ss = row16_sum(ss);                        // (the 16 lanes of a row: DPP, bit-identical to the xor-shuffle tree it replaces)
#endif
          const float r = rsqrtf(ss * (1.0f / 128.0f) + 1e-6f);
          float x[8];
          float (&y)[8] = yy[t];
#pragma unroll
          for (int c_ = 0; c_ < 4; ++c_) { x[c_] = v0[c_] * r * nw0v[c_]; x[4 + c_] = v1[c_] * r * nw1v[c_]; }
#pragma unroll
          for (int q = 0; q < 4; ++q) {                // pairs (2q, 2q+1): out = x*cos + rot*sin, rot = (-x_odd, x_even)
            const float co = q < 2 ? cs[t][0][2 * q] : cs[t][1][2 * q - 4], si = q < 2 ? cs[t][0][2 * q + 1] : cs[t][1][2 * q - 3];
            y[2 * q] = x[2 * q] * co - x[2 * q + 1] * si;
            y[2 * q + 1] = x[2 * q + 1] * co + x[2 * q] * si;
          }
        }
        // the four row stores back to back (e4m3 q / k / V^T images for lx_attn_fwd_fp8 are written by the 8-wave kernels' epilogue only:
        // the form of it that lived here measured 0.2 % slower per image, profiles/r04f_attnfp8_ab.txt, and went in round 5)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int m = mb + t * 4 + (lane >> 4);
          const float (&y)[8] = yy[t];
          u32x4 o = {pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]), pack_bf16x2(y[4], y[5]), pack_bf16x2(y[6], y[7])};
          if (m < M) *(u32x4*)(out + (size_t)m * out_ld + ncol) = o;
        }
      }
      return;
    }
    if (bf16_out) {
      f32x4 pv[4][2];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int row = t * 4 + (lane >> 4);
        pv[t][0] = *(const f32x4*)(pt + row * G4_PLD + c8); pv[t][1] = *(const f32x4*)(pt + row * G4_PLD + c8 + 4);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int row = t * 4 + (lane >> 4), m = mb + row;
        f32x4 v0 = pv[t][0], v1 = pv[t][1];
        if (m < M && col_ok) {
#pragma unroll
          for (int c_ = 0; c_ < 4; ++c_) { v0[c_] += bias0[c_]; v1[c_] += bias1[c_]; }
          if (gelu0) { v0 = gelu_tanh4(v0); v1 = gelu_tanh4(v1); }
          // (fp16 operands: the store is the next GEMM's A operand -- fp16, nearest even, saturated)
          u32x4 o = {pack_op16x2<F16>(v0[0], v0[1], f16_mx), pack_op16x2<F16>(v0[2], v0[3], f16_mx), pack_op16x2<F16>(v1[0], v1[1], f16_mx),
                     pack_op16x2<F16>(v1[2], v1[3], f16_mx)};
          *(u32x4*)((uint16_t*)P.C + (size_t)m * P.ldc + ncol) = o;
          if constexpr (SPLIT) {
            if (P.epilogue & LX_EPI_SPLIT_BF16) {     // the rounding residual x - bf16(x), as bf16, c_lo_off columns further (gemm_epilogue)
              float r[8];
#pragma unroll
              for (int c_ = 0; c_ < 4; ++c_) {
                r[c_] = v0[c_] - bf16_to_f32(f32_to_bf16(v0[c_]));
                r[4 + c_] = v1[c_] - bf16_to_f32(f32_to_bf16(v1[c_]));
              }
              u32x4 ol = {pack_bf16x2(r[0], r[1]), pack_bf16x2(r[2], r[3]), pack_bf16x2(r[4], r[5]), pack_bf16x2(r[6], r[7])};
              *(u32x4*)((uint16_t*)P.C + (size_t)m * P.ldc + ncol + P.c_lo_off) = ol;
            }
          }
        }
      }
    } else {
      f32x4 pv[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) pv[t] = *(const f32x4*)(pt + (t * 2 + (lane >> 5)) * G4_PLD + c4);
      if (epi == LX_EPI_RESID_F32) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < 8; ++t) asm volatile("" : "+v"(res[t]), "+v"(gat[t]));
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int row = t * 2 + (lane >> 5), m = mb + row;
        f32x4 v = pv[t];
        if (m < M && col_ok) {
#pragma unroll
          for (int c_ = 0; c_ < 4; ++c_) v[c_] += bias0[c_];
          if (gelu0) v = gelu_tanh4(v);
          if (epi == LX_EPI_RESID_F32) {
            f32x4 o = res[t];
            if (P.gate) {
#pragma unroll
              for (int c_ = 0; c_ < 4; ++c_) o[c_] = __builtin_fmaf(gat[t][c_], v[c_], o[c_]);
            } else {
#pragma unroll
              for (int c_ = 0; c_ < 4; ++c_) o[c_] += v[c_];
            }
            v = o;
          }
          *(f32x4*)((float*)P.C + (size_t)m * P.ldc + ncol) = v;
        }
      }
    }
  };
#define G4_B(I, NX) block(std::integral_constant<int, I>{}, std::integral_constant<int, NX>{})
#define G4_K(I, NX) park(std::integral_constant<int, I>{}, std::integral_constant<int, NX>{})
  if (!split_tile) {                                   // a whole tile: its own straight line (a branch in the middle of it cost 3 % per launch)
    if (mw0 < M) put(std::integral_constant<int, 0>{});
    G4_B(0, 1); G4_B(1, 2); G4_B(2, 3); G4_B(3, 4); G4_B(4, 5); G4_B(5, 6); G4_B(6, 7); G4_B(7, -1);
  } else if constexpr (NP == 2) {
    if (part == 0) {                                   // parks 4-7, then owns 0-3
      if (mw0 + 64 < M) put(std::integral_constant<int, 4>{});
      G4_K(4, 5); G4_K(5, 6); G4_K(6, 7); G4_K(7, 0);
      exchange();
      G4_B(0, 1); G4_B(1, 2); G4_B(2, 3); G4_B(3, -1);
    } else {                                           // parks 0-3, then owns 4-7
      if (mw0 < M) put(std::integral_constant<int, 0>{});
      G4_K(0, 1); G4_K(1, 2); G4_K(2, 3); G4_K(3, 4);
      exchange();
      G4_B(4, 5); G4_B(5, 6); G4_B(6, 7); G4_B(7, -1);
    }
  } else {                                             // thirds (consecutive blocks of a sequence alternate between the two patches: i & 1)
    if (part == 0) {                                   // parks 3-7, then owns 0-2
      if (mw0 + 48 < M) put(std::integral_constant<int, 3>{});
      G4_K(3, 4); G4_K(4, 5); G4_K(5, 6); G4_K(6, 7); G4_K(7, 0);
      exchange();
      G4_B(0, 1); G4_B(1, 2); G4_B(2, -1);
    } else if (part == 1) {                            // parks 0-2, 7, 6, then owns 3-5
      if (mw0 < M) put(std::integral_constant<int, 0>{});
      G4_K(0, 1); G4_K(1, 2); G4_K(2, 7); G4_K(7, 6); G4_K(6, 3);
      exchange();
      G4_B(3, 4); G4_B(4, 5); G4_B(5, -1);
    } else {                                           // parks 0-5, then owns 6-7
      if (mw0 < M) put(std::integral_constant<int, 0>{});
      G4_K(0, 1); G4_K(1, 2); G4_K(2, 3); G4_K(3, 4); G4_K(4, 5); G4_K(5, 6);
      exchange();
      G4_B(6, 7); G4_B(7, -1);
    }
  }
#undef G4_B
#undef G4_K
  if constexpr (F16) { if (bf16_out && !qkv_tile) report_f16_overflow(f16_mx, P.f16_ovf); }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  G4_STAMP(5)
#endif
}


}  // namespace
