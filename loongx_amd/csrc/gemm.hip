// gemm.hip -- grouped bf16 GEMM with fused epilogues for gfx950 (MI355X).
//
//   C[m,n] = sum_k A[m,k] * W[n,k]       A: activations [M,K], W: nn.Linear weight [N,K]
//
// Replaces every nn.Linear of the DiT forward (reference src/flux/block.py:27-29,46-48,81-83,154-160,
// 258-265,302-333; src/flux/transformer.py:92-93,115,244) -- see include/lx.h.
//
// Design (MI355X-first, no CUDA lineage):
//  * workgroup = 8 waves (512 threads), macro tile BM x 256 x 64 (BM = 256 or 128), 1 workgroup / CU;
//  * operands go HBM/L2 -> LDS with global_load_lds (16 B / lane, no VGPR round trip), double buffered,
//    one barrier per K step;
//  * LDS tile rows are 128 B (64 bf16); the 16-B slot index is XORed with (row>>1)&7 so that the
//    ds_read_b128 lane groups of an MFMA fragment read hit 16 distinct slots (conflict-free); because
//    global_load_lds writes lane-linear, the swizzle is applied to the per-lane SOURCE address;
//  * v_mfma_f32_32x32x16_bf16 with the weight tile as the MFMA "A" operand, so the accumulator layout is
//    lane = output row m, registers = 4 consecutive output columns n -> vector epilogue loads/stores;
//  * per-wave tile (BM/2) x 64: 2 W-fragments + BM/64 X-fragments feed 2*BM/64 MFMAs per 16-deep k step;
//  * epilogue fuses bias, rank-r LoRA up-projection, GELU(tanh), and the gated residual accumulate
//    X += gate * y in fp32 (block.py:224-234,269-272,326-334);
//  * blockIdx -> tile map is XCD-aware: each of the 8 XCDs (private L2) gets a contiguous run of tiles,
//    ordered in 4-tile-tall column groups so co-resident tiles share A / W panels in that L2.
#include "common.h"
#include <type_traits>

#ifndef LX_ACC_AGPR
#define LX_ACC_AGPR 0
#endif

namespace {

constexpr int BN = 256;
constexpr int BK = 64;
constexpr int NTHREADS = 512;
constexpr int GROUP_M = 4;   // M-tile rows per column group of the tile order (4 / 8 / 16 measured identical)

int env_int(const char* name, int dflt) {
  const char* s = getenv(name);
  return s ? atoi(s) : dflt;
}

constexpr int MAX_SUB = 2 * LX_GEMM_MAX_GROUP;   // a problem may be split into a 256-row-tile part and a 128-row-tile tail

struct GemmArgs {
  lx_gemm_desc p[MAX_SUB];
  int tile_start[MAX_SUB + 1];
  int m_base[MAX_SUB];       // row of the original problem at which this (sub)problem starts (for the gate batch index)
  int n;
};
static_assert(2 * sizeof(GemmArgs) + 16 <= 4096, "lx_gemm_mixed_kernel takes two plans by value: the kernarg segment is 4 KiB");

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// ---- LoRA up-projection as ONE extra MFMA k-step per 4 ranks ---------------------------------------------------------
// t (= x . A_down^T, fp32, from lx_lora_down) and up are split into bf16 hi + lo parts and the 16 k-slots carry the four cross
// terms (hi*hi, hi*lo, lo*hi, lo*lo) of 4 ranks: fp32-class accuracy (2^-16 relative) at the cost of 2*MI MFMAs, instead of a
// scalar epilogue loop.
//   Every load of the step -- the up rows of the wave's 64 columns, and t of its BM/2 rows from up to four K-split slabs of
// lx_lora_down -- is issued before the first value is used: one memory round trip. (One slab at a time, one row block at a
// time, the phase was 16 dependent round trips: ~5 us per tile, and with the condition rows in every round of a launch that
// is ~5 us per ROUND: -5.2 % per denoise step when it went.) lora_issue only loads; lora_sum adds the slabs in slab order
// (((s0 + s1) + s2) + s3 ...); lora_apply converts and runs the MFMAs.
template <int MI>
__device__ __forceinline__ void lora_issue(const lx_gemm_desc& P, int n0, int mw0, int nw0, int l31, int r0, int sp0, f32x4 (&u4)[2],
                                           f32x4 (&sv)[MI][4]) {
  const int R = P.lora_r, nsplit = P.lora_nsplit;
  const int toff = R * min(n0 / max(P.lora_mod_cols, 1), P.lora_toff_max);
  const int nvalid = min(R - r0, 4);
  // 16-B vector loads when rank, strides and bases allow it (always, for the ranks peft is used with); else element loads
  const bool vec = ((R | P.lora_ldt | P.lora_split_stride | toff) & 3) == 0 && ((((uintptr_t)P.lora_t) | ((uintptr_t)P.lora_up)) & 15) == 0;
  const float* up[2];
  const float* tp[MI];
#pragma unroll
  for (int j = 0; j < 2; ++j) up[j] = P.lora_up + (size_t)min(nw0 + j * 32 + l31, P.N - 1) * R + r0;
#pragma unroll
  for (int i = 0; i < MI; ++i) tp[i] = P.lora_t + (size_t)min(mw0 + i * 32 + l31, P.M - 1) * P.lora_ldt + toff + r0;
  if (vec) {                          // ONE branch around all loads, not one per load: they must issue back to back
    if (sp0 == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j) u4[j] = *(const f32x4*)up[j];
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) sv[i][q] = *(const f32x4*)(tp[i] + (size_t)min(sp0 + q, nsplit - 1) * P.lora_split_stride);
  } else {
    auto ld4 = [&](const float* p) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (e < nvalid) v[e] = p[e];
      return v;
    };
    if (sp0 == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j) u4[j] = ld4(up[j]);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) sv[i][q] = ld4(tp[i] + (size_t)min(sp0 + q, nsplit - 1) * P.lora_split_stride);
  }
}

template <int MI>
__device__ __forceinline__ void lora_sum(const lx_gemm_desc& P, int sp0, const f32x4 (&sv)[MI][4], f32x4 (&t4)[MI]) {
  const int nsplit = P.lora_nsplit;
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      if (sp0 + q == 0) t4[i] = sv[i][q];
      else if (sp0 + q < nsplit) {
#pragma unroll
        for (int e = 0; e < 4; ++e) t4[i][e] += sv[i][q][e];
      }
    }
}

template <int MI>
__device__ __forceinline__ void lora_apply(const f32x4 (&u4)[2], const f32x4 (&t4)[MI], int lhi, f32x16 (&acc)[2][MI]) {
  bf16x8 wf[2], xf[MI];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    u32x4 w;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const uint16_t h0 = f32_to_bf16(u4[j][2 * e]), h1 = f32_to_bf16(u4[j][2 * e + 1]);
      w[e] = (uint32_t)h0 | ((uint32_t)h1 << 16);                                                    // slots 0-3: up_hi
      w[2 + e] = pack_bf16x2(u4[j][2 * e] - bf16_to_f32(h0), u4[j][2 * e + 1] - bf16_to_f32(h1));  // slots 4-7: up_lo
    }
    wf[j] = __builtin_bit_cast(bf16x8, w);
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const f32x4 t = t4[i];
    u32x2 h;
    if (lhi == 0) {       // k-slots 0-7 pair with t_hi, slots 8-15 (upper half-wave) with t_lo
      h[0] = pack_bf16x2(t[0], t[1]);
      h[1] = pack_bf16x2(t[2], t[3]);
    } else {
      float lo[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) lo[e] = t[e] - bf16_to_f32(f32_to_bf16(t[e]));
      h[0] = pack_bf16x2(lo[0], lo[1]);
      h[1] = pack_bf16x2(lo[2], lo[3]);
    }
    u32x4 x = {h[0], h[1], h[0], h[1]};
    xf[i] = __builtin_bit_cast(bf16x8, x);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < MI; ++i)
      acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], xf[i], acc[j][i], 0, 0, 0);
}

// Rank <= 4 and <= 4 slabs (the shipped adapters: r = 4, 4 slabs): the whole step is one batch of loads, and it is done at the
// START of the tile -- loads issued ahead of the prologue's operand DMA, MFMAs into the still-empty accumulators while that DMA
// is in flight -- so that its memory round trip hides under the DMA latency the tile waits for anyway.
__device__ __forceinline__ bool lora_in_prologue(const lx_gemm_desc& P) { return P.lora_t != nullptr && P.lora_r <= 4 && P.lora_nsplit <= 4; }

// Shared epilogue: LoRA MFMA step, LDS transpose, coalesced bias / GELU / gate / residual / store.
__device__ __forceinline__ float clamp_e4m3(float x) { return fminf(fmaxf(x, -448.f), 448.f); }
__device__ __forceinline__ uint32_t pack_fp8x4(float a, float b, float c, float d) {
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(clamp_e4m3(a), clamp_e4m3(b), 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(clamp_e4m3(c), clamp_e4m3(d), w, true);
  return (uint32_t)w;
}


// ---- LX_EPI_QKV: RMSNorm(128) + RoPE on the k / q columns, V^T image for the v columns, inside the projection's epilogue --------
// (block.py:60-99: attn.norm_q / norm_k, apply_rotary_emb; replaces the qkv_prep pass over the projected buffer: one read + one
//  write of 3 D columns per token, 22 us x 57 launches per denoise step at S = 2560.) A 256-column tile is two whole heads of
// one kind (qkv_d % 256 == 0); a wave holds 64 columns, so the sum of squares of a head's row is the sum of two waves' partial
// sums, exchanged through LDS once per tile. Everything is computed in fp32 on the accumulators: one bf16 rounding instead of
// the two of the separate pass.
__device__ __forceinline__ int qkv_vt_interleave(int key) {  // within every 16 keys: [0-3, 8-11, 4-7, 12-15] (= rowops.hip)
  return (key & ~15) | (((key >> 2) & 1) << 3) | (((key >> 3) & 1) << 2) | (key & 3);
}

template <int BM, int MI>
__device__ __forceinline__ void gemm_epilogue_qkv(const lx_gemm_desc& P, f32x16 (&acc)[2][MI], char* smem, int m0, int n0, int m_base,
                                                  int wave, int wm, int wn, int lane, int l31, int lhi) {
  const int M = P.M, D = P.qkv_d, L = P.rows_per_batch;
  const int kind = n0 / D;                     // 0: k, 1: v, 2: q (tile-uniform)
  const int mw0 = m0 + wm * (BM / 2), nw0 = n0 + wn * 64;
  constexpr int EP_LD = 68;
  float* patch = (float*)smem + wave * (32 * EP_LD);
  float* ssq = (float*)smem + 8 * (32 * EP_LD);          // [8 waves][BM / 2]: per-row partial sums of squares
  // bias in the accumulator layout: n = nw0 + j*32 + 8*rq + 4*lhi + c
  if (P.bias) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const f32x4 b = *(const f32x4*)(P.bias + nw0 + j * 32 + rq * 8 + 4 * lhi);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[j][i][rq * 4 + c] += b[c];
      }
  }
  __syncthreads();                                   // every wave is done with the operand tiles
  auto to_patch = [&](int i) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        f32x4 v = {acc[j][i][rq * 4], acc[j][i][rq * 4 + 1], acc[j][i][rq * 4 + 2], acc[j][i][rq * 4 + 3]};
        *(f32x4*)(patch + l31 * EP_LD + j * 32 + rq * 8 + 4 * lhi) = v;
      }
    __builtin_amdgcn_wave_barrier();
  };
  const bool f8 = P.qkv_q8 != nullptr;              // e4m3 images for the fp8 attention kernel instead of the bf16 outputs
  if (kind == 1 && f8) {
    // v -> byte V^T image: a 32-key block is one half of a 64-key tile row; in the f8f6f4 operand order (byte j = g*32 + p holds key
    // (p>>4)*32 + 8*((p&15)>>2) + 4g + (p&3)) that half is bytes [half*16, +16) of each 32-byte group g. Lane = head dim: per block
    // two 16-byte stores per lane; the patch is read down a column (lanes on consecutive addresses: conflict-free).
    const int h = (nw0 - D) >> 7, d0 = (nw0 - D) & 127, H = D >> 7;
    const float vs = P.qkv_v_scale;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int mb = mw0 + i * 32;
      if (mb >= M) continue;
      to_patch(i);
      const int gm = m_base + mb, b = gm / L, p0 = gm - b * L;
      uint8_t* vtb = (uint8_t*)P.qkv_vt8 + ((size_t)(b * H + h) * 128 + d0 + lane) * P.qkv_vt_ld + P.qkv_vt_pos0 + (p0 & ~63) + ((p0 >> 5) & 1) * 16;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        float e[16];
#pragma unroll
        for (int pp = 0; pp < 16; ++pp) e[pp] = patch[(8 * (pp >> 2) + 4 * g + (pp & 3)) * EP_LD + lane] * vs;
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = pack_fp8x4(e[4 * q], e[4 * q + 1], e[4 * q + 2], e[4 * q + 3]);
        *(u32x4*)(vtb + g * 32) = o;
      }
      __builtin_amdgcn_wave_barrier();
    }
    return;
  }
  if (kind == 1) {
    // v: 32 keys x 64 head dims per block -> V^T rows of 32 slots (64 B), 16 B per lane
    const int h = (nw0 - D) >> 7, d0 = (nw0 - D) & 127, H = D >> 7;
    const int dl = lane >> 2, g = lane & 3;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int mb = mw0 + i * 32;
      if (mb >= M) continue;                         // M % 32 == 0: a block is whole or absent
      to_patch(i);
      const int gm = m_base + mb, b = gm / L, p0 = gm - b * L;
      uint16_t* vtb = (uint16_t*)P.qkv_vt + ((size_t)(b * H + h) * 128 + d0) * P.qkv_vt_ld + P.qkv_vt_pos0 + p0 + g * 8;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int d = it * 16 + dl;
        float e[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) e[k] = patch[qkv_vt_interleave(g * 8 + k) * EP_LD + d];
        u32x4 o = {pack_bf16x2(e[0], e[1]), pack_bf16x2(e[2], e[3]), pack_bf16x2(e[4], e[5]), pack_bf16x2(e[6], e[7])};
        *(u32x4*)(vtb + (size_t)d * P.qkv_vt_ld) = o;
      }
      __builtin_amdgcn_wave_barrier();
    }
    return;
  }
  // k / q: partial sums of squares of this wave's 64 columns, row = lane & 31 of each block
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) ss = __builtin_fmaf(acc[j][i][r], acc[j][i][r], ss);
    ss += __shfl_xor(ss, 32, 64);
    if (lhi == 0) ssq[wave * (BM / 2) + i * 32 + l31] = ss;
  }
  const float* __restrict__ nw = kind == 2 ? P.qkv_norm_q : P.qkv_norm_k;
  const int c8 = (lane & 7) * 8;
  const int hd = (nw0 & 127) + c8;                   // first of this lane's 8 columns within the head
  f32x4 w0 = *(const f32x4*)(nw + hd), w1 = *(const f32x4*)(nw + hd + 4);
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(w0), "+v"(w1)::"memory");
  __syncthreads();                                   // both halves of every head's sums are in LDS
  const int ncol = nw0 + c8;                         // (k columns start at 0: the same column in a separate key image)
  uint16_t* const out = (kind == 0 && P.qkv_k) ? (uint16_t*)P.qkv_k : (uint16_t*)P.C;
  const int out_ld = (kind == 0 && P.qkv_k) ? P.qkv_k_ld : P.ldc;
  uint8_t* const out8 = f8 ? (kind == 0 ? (uint8_t*)P.qkv_k8 : (uint8_t*)P.qkv_q8) + (ncol - kind * D) : nullptr;
  const float sc8 = kind == 0 ? P.qkv_k_scale : P.qkv_q_scale;
  const float* own = ssq + wave * (BM / 2);
  const float* oth = ssq + (wave ^ 1) * (BM / 2);
  // RoPE rows of a 32-row block: 8 x 16 B per lane. vmcnt is one in-order queue of loads AND stores (see gemm_epilogue): the rows
  // of block i+1 are requested BEFORE block i's stores, so that waiting for them (vmcnt(4): only the four stores behind them may
  // still be in flight) never waits for a store's acknowledgement. The loads are inline asm: hipcc's own wait-count pass, which
  // falls back to vmcnt(0) behind any branch, does not see them, and the counted waits below are the only ones. Blocks are whole
  // or absent (M % 32 == 0) and a table row index is always < rows_per_batch, so loads and counts need no conditions.
  f32x4 cs[2][4][2];
  auto rope_rows = [&](int i, f32x4 (&c)[4][2]) {
    const int gm = m_base + mw0 + i * 32, b = gm / L, p0 = gm - b * L;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float* rp = P.qkv_rope + (size_t)(p0 + t * 8 + (lane >> 3)) * 128 + hd;
      asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16" : "=&v"(c[t][0]), "=&v"(c[t][1]) : "v"(rp) : "memory");
    }
  };
  rope_rows(0, cs[0]);
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int mb = mw0 + i * 32;
    if (mb >= M) break;                              // (the loads in flight land in dead registers)
    f32x4 (&cur)[4][2] = cs[i & 1];
    to_patch(i);
    if (i == 0)
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(cur[0][0]), "+v"(cur[0][1]), "+v"(cur[1][0]), "+v"(cur[1][1]), "+v"(cur[2][0]), "+v"(cur[2][1]), "+v"(cur[3][0]), "+v"(cur[3][1])::"memory");
    else
      asm volatile("s_waitcnt vmcnt(4)" : "+v"(cur[0][0]), "+v"(cur[0][1]), "+v"(cur[1][0]), "+v"(cur[1][1]), "+v"(cur[2][0]), "+v"(cur[2][1]), "+v"(cur[3][0]), "+v"(cur[3][1])::"memory");
    if (i + 1 < MI) rope_rows(i + 1, cs[(i + 1) & 1]);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int row = t * 8 + (lane >> 3), m = mb + row;
      const f32x4 v0 = *(const f32x4*)(patch + row * EP_LD + c8);
      const f32x4 v1 = *(const f32x4*)(patch + row * EP_LD + c8 + 4);
      const float r = rsqrtf((own[i * 32 + row] + oth[i * 32 + row]) * (1.0f / 128.0f) + 1e-6f);
      float x[8], y[8];
#pragma unroll
      for (int c = 0; c < 4; ++c) { x[c] = v0[c] * r * w0[c]; x[4 + c] = v1[c] * r * w1[c]; }
#pragma unroll
      for (int q = 0; q < 4; ++q) {                  // pairs (2q, 2q+1): out = x*cos + rot*sin, rot = (-x_odd, x_even)
        const float co = q < 2 ? cur[t][0][2 * q] : cur[t][1][2 * q - 4], si = q < 2 ? cur[t][0][2 * q + 1] : cur[t][1][2 * q - 3];
        y[2 * q] = x[2 * q] * co - x[2 * q + 1] * si;
        y[2 * q + 1] = x[2 * q + 1] * co + x[2 * q] * si;
      }
      if (f8) {                                      // (tile-uniform; the same four stores per block as the bf16 form: the vmcnt counts hold)
        u32x2 o8 = {pack_fp8x4(y[0] * sc8, y[1] * sc8, y[2] * sc8, y[3] * sc8), pack_fp8x4(y[4] * sc8, y[5] * sc8, y[6] * sc8, y[7] * sc8)};
        *(u32x2*)(out8 + (size_t)m * P.qkv_ld8) = o8;
      } else {
        u32x4 o = {pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]), pack_bf16x2(y[4], y[5]), pack_bf16x2(y[6], y[7])};
        *(u32x4*)(out + (size_t)m * out_ld + ncol) = o;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

template <int BM, int MI, bool SPLIT = false, bool FP8 = false>
__device__ __forceinline__ void gemm_epilogue(const lx_gemm_desc& P, f32x16 (&acc)[2][MI], char* smem, int m0, int n0, int m_base,
                                              int wave, int wm, int wn, int lane, int l31, int lhi, bool lora_done, int i_begin = 0, int i_end = MI) {
  // [i_begin, i_end): the 32-row blocks of each wave's tile that this workgroup finishes (all of them, except in the pair kernel)
  const int M = P.M, N = P.N;
  // ---- epilogue ----------------------------------------------------------------------------------
  // acc[j][i][r]: m = m0 + wm*BM/2 + i*32 + l31 ; n = n0 + wn*64 + j*32 + 8*(r>>2) + 4*lhi + (r&3)
  const int epi = P.epilogue & 0xff;
  const bool do_gelu = (P.epilogue & LX_EPI_GELU) != 0;
  const int mw0 = m0 + wm * (BM / 2);          // first row of this wave's tile
  const int nw0 = n0 + wn * 64;                // first column of this wave's tile

  // (1) LoRA up-projection (see lora_issue / lora_apply below). When rank and slab count fit one batch the whole step has
  //     already been done in the tile's prologue (lora_in_prologue) and there is nothing to do here.
  if (P.lora_t != nullptr && !lora_done) {
    const int R = P.lora_r, nsplit = P.lora_nsplit;
    for (int r0 = 0; r0 < R; r0 += 4) {
      f32x4 u4[2], t4[MI];
      for (int sp0 = 0; sp0 < nsplit; sp0 += 4) {          // K-split partial slabs from lx_lora_down, four per round trip
        f32x4 sv[MI][4];
        lora_issue<MI>(P, n0, mw0, nw0, l31, r0, sp0, u4, sv);
        lora_sum<MI>(P, sp0, sv, t4);
      }
      lora_apply<MI>(u4, t4, lhi, acc);
    }
  }

  if constexpr (!SPLIT && !FP8) {
    if ((P.epilogue & LX_EPI_QKV) && n0 < 3 * P.qkv_d) {           // tile-uniform: the projection tiles of a (fused) launch
      gemm_epilogue_qkv<BM, MI>(P, acc, smem, m0, n0, m_base, wave, wm, wn, lane, l31, lhi);
      return;
    }
  }

  // (2) transpose each 32x64 accumulator block through a wave-private LDS patch so that every global access of
  //     the epilogue (bias, gate, residual read-modify-write, stores) is a coalesced 16-B-per-lane row access.
  //     vmcnt counts loads and stores in one in-order queue: a load issued behind a store cannot be waited for without
  //     waiting for that store's acknowledgement from L2 first. So no load may sit between the stores: the bias (a function
  //     of the column only) is loaded once per tile, and the residual / gate rows of a 32-row block are all loaded before
  //     the block's first store (one exposed store latency per block instead of one per 4-row group: -5...-9 us per tile).
  __syncthreads();                                   // every wave is done with the operand tiles
  constexpr int EP_LD = 68;                          // fp32 row stride of the patch (64 + 4 pad)
  float* patch = (float*)smem + wave * (32 * EP_LD);
  const bool bf16_out = epi == LX_EPI_STORE_BF16 || (FP8 && epi == LX_EPI_STORE_FP8);      // the 8-columns-per-lane store shape
  const int c8 = (lane & 7) * 8, c4 = (lane & 15) * 4;
  const int ncol = nw0 + (bf16_out ? c8 : c4);       // first of this lane's 8 (bf16 store) or 4 (fp32 paths) columns
  const bool col_ok = ncol < N;
  f32x4 bias0 = {0.f, 0.f, 0.f, 0.f}, bias1 = {0.f, 0.f, 0.f, 0.f};
  if (P.bias && col_ok) {
    bias0 = *(const f32x4*)(P.bias + ncol);
    if (bf16_out) bias1 = *(const f32x4*)(P.bias + ncol + 4);
  }
  // The loads above sit under a condition, and hipcc's wait-count pass then re-waits vmcnt(0) at every later use of their
  // registers -- which, inside the store loops below, means waiting for the previous store after all. Wait here, once, and
  // hand the values on through an empty asm so that they are no longer "results of a load" to the compiler.
  // fp8 GEMMs: the accumulators are in units of 1 / (activation scale x weight-row scale): per-column de-scale first
  f32x4 cs0 = {1.f, 1.f, 1.f, 1.f}, cs1 = {1.f, 1.f, 1.f, 1.f};
  if constexpr (FP8) {
    if (P.col_scale && col_ok) {
      cs0 = *(const f32x4*)(P.col_scale + ncol);
      if (bf16_out) cs1 = *(const f32x4*)(P.col_scale + ncol + 4);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(bias0), "+v"(bias1), "+v"(cs0), "+v"(cs1)::"memory");
  const bool gelu0 = do_gelu && ncol >= P.gelu_col_start;      // gelu_col_start is a multiple of 8: one answer per lane
  auto to_patch = [&](int i) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        f32x4 v = {acc[j][i][rq * 4], acc[j][i][rq * 4 + 1], acc[j][i][rq * 4 + 2], acc[j][i][rq * 4 + 3]};
        *(f32x4*)(patch + l31 * EP_LD + j * 32 + rq * 8 + 4 * lhi) = v;
      }
    __builtin_amdgcn_wave_barrier();
  };
  // One specialised block loop per output kind (the kind is wave-uniform): with the three kinds inside one loop, the waits
  // hipcc places at the control-flow joins are vmcnt(0) again.
  if (bf16_out) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      if (i < i_begin || i >= i_end) continue;
      const int mb = mw0 + i * 32;
      to_patch(i);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int row = t * 8 + (lane >> 3), m = mb + row;
        f32x4 v0 = *(const f32x4*)(patch + row * EP_LD + c8);
        f32x4 v1 = *(const f32x4*)(patch + row * EP_LD + c8 + 4);
        if (m < M && col_ok) {
          if constexpr (FP8) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { v0[c] *= cs0[c]; v1[c] *= cs1[c]; }
          }
#pragma unroll
          for (int c = 0; c < 4; ++c) { v0[c] += bias0[c]; v1[c] += bias1[c]; }
          if (gelu0) { v0 = gelu_tanh4(v0); v1 = gelu_tanh4(v1); }
          if constexpr (FP8) {
            if (epi == LX_EPI_STORE_FP8) {        // e4m3 output (x out_scale): the A operand of the next fp8 GEMM
              const float os = P.out_scale;
              *(u32x2*)((uint8_t*)P.C + (size_t)m * P.ldc + ncol) = u32x2{pack_fp8x4(v0[0] * os, v0[1] * os, v0[2] * os, v0[3] * os),
                                                                          pack_fp8x4(v1[0] * os, v1[1] * os, v1[2] * os, v1[3] * os)};
              continue;
            }
          }
          u32x4 o = {pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v0[2], v0[3]), pack_bf16x2(v1[0], v1[1]), pack_bf16x2(v1[2], v1[3])};
          *(u32x4*)((uint16_t*)P.C + (size_t)m * P.ldc + ncol) = o;
          if constexpr (SPLIT) {
            // precise mode (LX_EPI_SPLIT_BF16): the rounding residual x - bf16(x), itself rounded to bf16, goes c_lo_off columns
            // further: hi + lo carries 16 mantissa bits of x to the consumer GEMM (which multiplies both, k_segs >= 2)
            if (P.epilogue & LX_EPI_SPLIT_BF16) {
              float r[8];
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                r[c] = v0[c] - bf16_to_f32(f32_to_bf16(v0[c]));
                r[4 + c] = v1[c] - bf16_to_f32(f32_to_bf16(v1[c]));
              }
              u32x4 ol = {pack_bf16x2(r[0], r[1]), pack_bf16x2(r[2], r[3]), pack_bf16x2(r[4], r[5]), pack_bf16x2(r[6], r[7])};
              *(u32x4*)((uint16_t*)P.C + (size_t)m * P.ldc + ncol + P.c_lo_off) = ol;
            }
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  } else if (epi == LX_EPI_RESID_F32) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      if (i < i_begin || i >= i_end) continue;
      const int mb = mw0 + i * 32;
      // residual + gate rows of this block, issued before the patch is even written
      f32x4 res[8], gat[8];
      // batch (= gate row) of each of the block's rows: one wave-uniform division per block when a batch has >= 32 rows (then the
      // block straddles at most one batch boundary), instead of a ~25-instruction integer division per row group and lane
      const int rpb = P.rows_per_batch;
      const int b_first = (m_base + mb) / rpb, rem_first = (m_base + mb) - b_first * rpb;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int rr = t * 4 + (lane >> 4), m = mb + rr;
        if (m < M && col_ok) {
          res[t] = *(const f32x4*)((const float*)P.C + (size_t)m * P.ldc + ncol);
          const int b = rpb >= 32 ? b_first + (rem_first + rr >= rpb ? 1 : 0) : (m_base + m) / rpb;
          if (P.gate) gat[t] = *(const f32x4*)(P.gate + (size_t)b * P.gate_ld + ncol);
        }
      }
      to_patch(i);
      // same reason as for the bias: one explicit wait for the block's rows, none in the store loop
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int t = 0; t < 8; ++t) asm volatile("" : "+v"(res[t]), "+v"(gat[t]));
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int row = t * 4 + (lane >> 4), m = mb + row;
        f32x4 v = *(const f32x4*)(patch + row * EP_LD + c4);
        if (m < M && col_ok) {
          if constexpr (FP8) {
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] *= cs0[c];
          }
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] += bias0[c];
          if (gelu0) v = gelu_tanh4(v);
          f32x4 o = res[t];
          if (P.gate) {
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = __builtin_fmaf(gat[t][c], v[c], o[c]);      // explicit: not left to the contraction heuristics
          } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] += v[c];
          }
          *(f32x4*)((float*)P.C + (size_t)m * P.ldc + ncol) = o;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  } else {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      if (i < i_begin || i >= i_end) continue;
      const int mb = mw0 + i * 32;
      to_patch(i);
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int row = t * 4 + (lane >> 4), m = mb + row;
        f32x4 v = *(const f32x4*)(patch + row * EP_LD + c4);
        if (m < M && col_ok) {
          if constexpr (FP8) {
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] *= cs0[c];
          }
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] += bias0[c];
          if (gelu0) v = gelu_tanh4(v);
          *(f32x4*)((float*)P.C + (size_t)m * P.ldc + ncol) = v;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

template <int BM>
constexpr int gemm_lds_bytes() { return BM == 128 ? 3 * (128 * BK * 2) + 3 * (BN * BK * 2) : 2 * (256 * BK * 2) + 2 * (BN * BK * 2); }

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

// K tiles [kt0, kt1) of output tile (m0, n0) accumulated into acc (which the caller has cleared). `after_issue` runs between the
// issue of the prologue's operand DMA and the wait for it. `smem` = the workgroup's LDS
// buffer (gemm_lds_bytes<BM>() bytes, 1 KiB aligned). On return no wave reads the operand rings any more.
template <int BM, bool SPLIT = false, class F>
__device__ __forceinline__ void gemm_mainloop(const lx_gemm_desc& P, const int m0, const int n0, const int tn, const int kt0, const int kt1,
                                              char* smem, f32x16 (&acc)[2][BM / 64], const int tid, F&& after_issue) {
  constexpr int MI = BM / 64;               // 32-row m-blocks per wave
  constexpr int A_BYTES = BM * BK * 2;
  constexpr int W_BYTES = BN * BK * 2;
  // LDS rings. The activation operand A is L2/MALL-hot (just written by the previous kernel); the weight operand W streams
  // cold from HBM and needs more lead (measured: long-K GEMMs lose 21-23 % with a single K tile of DMA in flight).
  //   BM=128: A ring 3 x 16 KiB + W ring 3 x 32 KiB = 144 KiB: two K tiles of lead (long-K ff.net.2 / proj_out GEMMs:
  //           cold-weight penalty 21-23 % -> 0).
  //   BM=256: 2 x (32 + 32) KiB. A 3-deep W ring (160 KiB total) was measured: no gain at K=3072 (in-box A/B 43.6 vs
  //           43.2 ms per step), so the wide-N GEMMs keep two stages.
  // In-flight DMA is tracked with counted s_waitcnt vmcnt + a raw s_barrier (a __syncthreads() would drain it).
  constexpr int NSA = BM == 128 ? 3 : 2;
  constexpr int NSW = BM == 128 ? 3 : 2;
  constexpr int W_BASE = NSA * A_BYTES;
  constexpr int WAIT_STEADY = BM == 128 ? MI + 4 : (NSW == 3 ? 4 : 0);   // DMA instructions allowed in flight across the K-tile barrier
  static_assert(W_BASE + NSW * W_BYTES <= 160 * 1024, "LDS budget");
  static_assert(W_BASE + NSW * W_BYTES >= 8 * 32 * 68 * 4, "epilogue patch must fit");
  static_assert(W_BASE + NSW * W_BYTES == gemm_lds_bytes<BM>(), "gemm_lds_bytes out of sync with the ring layout");

  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int M = P.M, N = P.N, K = P.K;
  const bool w_tiled = (P.epilogue & LX_W_TILED) != 0;

  // ---- global->LDS staging (LDS-DMA) -----------------------------------------------------------------
  // one DMA instruction moves 1 KiB per wave = 8 tile rows of 128 B; lane -> (row = lane>>3, slot = lane&7).
  // Buffer addressing (SRSRC = this tile's operand origin, per-lane 32-bit byte offset in voffset, K position in soffset):
  // measured against flat-global 64-bit per-lane addresses in the same loop (tools/ubench/loop_rate): -260 stall cycles and
  // -6.8 % wall per K tile; the SGPR-base + 32-bit-offset global form is slower than either.
  uint32_t aoff[MI], woff[4];
  {
    const int rsub = lane >> 3, pslot = lane & 7;
#pragma unroll
    for (int j = 0; j < MI; ++j) {
      const int row = (j * 8 + wave) * 8 + rsub;
      const int lslot = pslot ^ ((row >> 1) & 7);
      aoff[j] = (uint32_t)((min(m0 + row, M - 1) - m0) * P.lda + lslot * 8) * 2u;
    }
    // W: either nn.Linear row-major [N,K], or (LX_W_TILED) pre-tiled at load time into the LDS image itself:
    // [N/256][K/64] blocks of 32 KiB, rows of 128 B with the XOR swizzle already applied, so a stage is ONE
    // contiguous 32 KiB read (DRAM-page / TLB friendly when the weights stream cold from HBM) copied verbatim.
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (w_tiled) {
        woff[j] = (uint32_t)((j * 8 + wave) * 512 + lane * 8) * 2u;
      } else {
        const int row = (j * 8 + wave) * 8 + rsub;
        const int lslot = pslot ^ ((row >> 1) & 7);
        woff[j] = (uint32_t)((min(n0 + row, N - 1) - n0) * P.ldw + lslot * 8) * 2u;
      }
    }
  }
  // Split-bf16 ("precise") problems run k_segs passes over K in ONE accumulation: segment 0 = A_hi x W_hi, 1 = A_lo x W_hi
  // (A_lo lives a_lo_off columns after A_hi), 2 = A_hi x W_lo (W is then [N, 2K] = [W_hi | W_lo]). K-tile index t of the loop
  // -> (segment, tile within the segment) -> source offsets; for !SPLIT the two maps below are the identity.
  const int nk1 = K / BK;
  const int kw_tiles = SPLIT && P.k_segs == 3 ? 2 * nk1 : nk1;      // K tiles per weight row block
  const __bf16* a_org = (const __bf16*)P.A + (size_t)m0 * P.lda;
  const __bf16* w_org = w_tiled ? (const __bf16*)P.W + ((size_t)tn * kw_tiles) * (BN * BK) : (const __bf16*)P.W + (size_t)n0 * P.ldw;
  const lx_rsrc_t rs_a = lx_make_rsrc(a_org), rs_w = lx_make_rsrc(w_org);
  const int w_kstride_b = (w_tiled ? BN * BK : BK) * 2;       // bytes between consecutive K tiles of the W operand
  auto a_soff = [&](int t) -> int {
    if constexpr (!SPLIT) return t * (BK * 2);
    else {
      const int seg = t >= 2 * nk1 ? 2 : (t >= nk1 ? 1 : 0);
      return (t - seg * nk1) * (BK * 2) + (seg == 1 ? P.a_lo_off * 2 : 0);
    }
  };
  auto w_soff = [&](int t) -> int {
    if constexpr (!SPLIT) return t * w_kstride_b;
    else {
      const int seg = t >= 2 * nk1 ? 2 : (t >= nk1 ? 1 : 0);
      return (t - seg * nk1 + (seg == 2 ? nk1 : 0)) * w_kstride_b;
    }
  };
  auto stage_a = [&](int kt, int slot) {
    char* base = smem + slot * A_BYTES;
    const int so = a_soff(kt0 + kt);
#pragma unroll
    for (int j = 0; j < MI; ++j)
      lx_buf_to_lds(rs_a, (lptr_t)(base + (j * 8 + wave) * 1024), aoff[j], so);
  };
  auto stage_w = [&](int kt, int slot) {
    char* base = smem + W_BASE + slot * W_BYTES;
    const int so = w_soff(kt0 + kt);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      lx_buf_to_lds(rs_w, (lptr_t)(base + (j * 8 + wave) * 1024), woff[j], so);
  };

  // ---- fragment read offsets -----------------------------------------------------------------------
  const int sw = (l31 >> 1) & 7;
  int slot_off[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) slot_off[ks] = ((ks * 2 + lhi) ^ sw) * 16;
  const int a_row_off = (wm * (BM / 2) + l31) * 128;
  const int w_row_off = (wn * 64 + l31) * 128;

  // ---- main loop: software pipelined ------------------------------------------------------------------
  // Fragment registers are double buffered (set A / set B alternate over the four 16-deep k steps of a K tile):
  // the LDS reads of step s+1 are issued BEFORE the MFMAs of step s.  The single barrier of a K tile sits between
  // steps 2 and 3, where every wave still holds 8 MFMAs of ready work: behind it the DMA of tile kt+2 is issued
  // into the buffer tile kt just vacated and the first fragments of tile kt+1 are fetched under step 3's MFMAs.
  auto load_frags = [&](int sa, int sw_, int ks, bf16x8 (&wf)[2], bf16x8 (&xf)[MI]) {
    const char* pa = smem + sa * A_BYTES + a_row_off + slot_off[ks];
    const char* pw = smem + W_BASE + sw_ * W_BYTES + w_row_off + slot_off[ks];
#pragma unroll
    for (int j = 0; j < 2; ++j) wf[j] = *(const bf16x8*)(pw + j * 32 * 128);
#pragma unroll
    for (int i = 0; i < MI; ++i) xf[i] = *(const bf16x8*)(pa + i * 32 * 128);
  };
  auto mma_j = [&](int j, const bf16x8 (&wf)[2], const bf16x8 (&xf)[MI]) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      // Accumulators live in AGPRs ("a" constraint; hipcc's own choice is arch VGPRs). Measured (tools/ubench/agpr_rate):
      // with LDS-DMA running on the CU, a K tile of MFMAs costs 1.36 us with AGPR accumulators vs 1.74 us with VGPR ones.
      if (LX_ACC_AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[j][i]) : "v"(wf[j]), "v"(xf[i]));
      else acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], xf[i], acc[j][i], 0, 0, 0);
    }
  };
  const int nkt = kt1 - kt0;                          // K tiles of this segment; `kt` below counts from kt0
  bf16x8 wfA[2], xfA[MI], wfB[2], xfB[MI];
  // prologue: A tiles 0..NSA-1 and W tiles 0..NSW-1 in (A0 W0 A1 W1 [A2] W2) order; wait only for tile 0
  {
    stage_a(0, 0);
    stage_w(0, 0);
    if (nkt > 1) { stage_a(1, 1); stage_w(1, 1); }
    if (nkt > 2) { if constexpr (NSA > 2) stage_a(2, 2); if constexpr (NSW > 2) stage_w(2, 2); }
    after_issue();                                            // work that fits under the DMA latency (LoRA step)
    if (nkt > 2 && NSW > 2) {
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // BM=128: A1 W1 A2 W2 / BM=256: A1 W1 W2 may stay in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    static_assert((BM == 128 && 2 * (MI + 4) == 12) || (BM == 256 && MI + 4 + 4 == 12), "prologue wait count");
  }
  __syncthreads();
  load_frags(0, 0, 0, wfA, xfA);
  // One k step = [MI MFMAs] [6 LDS reads for the NEXT step] [MI MFMAs]; sched_barrier(0) pins that order.  The reads sit
  // in the middle of an MFMA group so that the wait hipcc places in front of a group's first MFMA only ever covers
  // reads issued a whole group earlier (it is conservative across the loop back-edge and would otherwise stall on
  // the reads just issued).
#define LX_STEP(CUR_W, CUR_X, NEXT_STMT)            \
  __builtin_amdgcn_s_setprio(1);                    \
  mma_j(0, CUR_W, CUR_X);                           \
  __builtin_amdgcn_s_setprio(0);                    \
  __builtin_amdgcn_sched_barrier(0);                \
  NEXT_STMT;                                        \
  __builtin_amdgcn_sched_barrier(0);                \
  __builtin_amdgcn_s_setprio(1);                    \
  mma_j(1, CUR_W, CUR_X);                           \
  __builtin_amdgcn_s_setprio(0);                    \
  __builtin_amdgcn_sched_barrier(0);
  int ca = 0, cw = 0;                                  // ring slots of tile kt
  for (int kt = 0; kt < nkt; ++kt) {
    const int na = ca + 1 == NSA ? 0 : ca + 1;
    const int nw = cw + 1 == NSW ? 0 : cw + 1;
    LX_STEP(wfA, xfA, load_frags(ca, cw, 1, wfB, xfB))
    LX_STEP(wfB, xfB, load_frags(ca, cw, 2, wfA, xfA))
    LX_STEP(wfA, xfA, load_frags(ca, cw, 3, wfB, xfB))
    // Tile kt+1 (A and W) must have landed: everything older than the last WAIT_STEADY DMA instructions this wave issued
    // (= the W [and A] pieces of tile kt+2) is then complete. My LDS reads of tile kt are done (both k-steps are in
    // registers). Then the raw barrier makes that true for every wave.
    if (kt + 2 < nkt && WAIT_STEADY > 0) {
      if constexpr (WAIT_STEADY == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // Role split: waves w and w+4 share a SIMD. The lower half issues its DMA pieces right behind the barrier while
    // its SIMD partner runs MFMAs; the upper half runs step 3 first and issues its pieces afterwards. (Issuing from
    // both at once leaves the matrix pipe idle while the LDS-DMA pieces issue: measured +5 %.)  A goes first, W second:
    // the counted wait above relies on that order.
    if (wm == 0) {
      if (kt + NSA < nkt) stage_a(kt + NSA, ca);       // into the slots tile kt just vacated
      if (kt + NSW < nkt) stage_w(kt + NSW, cw);
    }
    __builtin_amdgcn_sched_barrier(0);
    LX_STEP(wfB, xfB, if (kt + 1 < nkt) load_frags(na, nw, 0, wfA, xfA))
    if (wm == 1) {
      if (kt + NSA < nkt) stage_a(kt + NSA, ca);
      if (kt + NSW < nkt) stage_w(kt + NSW, cw);
    }
    __builtin_amdgcn_sched_barrier(0);
    ca = na;
    cw = nw;
  }
#undef LX_STEP
  // the inline-asm MFMAs are opaque to the hazard recogniser: cover MFMA write -> v_accvgpr_read by hand (18 wait states)
  if (LX_ACC_AGPR) asm volatile("s_nop 15\n s_nop 7" ::: "memory");
}

template <int MI>
__device__ __forceinline__ void acc_clear(f32x16 (&acc)[2][MI]) {
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
}

// One whole output tile. `pid` = index of this workgroup among the launch's tiles of height BM.
template <int BM, bool SPLIT = false>
__device__ __forceinline__ void gemm_tile(const GemmArgs& args, const int pid, char* smem) {
  constexpr int MI = BM / 64;
  // ---- XCD-aware block -> tile map: each XCD (pid & 7) owns a contiguous run of the tile order ----
  const int total = args.tile_start[MAX_SUB];     // plan_add keeps every entry past the last problem equal to the total
  int lid;
  {
    const int q = total >> 3, r = total & 7;
    const int xcd = pid & 7, inx = pid >> 3;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + inx;
  }
  const int g = tile_group(args, lid);
  const lx_gemm_desc P = args.p[g];            // by value: one batch of scalar loads
  int tm, tn;
  tile_coords<BM>(P, lid - args.tile_start[g], tm, tn);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  f32x16 acc[2][MI];
  acc_clear<MI>(acc);
  const int wm = wave >> 2, wn = wave & 3, l31 = lane & 31, lhi = lane >> 5;
  const int m0 = tm * BM, n0 = tn * BN;
  const bool lora_early = lora_in_prologue(P);
  f32x4 u4[2], sv[MI][4];
  if (lora_early) lora_issue<MI>(P, n0, m0 + wm * (BM / 2), n0 + wn * 64, l31, 0, 0, u4, sv);
  const int nkt = SPLIT ? (P.K / BK) * max(P.k_segs, 1) : P.K / BK;
  gemm_mainloop<BM, SPLIT>(P, m0, n0, tn, 0, nkt, smem, acc, tid, [&]() {
    if (lora_early) {
      f32x4 t4[MI];
      lora_sum<MI>(P, 0, sv, t4);
      lora_apply<MI>(u4, t4, lhi, acc);
    }
  });
  gemm_epilogue<BM, MI, SPLIT>(P, acc, smem, m0, n0, args.m_base[g], wave, wm, wn, lane, l31, lhi, lora_early);
}

template <int BM>
__global__ __launch_bounds__(NTHREADS) void lx_gemm_kernel(const GemmArgs args) {
  __shared__ __attribute__((aligned(1024))) char smem[gemm_lds_bytes<BM>()];
  gemm_tile<BM>(args, blockIdx.x, smem);
}

// Precise mode: the same tile with the split-bf16 K map and the hi/lo output split (separate kernels, so that the bf16 fast
// path above keeps its exact instruction stream).
template <int BM>
__global__ __launch_bounds__(NTHREADS) void lx_gemm_split_kernel(const GemmArgs args) {
  __shared__ __attribute__((aligned(1024))) char smem[gemm_lds_bytes<BM>()];
  gemm_tile<BM, true>(args, blockIdx.x, smem);
}

// ---- fp8 (OCP e4m3) GEMM: BASELINE configs[4] ("fp8 MFMA ... path") ---------------------------------------------------------
// A [M, K] and W [N, K] are e4m3 BYTES (values pre-multiplied by an activation scale / per-row weight scales); the products run on
// v_mfma_f32_32x32x64_f8f6f4 (64-deep, twice the bf16 rate), fp32 accumulate; the epilogue multiplies column n by col_scale[n]
// (= 1 / (activation scale x weight scale of row n)) before bias / GELU / gate / residual, and can emit e4m3 again for the next
// GEMM (LX_EPI_STORE_FP8 x out_scale). A K tile is 128 elements = 128 B per row: the LDS image, the XOR swizzle, the
// buffer-addressed LDS-DMA staging, the rings, the barrier / counted-vmcnt protocol and the role split are byte for byte those of the
// bf16 loop; what changes is the fragment shape (the lane's 32 bytes = 16-B slots 4 ks + 2 g, + 1 of its row: two ds_read_b128) and
// the step count (two 64-deep k steps per tile instead of four 16-deep ones). Operand convention: lane (row = lane % 32, g =
// lane / 32) supplies 32 bytes, byte p of group g is k = 32 g + p on both operands (tools/ubench/fp8_mfma).
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int BM, class F>
__device__ __forceinline__ void gemm_mainloop_fp8(const lx_gemm_desc& P, const int m0, const int n0, const int tn, const int kt0, const int kt1,
                                                  char* smem, f32x16 (&acc)[2][BM / 64], const int tid, F&& after_issue) {
  constexpr int MI = BM / 64;
  constexpr int KB = 128;                   // bytes (= elements) per row of a K tile
  constexpr int A_BYTES = BM * KB;
  constexpr int W_BYTES = BN * KB;
  constexpr int NSA = BM == 128 ? 3 : 2;
  constexpr int NSW = BM == 128 ? 3 : 2;
  constexpr int W_BASE = NSA * A_BYTES;
  constexpr int WAIT_STEADY = BM == 128 ? MI + 4 : 0;
  static_assert(W_BASE + NSW * W_BYTES == gemm_lds_bytes<BM>(), "fp8 rings must have the bf16 rings' geometry");
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int M = P.M, N = P.N, K = P.K;
  const bool w_tiled = (P.epilogue & LX_W_TILED) != 0;
  uint32_t aoff[MI], woff[4];
  {
    const int rsub = lane >> 3, pslot = lane & 7;
#pragma unroll
    for (int j = 0; j < MI; ++j) {
      const int row = (j * 8 + wave) * 8 + rsub;
      const int lslot = pslot ^ ((row >> 1) & 7);
      aoff[j] = (uint32_t)((min(m0 + row, M - 1) - m0) * P.lda + lslot * 16);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (w_tiled) {
        woff[j] = (uint32_t)((j * 8 + wave) * 1024 + lane * 16);
      } else {
        const int row = (j * 8 + wave) * 8 + rsub;
        const int lslot = pslot ^ ((row >> 1) & 7);
        woff[j] = (uint32_t)((min(n0 + row, N - 1) - n0) * P.ldw + lslot * 16);
      }
    }
  }
  const uint8_t* a_org = (const uint8_t*)P.A + (size_t)m0 * P.lda;
  const uint8_t* w_org = w_tiled ? (const uint8_t*)P.W + ((size_t)tn * (K / KB)) * (BN * KB) : (const uint8_t*)P.W + (size_t)n0 * P.ldw;
  const lx_rsrc_t rs_a = lx_make_rsrc(a_org), rs_w = lx_make_rsrc(w_org);
  const int w_kstride_b = w_tiled ? BN * KB : KB;
  auto stage_a = [&](int kt, int slot) {
    char* base = smem + slot * A_BYTES;
#pragma unroll
    for (int j = 0; j < MI; ++j) lx_buf_to_lds(rs_a, (lptr_t)(base + (j * 8 + wave) * 1024), aoff[j], (kt0 + kt) * KB);
  };
  auto stage_w = [&](int kt, int slot) {
    char* base = smem + W_BASE + slot * W_BYTES;
#pragma unroll
    for (int j = 0; j < 4; ++j) lx_buf_to_lds(rs_w, (lptr_t)(base + (j * 8 + wave) * 1024), woff[j], (kt0 + kt) * w_kstride_b);
  };
  const int sw = (l31 >> 1) & 7;
  int slot_off[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int h = 0; h < 2; ++h) slot_off[ks][h] = ((ks * 4 + lhi * 2 + h) ^ sw) * 16;
  const int a_row_off = (wm * (BM / 2) + l31) * 128;
  const int w_row_off = (wn * 64 + l31) * 128;
  auto frag = [&](const char* p, int ks) {
    const u32x4 lo = *(const u32x4*)(p + slot_off[ks][0]), hi = *(const u32x4*)(p + slot_off[ks][1]);
    return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
  };
  auto load_frags = [&](int sa, int sw_, int ks, i32x8 (&wf)[2], i32x8 (&xf)[MI]) {
    const char* pa = smem + sa * A_BYTES + a_row_off;
    const char* pw = smem + W_BASE + sw_ * W_BYTES + w_row_off;
#pragma unroll
    for (int j = 0; j < 2; ++j) wf[j] = frag(pw + j * 32 * 128, ks);
#pragma unroll
    for (int i = 0; i < MI; ++i) xf[i] = frag(pa + i * 32 * 128, ks);
  };
  auto mma_j = [&](int j, const i32x8 (&wf)[2], const i32x8 (&xf)[MI]) {
#pragma unroll
    for (int i = 0; i < MI; ++i) acc[j][i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[j], xf[i], acc[j][i], 0, 0, 0, 0, 0, 0);
  };
  const int nkt = kt1 - kt0;
  i32x8 wfA[2], xfA[MI], wfB[2], xfB[MI];
  {
    stage_a(0, 0);
    stage_w(0, 0);
    if (nkt > 1) { stage_a(1, 1); stage_w(1, 1); }
    if (nkt > 2) { if constexpr (NSA > 2) stage_a(2, 2); if constexpr (NSW > 2) stage_w(2, 2); }
    after_issue();
    if (nkt > 2 && NSW > 2) {
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  load_frags(0, 0, 0, wfA, xfA);
#define LX_STEP8(CUR_W, CUR_X, NEXT_STMT)           \
  __builtin_amdgcn_s_setprio(1);                    \
  mma_j(0, CUR_W, CUR_X);                           \
  __builtin_amdgcn_s_setprio(0);                    \
  __builtin_amdgcn_sched_barrier(0);                \
  NEXT_STMT;                                        \
  __builtin_amdgcn_sched_barrier(0);                \
  __builtin_amdgcn_s_setprio(1);                    \
  mma_j(1, CUR_W, CUR_X);                           \
  __builtin_amdgcn_s_setprio(0);                    \
  __builtin_amdgcn_sched_barrier(0);
  int ca = 0, cw = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    const int na = ca + 1 == NSA ? 0 : ca + 1;
    const int nw = cw + 1 == NSW ? 0 : cw + 1;
    LX_STEP8(wfA, xfA, load_frags(ca, cw, 1, wfB, xfB))
    if (kt + 2 < nkt && WAIT_STEADY > 0) {
      asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (wm == 0) {
      if (kt + NSA < nkt) stage_a(kt + NSA, ca);
      if (kt + NSW < nkt) stage_w(kt + NSW, cw);
    }
    __builtin_amdgcn_sched_barrier(0);
    LX_STEP8(wfB, xfB, if (kt + 1 < nkt) load_frags(na, nw, 0, wfA, xfA))
    if (wm == 1) {
      if (kt + NSA < nkt) stage_a(kt + NSA, ca);
      if (kt + NSW < nkt) stage_w(kt + NSW, cw);
    }
    __builtin_amdgcn_sched_barrier(0);
    ca = na;
    cw = nw;
  }
#undef LX_STEP8
}

template <int BM>
__global__ __launch_bounds__(NTHREADS) void lx_gemm_fp8_kernel(const GemmArgs args) {
  __shared__ __attribute__((aligned(1024))) char smem[gemm_lds_bytes<BM>()];
  constexpr int MI = BM / 64;
  const int pid = blockIdx.x;
  const int total = args.tile_start[MAX_SUB];
  int lid;
  {
    const int q = total >> 3, r = total & 7;
    const int xcd = pid & 7, inx = pid >> 3;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + inx;
  }
  const int g = tile_group(args, lid);
  const lx_gemm_desc P = args.p[g];
  int tm, tn;
  tile_coords<BM>(P, lid - args.tile_start[g], tm, tn);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  f32x16 acc[2][MI];
  acc_clear<MI>(acc);
  const int wm = wave >> 2, wn = wave & 3, l31 = lane & 31, lhi = lane >> 5;
  const int m0 = tm * BM, n0 = tn * BN;
  const bool lora_early = lora_in_prologue(P);
  f32x4 u4[2], sv[MI][4];
  if (lora_early) lora_issue<MI>(P, n0, m0 + wm * (BM / 2), n0 + wn * 64, l31, 0, 0, u4, sv);
  // The LoRA term is NOT in the accumulator's units (acc * col_scale): pre-divide the up rows by col_scale so that one scale fits all
  gemm_mainloop_fp8<BM>(P, m0, n0, tn, 0, P.K / 128, smem, acc, tid, [&]() {
    if (lora_early) {
      f32x4 t4[MI];
      lora_sum<MI>(P, 0, sv, t4);
      if (P.col_scale) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float inv = 1.0f / P.col_scale[min(n0 + wn * 64 + j * 32 + l31, P.N - 1)];
#pragma unroll
          for (int e = 0; e < 4; ++e) u4[j][e] *= inv;
        }
      }
      lora_apply<MI>(u4, t4, lhi, acc);
    }
  });
  gemm_epilogue<BM, MI, false, true>(P, acc, smem, m0, n0, args.m_base[g], wave, wm, wn, lane, l31, lhi, lora_early);
}

// Mixed-height launch: `big` holds full rounds of 256-row tiles, `tail` the remaining rows as 128-row tiles, in ONE grid
// [big tiles | padding to a multiple of 8 | tail tiles]. Launched one after the other, the tail (e.g. 168 tiles on 256 CUs) only
// starts when the last big round has drained everywhere; in one grid a CU that finishes its last big tile picks up a tail tile
// at once. The padding keeps blockIdx % 8 (the XCD a workgroup lands on) equal to pid % 8 for the tail's tile map.
__global__ __launch_bounds__(NTHREADS) void lx_gemm_mixed_kernel(const GemmArgs big, const GemmArgs tail, const int n_big_pad) {
  __shared__ __attribute__((aligned(1024))) char smem[gemm_lds_bytes<128>() > gemm_lds_bytes<256>() ? gemm_lds_bytes<128>() : gemm_lds_bytes<256>()];
  const int bid = blockIdx.x;
  if (bid < big.tile_start[MAX_SUB]) gemm_tile<256>(big, bid, smem);
  else if (bid >= n_big_pad) gemm_tile<128>(tail, bid - n_big_pad, smem);
}

// ---- 256-row tiles for launches with too few of them: two workgroups per tile, each half of K ------------------------------
// The N = 3072 projections of the DiT have 120 tiles of 256 x 256 at M = 2560: one per CU would leave half the chip idle, so
// they ran as 240 tiles of 128 x 256 -- whose main loop moves 1.5x the LDS-DMA bytes per flop and measures 13 % slower per flop
// (0.867 vs 1.50 / 2 us per K tile). Here the 256 x 256 tile is kept and two workgroups (blockIdx p and p ^ 8: same XCD, same L2)
// take K ranges [0, nkt/2) and [nkt/2, nkt) of it. Afterwards they swap halves instead of one of them collecting everything:
// in workgroup h every wave keeps the accumulators of its 32-row blocks 2h and 2h+1 and sends the other two blocks (128 KiB per
// workgroup, lane-linear fp32, straight from registers, agent-scope write-through) to the partner's slot; each then adds what
// it received and runs the fused epilogue on its two blocks per wave -- all eight waves busy, the epilogue as short as the
// 128-row kernel's. One fp32 addition per element, commutative, so both halves of the tile round the same way and the result
// does not depend on timing.
//   Hand-off: stores, s_waitcnt vmcnt(0), barrier, flag[p] = 1; then wait for flag[p ^ 8] and clear it (each flag has one writer
// and one reader, and ends the launch at 0: hipGraph replays need no reset). The data path assumes nothing about placement (sc1
// stores AND sc1 loads: correct across XCDs; p ^ 8 is only the likely-same-L2 choice). Both partners wait for each other, so
// both must get a CU: a launch has at most 256 workgroups of one per CU on a 256-CU device (checked on the host), so all of them
// are resident together unless something else holds CUs; the poll is bounded and raises the workspace's error word (the host
// reads it with lx_gemm_workspace_status) instead of hanging or trapping if a partner does not show up in time.
constexpr int PAIR_SLOT_FLOATS = 8 * 16 * 64 * 4;   // eight waves x 16 x f32x4 per lane = 128 KiB
constexpr int PAIR_AUX_SC1 = 16;                    // gfx940+ buffer cache policy: sc1 (agent scope)
constexpr int PAIR_MAX_WG = 256;

__global__ __launch_bounds__(NTHREADS) void lx_gemm_pair_kernel(const GemmArgs args, float* __restrict__ slots, int* __restrict__ flags) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 256, MI = 4;
  __shared__ __attribute__((aligned(1024))) char smem[gemm_lds_bytes<256>()];
  const int pid = blockIdx.x, xcd = pid & 7, inx = pid >> 3;
  const int jx = inx >> 1, half = inx & 1;
  const int total = args.tile_start[MAX_SUB];     // plan_add keeps every entry past the last problem equal to the total
  const int q = total >> 3, r = total & 7;
  if (jx >= q + (xcd < r ? 1 : 0)) return;                  // (both partners of a tile that does not exist leave together)
  const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + jx;
  const int g = tile_group(args, lid);
  const lx_gemm_desc P = args.p[g];            // by value: one batch of scalar loads
  int tm, tn;
  tile_coords<BM>(P, lid - args.tile_start[g], tm, tn);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3, l31 = lane & 31, lhi = lane >> 5;
  const int m0 = tm * BM, n0 = tn * BN;
  const int nkt = P.K / BK, kmid = nkt >> 1;
  f32x16 acc[2][MI];
  acc_clear<MI>(acc);
  const bool lora_early = lora_in_prologue(P);              // the LoRA term enters once: through workgroup 0 of the pair
  f32x4 u4[2], sv[MI][4];
  if (lora_early && half == 0) lora_issue<MI>(P, n0, m0 + wm * (BM / 2), n0 + wn * 64, l31, 0, 0, u4, sv);
  gemm_mainloop<BM, false>(P, m0, n0, tn, half ? kmid : 0, half ? nkt : kmid, smem, acc, tid, [&]() {
    if (lora_early && half == 0) {
      f32x4 t4[MI];
      lora_sum<MI>(P, 0, sv, t4);
      lora_apply<MI>(u4, t4, lhi, acc);
    }
  });
  // ---- swap halves with the partner: every wave keeps two of its four 32-row blocks and sends the other two ----
  const int partner = pid ^ 8;
  const uint32_t lane_off = (uint32_t)(wave * 16 * 1024 + lane * 16);     // slot: [wave][j][kept block][rq][lane] x f32x4
  auto send = [&](auto I0) {
    constexpr int i0 = decltype(I0)::value;
    const lx_rsrc_t rs = lx_make_rsrc(slots + (size_t)pid * PAIR_SLOT_FLOATS);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const f32x16& a = acc[j][i0 + ii];
          u32x4 v = {__float_as_uint(a[rq * 4]), __float_as_uint(a[rq * 4 + 1]), __float_as_uint(a[rq * 4 + 2]), __float_as_uint(a[rq * 4 + 3])};
          __builtin_amdgcn_raw_buffer_store_b128(v, rs, lane_off, ((j * 2 + ii) * 4 + rq) * 1024, PAIR_AUX_SC1);
        }
  };
  auto recv = [&](auto I0) {
    constexpr int i0 = decltype(I0)::value;
    const lx_rsrc_t rs = lx_make_rsrc(slots + (size_t)partner * PAIR_SLOT_FLOATS);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      u32x4 v[2][4];
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) v[ii][rq] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off, ((j * 2 + ii) * 4 + rq) * 1024, PAIR_AUX_SC1);
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq)
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[j][i0 + ii][rq * 4 + c] += __uint_as_float(v[ii][rq][c]);
      __builtin_amdgcn_sched_barrier(0);       // 8 loads in flight, not 16: the accumulators leave few free registers
    }
  };
  // workgroup `half` keeps blocks 2*half, 2*half + 1 (compile-time register indices on both sides of the branch)
  if (half == 0) send(std::integral_constant<int, 2>{});
  else send(std::integral_constant<int, 0>{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __hip_atomic_store(flags + pid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // bounded: ~1-2 s of polling (each iteration is one L2 round trip), against microseconds of expected wait. A partner that never
    // shows up (a device whose CUs are held by something else for good) sets the workspace's error word and lets this workgroup
    // finish with an invalid tile: the host sees it in lx_gemm_workspace_status, the context survives, nothing hangs.
    int spins = 0;
    bool ok = true;
    while (__hip_atomic_load(flags + partner, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 20)) { ok = false; break; }
    }
    if (ok) __hip_atomic_store(flags + partner, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(flags + PAIR_MAX_WG, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // error word: lx_gemm_workspace_status
  }
  __syncthreads();
  if (half == 0) recv(std::integral_constant<int, 0>{});
  else recv(std::integral_constant<int, 2>{});
  gemm_epilogue<BM, MI>(P, acc, smem, m0, n0, args.m_base[g], wave, wm, wn, lane, l31, lhi, lora_early, half * 2, half * 2 + 2);
#endif
}


// ---------------------------------------------------------------------------------------------------------------------
// lx_gemm4_kernel: the same 256 x 256 x 64 tile with ONE wave per SIMD -- 256 threads, 2 x 2 waves of 128 x 128, the 256 accumulator
// registers of a wave in AGPRs, v_mfma_f32_16x16x32_bf16 (a weight fragment is held for eight MFMAs), ALL fragments of a K tile in
// registers (2 k-steps x (8 + 8) x 4 VGPRs), one ds_read_b128 or one LDS-DMA piece per MFMA gap, two barriers per K tile: the reads
// of a stage are finished (registers) before its refill is issued, so two stages of 64 KiB suffice. This is the shape of the vendor
// library's fastest kernel on this part; measured against the 8-wave loop above in tools/ubench/loop4w_rate:
// 1.39 vs 1.58 us per K tile (the fragment bytes read from LDS per MFMA cycle are 2/3, the wave count per SIMD half). Same LDS image,
// swizzle, pre-tiled weights and tile map as gemm_tile. Epilogues so far: bias / GELU / bf16 or fp32 store / gated residual (no LoRA,
// no LX_EPI_QKV: those launches stay on the kernels above).
constexpr int G4_THREADS = 256;
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
constexpr int SK_SLOT_FLOATS = 256 * 256;

// SPLIT = the split-bf16 ("precise") problems of lx_gemm_split_kernel: k_segs passes over K in one accumulation (A_hi W_hi, A_lo W_hi,
// A_hi W_lo: the K-tile index of the loop maps to (segment, tile) -> source offsets, once per K tile on the scalar unit) and the
// hi / lo output pair of LX_EPI_SPLIT_BF16. A separate instantiation, so that the bf16 path keeps its exact instruction stream.
template <bool SPLIT>
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
    lid = sk_full + (r >> 1);
    part = r & 1;
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
  const int kt_begin = split_tile && part ? nkt >> 1 : 0;
  const int kt_end = split_tile && !part ? nkt >> 1 : nkt;

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
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(wf[ks][j]), "v"(af[ks][i]));
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
  float* my_slot = nullptr;
  if (split_tile) {
    my_slot = sk_slots + (size_t)(pid - sk_full) * SK_SLOT_FLOATS;
    partner = sk_slots + (size_t)((pid - sk_full) ^ 1) * SK_SLOT_FLOATS;    // read with sc1 loads (written with sc1 stores): no cache maintenance
  }
  auto pld4 = [&](size_t off_floats) {                 // 16 B of the partner's slot, agent scope (sc1: not from this CU's L1 / a stale L2 line)
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(lx_make_rsrc(partner), (int)(off_floats * 4), 0, PAIR_AUX_SC1);
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
      const int me = pid - sk_full, other = me ^ 1;
      if (!(sk_parts == 3 && part == 1))               // (3: fault injection, part 1 never raises its flag -- tools/race_screen_g4.py)
        __hip_atomic_store(sk_flags + me, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      bool ok = true;
      while (__hip_atomic_load(sk_flags + other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 20)) { ok = false; break; }
      }
      if (ok) __hip_atomic_store(sk_flags + other, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (each flag: raised by its owner, reset by its reader)
      else __hip_atomic_store(sk_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // lx_gemm_workspace_status reports it
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
      f32x4 pa[4][2];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const size_t po = (size_t)(lm0 + i * 16 + t * 4 + (lane >> 4)) * 256 + ln0 + c8;
        pa[t][0] = pld4(po); pa[t][1] = pld4(po + 4);
      }
      float* pw = patch + (i & 1) * (16 * G4_PLD);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float* q_ = pw + (t * 4 + (lane >> 4)) * G4_PLD + c8;
        f32x4 v0 = *(const f32x4*)q_, v1 = *(const f32x4*)(q_ + 4);
#pragma unroll
        for (int c_ = 0; c_ < 4; ++c_) { v0[c_] += pa[t][0][c_]; v1[c_] += pa[t][1][c_]; }
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
        if (P.qkv_q8) {
          // e4m3 V^T image (gemm_epilogue_qkv): in the f8f6f4 operand order a 64-key tile row is two 32-byte groups g, byte p of group g =
          // key (p >> 4) * 32 + 8 * ((p & 15) >> 2) + 4 g + (p & 3): this block's 16 keys are 8 bytes of each group, the same two key
          // sets {0-3, 8-11} / {4-7, 12-15} as the bf16 image's interleave -- lane (d, g = gk) stores its 8 keys as 8 bytes
          const float vs = P.qkv_v_scale;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int d = t * 32 + (lane >> 1);
            const float bd = P.bias ? P.bias[nw0 + d] : 0.f;
            u32x2 o8 = {pack_fp8x4((e[t][0] + bd) * vs, (e[t][1] + bd) * vs, (e[t][2] + bd) * vs, (e[t][3] + bd) * vs),
                        pack_fp8x4((e[t][4] + bd) * vs, (e[t][5] + bd) * vs, (e[t][6] + bd) * vs, (e[t][7] + bd) * vs)};
            *(u32x2*)((uint8_t*)P.qkv_vt8 + ((size_t)(b * H + h) * 128 + d) * P.qkv_vt_ld + P.qkv_vt_pos0 + (p0 & ~63) + gk * 32 + ((p0 >> 5) & 1) * 16 +
                      ((p0 >> 4) & 1) * 8) = o8;
          }
          return;
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
          ss += __shfl_xor(ss, 1, 64); ss += __shfl_xor(ss, 2, 64); ss += __shfl_xor(ss, 4, 64); ss += __shfl_xor(ss, 8, 64);
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
        // the four row stores back to back, the output kind decided ONCE per block (a tile-uniform branch inside the row loop costs a
        // wait at every join: the parked-store branch of the fp32 path cost 4 us per round)
        if (P.qkv_q8) {                                  // e4m3 rows (x the tensor's scale) for lx_attn_fwd_fp8 instead of the bf16 outputs
          const float sc8 = qkind == 0 ? P.qkv_k_scale : P.qkv_q_scale;
          uint8_t* const o8p = (qkind == 0 ? (uint8_t*)P.qkv_k8 : (uint8_t*)P.qkv_q8) + (ncol - qkind * D);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int m = mb + t * 4 + (lane >> 4);
            const float (&y)[8] = yy[t];
            u32x2 o8 = {pack_fp8x4(y[0] * sc8, y[1] * sc8, y[2] * sc8, y[3] * sc8), pack_fp8x4(y[4] * sc8, y[5] * sc8, y[6] * sc8, y[7] * sc8)};
            if (m < M) *(u32x2*)(o8p + (size_t)m * P.qkv_ld8) = o8;
          }
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int m = mb + t * 4 + (lane >> 4);
            const float (&y)[8] = yy[t];
            u32x4 o = {pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]), pack_bf16x2(y[4], y[5]), pack_bf16x2(y[6], y[7])};
            if (m < M) *(u32x4*)(out + (size_t)m * out_ld + ncol) = o;
          }
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
          u32x4 o = {pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v0[2], v0[3]), pack_bf16x2(v1[0], v1[1]), pack_bf16x2(v1[2], v1[3])};
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
  } else if (part == 0) {                              // parks 4-7, then owns 0-3
    if (mw0 + 64 < M) put(std::integral_constant<int, 4>{});
    G4_K(4, 5); G4_K(5, 6); G4_K(6, 7); G4_K(7, 0);
    exchange();
    G4_B(0, 1); G4_B(1, 2); G4_B(2, 3); G4_B(3, -1);
  } else {                                             // parks 0-3, then owns 4-7
    if (mw0 < M) put(std::integral_constant<int, 0>{});
    G4_K(0, 1); G4_K(1, 2); G4_K(2, 3); G4_K(3, 4);
    exchange();
    G4_B(4, 5); G4_B(5, 6); G4_B(6, 7); G4_B(7, -1);
  }
#undef G4_B
#undef G4_K
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  G4_STAMP(5)
#endif
}

}  // namespace

// ---- launch planning ---------------------------------------------------------------------------------------------
// One workgroup per CU, so a launch runs in "rounds" of 256 tiles. Per-round cost model calibrated on MI355X with
// rocprofv3 kernel durations over a K sweep (tools/gemm_ksweep.py): fixed (launch + prologue + epilogue burst) plus a
// per-64-deep-K-tile slope at full occupancy.
static double round_us(int bm, int K) { return bm == 256 ? 15.0 + 1.81 * (K / 64) : 10.5 + 1.06 * (K / 64); }

// runtime switches, read once per process (LX_GEMM_BM, LX_GEMM_PAIR, LX_GEMM_PAIR_MIN_KT, LX_GEMM_MIXED_ONE_GRID)
struct GemmEnv { int bm, pair, pair_min_kt, one_grid, g4, sk, g4_fault, g4_q8, sk_tail_div, sk_max_rounds; };
static GemmEnv read_gemm_env() {
  return GemmEnv{env_int("LX_GEMM_BM", 0), env_int("LX_GEMM_PAIR", 1), env_int("LX_GEMM_PAIR_MIN_KT", 96), env_int("LX_GEMM_MIXED_ONE_GRID", 1),
                 env_int("LX_GEMM4", 1), env_int("LX_GEMM4_SK", 1), env_int("LX_GEMM4_FAULT", 0), env_int("LX_GEMM4_Q8", 0), env_int("LX_GEMM4_TAIL_DIV", 0), env_int("LX_GEMM4_SK_ROUNDS", 16)};
}
static GemmEnv g_gemm_env = read_gemm_env();
static const GemmEnv& gemm_env() { return g_gemm_env; }
extern "C" void lx_gemm_reload_env(void) { g_gemm_env = read_gemm_env(); }

// ---- pair-kernel scratch: one accumulator-exchange slot + flag per workgroup, in a CALLER-PROVIDED workspace ------------------
// (lx_gemm_bf16_ws): [PAIR_MAX_WG slots of 128 KiB | PAIR_MAX_WG flags | error word]. The library keeps no scratch of its own:
// whoever owns a stream owns its workspace, so launches on different streams never share slots or flags.
namespace {
constexpr size_t PAIR_WS_BYTES = (size_t)PAIR_MAX_WG * PAIR_SLOT_FLOATS * sizeof(float) + (PAIR_MAX_WG + 64) * sizeof(int);
// IN FRONT of it, the split-tile area of lx_gemm4_kernel: [256 slots of 256 KiB | 256 flags + pad]; the pair area stays LAST, so the one
// error word both kernels raise is still the int at (end - 64 ints): callers that poll it asynchronously read that position
constexpr size_t SK_AREA_BYTES = (((size_t)256 * SK_SLOT_FLOATS * sizeof(float) + (256 + 64) * sizeof(int)) + 255) & ~(size_t)255;
constexpr size_t PAIR_OFF = SK_AREA_BYTES;               // byte offset of the pair area
constexpr size_t SK_WS_BYTES = PAIR_OFF + PAIR_WS_BYTES;
static_assert(PAIR_WS_BYTES % 4 == 0, "the error word is an aligned int");

int device_cus() {
  static int n = -1;
  if (n < 0) {
    int dev = 0;
    hipDeviceProp_t pr;
    n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 0;
  }
  return n;
}
}  // namespace

static long tiles_of(const lx_gemm_desc& p, int bm) { return (long)((p.M + bm - 1) / bm) * ((p.N + BN - 1) / BN); }

static lx_gemm_desc sub_rows(const lx_gemm_desc& p, int r0, int rows) {   // rows [r0, r0+rows) of a problem
  lx_gemm_desc q = p;
  q.A = (const uint16_t*)p.A + (size_t)r0 * p.lda;
  const int epi = p.epilogue & 0xff;
  q.C = epi == LX_EPI_STORE_BF16 ? (void*)((uint16_t*)p.C + (size_t)r0 * p.ldc) : (void*)((float*)p.C + (size_t)r0 * p.ldc);
  if (p.lora_t) q.lora_t = p.lora_t + (size_t)r0 * p.lora_ldt;
  if (p.qkv_k) q.qkv_k = (uint16_t*)p.qkv_k + (size_t)r0 * p.qkv_k_ld;
  if (p.qkv_q8) {
    q.qkv_q8 = (uint8_t*)p.qkv_q8 + (size_t)r0 * p.qkv_ld8;
    q.qkv_k8 = (uint8_t*)p.qkv_k8 + (size_t)r0 * p.qkv_ld8;
  }
  q.M = rows;
  return q;
}

static int launch_plan(const GemmArgs& a, int bm, hipStream_t s, bool split = false) {
  if (a.n == 0) return LX_OK;
  const int t = a.tile_start[a.n];
  if (split) {
    if (bm == 256) hipLaunchKernelGGL(lx_gemm_split_kernel<256>, dim3(t), dim3(NTHREADS), 0, s, a);
    else hipLaunchKernelGGL(lx_gemm_split_kernel<128>, dim3(t), dim3(NTHREADS), 0, s, a);
  } else if (bm == 256) hipLaunchKernelGGL(lx_gemm_kernel<256>, dim3(t), dim3(NTHREADS), 0, s, a);
  else hipLaunchKernelGGL(lx_gemm_kernel<128>, dim3(t), dim3(NTHREADS), 0, s, a);
  LX_LAUNCH_CHECK("lx_gemm_bf16");
  return LX_OK;
}

static void plan_add(GemmArgs& a, const lx_gemm_desc& p, int m_base, int bm) {
  a.p[a.n] = p;
  a.m_base[a.n] = m_base;
  a.tile_start[a.n + 1] = a.tile_start[a.n] + (int)tiles_of(p, bm);
  ++a.n;
  for (int i = a.n + 1; i <= MAX_SUB; ++i) a.tile_start[i] = a.tile_start[a.n];
}

extern "C" size_t lx_gemm_workspace_bytes(void) { return SK_WS_BYTES; }
#ifdef LX_G4_PROBE
extern "C" int lx_g4_probe_read(unsigned long long* host, size_t n_u64) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(lx_g4_probe_buf), n_u64 * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int lx_gemm_workspace_status(void* workspace, void* stream) {
  LX_CHECK_ARG(workspace, "lx_gemm_workspace_status: NULL workspace");
  int* err = (int*)((char*)workspace + PAIR_OFF + (size_t)PAIR_MAX_WG * PAIR_SLOT_FLOATS * sizeof(float)) + PAIR_MAX_WG;
  int v = 0;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemcpyAsync(&v, err, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
    lx_set_error("lx_gemm_workspace_status: reading the error word failed: %s", hipGetErrorString(hipGetLastError()));
    return LX_ERR_LAUNCH;
  }
  if (v == 0) return LX_OK;
  // a timed-out pair leaves flags behind: reset all of them with the error word so the workspace is usable again
  (void)hipMemsetAsync((char*)workspace + PAIR_OFF + (size_t)PAIR_MAX_WG * PAIR_SLOT_FLOATS * sizeof(float), 0, (PAIR_MAX_WG + 64) * sizeof(int), s);
  (void)hipMemsetAsync((char*)workspace + (size_t)256 * SK_SLOT_FLOATS * sizeof(float), 0, (256 + 64) * sizeof(int), s);
  (void)hipStreamSynchronize(s);
  lx_set_error("lx_gemm_bf16_ws: a split-K pair workgroup timed out waiting for its partner (CUs held by other work?); the results "
               "of that launch are invalid. Re-run with the workspace omitted (lx_gemm_bf16) or LX_GEMM_PAIR=0");
  return LX_ERR_LAUNCH;
}

extern "C" int lx_gemm_bf16(const lx_gemm_desc* problems, int n, void* stream) { return lx_gemm_bf16_ws(problems, n, nullptr, 0, stream); }

extern "C" int lx_gemm_bf16_ws(const lx_gemm_desc* problems, int n, void* workspace, size_t ws_bytes, void* stream) {
  LX_CHECK_ARG(problems && n >= 1 && n <= LX_GEMM_MAX_GROUP, "lx_gemm_bf16: n=%d out of range [1,%d]", n, LX_GEMM_MAX_GROUP);
  long t256 = 0, t128 = 0;
  int kmax = 0;
  bool split = false, fp8 = false, qkv = false;
  for (int i = 0; i < n; ++i) fp8 = fp8 || (problems[i].epilogue & LX_OPERANDS_FP8) != 0;
  for (int i = 0; i < n; ++i) {
    const lx_gemm_desc& p = problems[i];
    if (fp8) {
      LX_CHECK_ARG((p.epilogue & LX_OPERANDS_FP8) != 0 && p.k_segs <= 1 && !(p.epilogue & LX_EPI_SPLIT_BF16), "lx_gemm_bf16[%d]: LX_OPERANDS_FP8 must be set on every problem of a launch and excludes the split-bf16 mode", i);
      LX_CHECK_ARG(p.K % 128 == 0 && p.lda % 16 == 0 && p.ldw % 16 == 0, "lx_gemm_bf16[%d]: fp8 operands need K %% 128 == 0 and lda / ldw %% 16 == 0 (K=%d)", i, p.K);
      if (p.col_scale) LX_CHECK_ARG(((uintptr_t)p.col_scale & 15) == 0, "lx_gemm_bf16[%d]: col_scale must be 16-byte aligned", i);
      if (p.lora_t) LX_CHECK_ARG(p.lora_r <= 4 && p.lora_nsplit <= 4, "lx_gemm_bf16[%d]: the fp8 path applies LoRA in the tile prologue only (rank <= 4, <= 4 slabs)", i);
      if ((p.epilogue & 0xff) == LX_EPI_STORE_FP8) LX_CHECK_ARG(p.out_scale > 0.f && p.ldc % 8 == 0, "lx_gemm_bf16[%d]: LX_EPI_STORE_FP8 needs out_scale > 0 and ldc %% 8 == 0", i);
    }
    LX_CHECK_ARG(p.k_segs >= 0 && p.k_segs <= 3, "lx_gemm_bf16[%d]: k_segs=%d must be 0..3", i, p.k_segs);
    const int segs = p.k_segs > 1 ? p.k_segs : 1;
    if (segs > 1 || (p.epilogue & LX_EPI_SPLIT_BF16)) split = true;
    if (segs > 1) LX_CHECK_ARG(p.a_lo_off >= p.K && p.a_lo_off % 8 == 0 && p.lda >= p.a_lo_off + p.K, "lx_gemm_bf16[%d]: a_lo_off=%d needs K <= a_lo_off, a_lo_off + K <= lda, multiple of 8", i, p.a_lo_off);
    if (segs == 3) LX_CHECK_ARG(p.ldw >= 2 * p.K, "lx_gemm_bf16[%d]: k_segs = 3 reads W as [N, 2K] = [W_hi | W_lo]: ldw=%d < 2K", i, p.ldw);
    if (p.epilogue & LX_EPI_SPLIT_BF16) LX_CHECK_ARG((p.epilogue & 0xff) == LX_EPI_STORE_BF16 && p.c_lo_off >= p.N && p.c_lo_off % 8 == 0 && p.ldc >= p.c_lo_off + p.N,
                                                     "lx_gemm_bf16[%d]: LX_EPI_SPLIT_BF16 needs a bf16 store and N <= c_lo_off, c_lo_off + N <= ldc, multiple of 8", i);
    LX_CHECK_ARG(p.A && p.W && p.C, "lx_gemm_bf16[%d]: NULL operand", i);
    LX_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "lx_gemm_bf16[%d]: bad shape M=%d N=%d K=%d", i, p.M, p.N, p.K);
    LX_CHECK_ARG(p.K % BK == 0, "lx_gemm_bf16[%d]: K=%d must be a multiple of %d", i, p.K, BK);
    LX_CHECK_ARG(p.N % 8 == 0, "lx_gemm_bf16[%d]: N=%d must be a multiple of 8", i, p.N);
    LX_CHECK_ARG(p.lda % 8 == 0 && p.ldw % 8 == 0 && p.lda >= p.K && p.ldw >= p.K, "lx_gemm_bf16[%d]: lda/ldw must be >= K and multiples of 8", i);
    LX_CHECK_ARG(((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.W & 15) == 0 && ((uintptr_t)p.C & 15) == 0, "lx_gemm_bf16[%d]: operands must be 16-byte aligned", i);
    LX_CHECK_ARG(p.ldc % 8 == 0 && p.ldc >= p.N, "lx_gemm_bf16[%d]: ldc=%d must be >= N and a multiple of 8", i, p.ldc);
    const int epi = p.epilogue & 0xff;
    LX_CHECK_ARG(epi >= LX_EPI_STORE_BF16 && (epi <= LX_EPI_RESID_F32 || (fp8 && epi == LX_EPI_STORE_FP8)), "lx_gemm_bf16[%d]: unknown epilogue %d", i, p.epilogue);
    LX_CHECK_ARG(p.rows_per_batch > 0, "lx_gemm_bf16[%d]: rows_per_batch must be > 0", i);
    if (p.epilogue & LX_W_TILED) LX_CHECK_ARG(p.N % BN == 0 && p.ldw == (segs == 3 ? 2 * p.K : p.K), "lx_gemm_bf16[%d]: LX_W_TILED needs N %% 256 == 0 and ldw == K (2K with k_segs = 3)", i);
    if (p.gate) LX_CHECK_ARG(p.gate_ld >= p.N && p.gate_ld % 4 == 0, "lx_gemm_bf16[%d]: gate_ld=%d", i, p.gate_ld);
    if (p.lora_t) {
      LX_CHECK_ARG(p.lora_up && p.lora_r >= 1 && p.lora_r <= 16, "lx_gemm_bf16[%d]: LoRA needs lora_up and 1 <= r <= 16", i);
      LX_CHECK_ARG(p.lora_mod_cols <= 0 || p.lora_mod_cols % BN == 0, "lx_gemm_bf16[%d]: lora_mod_cols must be a multiple of %d", i, BN);
    }
    if (p.epilogue & LX_EPI_QKV) {
      LX_CHECK_ARG(!fp8 && segs == 1 && !(p.epilogue & LX_EPI_SPLIT_BF16) && epi == LX_EPI_STORE_BF16, "lx_gemm_bf16[%d]: LX_EPI_QKV goes with plain bf16 operands and LX_EPI_STORE_BF16", i);
      LX_CHECK_ARG(p.qkv_d > 0 && p.qkv_d % BN == 0 && (p.N >= 3 * p.qkv_d || p.N % BN == 0), "lx_gemm_bf16[%d]: LX_EPI_QKV needs qkv_d %% 256 == 0 and whole projection tiles (qkv_d=%d N=%d)", i, p.qkv_d, p.N);
      LX_CHECK_ARG(p.rows_per_batch % 32 == 0 && p.M % 32 == 0, "lx_gemm_bf16[%d]: LX_EPI_QKV needs rows_per_batch %% 32 == 0 and M %% 32 == 0 (%d, %d)", i, p.rows_per_batch, p.M);
      const bool q8 = p.qkv_q8 != nullptr;
      LX_CHECK_ARG(p.qkv_norm_q && p.qkv_norm_k && (p.qkv_vt || q8) && p.qkv_rope && ((uintptr_t)p.qkv_norm_q & 15) == 0 && ((uintptr_t)p.qkv_norm_k & 15) == 0 && ((uintptr_t)p.qkv_vt & 15) == 0 &&
                   ((uintptr_t)p.qkv_rope & 15) == 0, "lx_gemm_bf16[%d]: LX_EPI_QKV needs 16-byte aligned qkv_norm_q / qkv_norm_k / qkv_rope / qkv_vt", i);
      if (q8) {
        LX_CHECK_ARG(p.qkv_k8 && p.qkv_vt8 && (((uintptr_t)p.qkv_q8 | (uintptr_t)p.qkv_k8 | (uintptr_t)p.qkv_vt8) & 15) == 0 && p.qkv_ld8 % 16 == 0 && p.qkv_ld8 >= p.qkv_d,
                     "lx_gemm_bf16[%d]: LX_EPI_QKV e4m3 outputs need 16-byte aligned qkv_q8 / qkv_k8 / qkv_vt8 and qkv_ld8 %% 16 == 0, >= qkv_d", i);
        LX_CHECK_ARG(p.qkv_q_scale > 0.f && p.qkv_k_scale > 0.f && p.qkv_v_scale > 0.f && !p.qkv_k, "lx_gemm_bf16[%d]: LX_EPI_QKV e4m3 outputs need positive scales and no separate bf16 key image", i);
      }
      LX_CHECK_ARG(p.qkv_vt_ld > 0 && p.qkv_vt_ld % 64 == 0 && p.qkv_vt_pos0 >= 0 && p.qkv_vt_pos0 % 64 == 0 && p.qkv_vt_pos0 + p.rows_per_batch <= p.qkv_vt_ld,
                   "lx_gemm_bf16[%d]: qkv_vt_ld / qkv_vt_pos0 must be multiples of 64 with the stream inside a V^T row", i);
      if (p.qkv_k) LX_CHECK_ARG(((uintptr_t)p.qkv_k & 15) == 0 && p.qkv_k_ld % 8 == 0 && p.qkv_k_ld >= p.qkv_d, "lx_gemm_bf16[%d]: qkv_k must be 16-byte aligned with qkv_k_ld %% 8 == 0, >= qkv_d", i);
      qkv = true;
    }
    if (p.bias) LX_CHECK_ARG(((uintptr_t)p.bias & 15) == 0, "lx_gemm_bf16[%d]: bias must be 16-byte aligned", i);
    t256 += tiles_of(p, 256);
    t128 += tiles_of(p, 128);
    kmax = p.K * segs > kmax ? p.K * segs : kmax;
  }
  hipStream_t s = (hipStream_t)stream;
  if (fp8) {     // e4m3 operands: all 256-row tiles or all 128-row tiles
    const int ncu = device_cus() > 0 ? device_cus() : 256;
    const double ca = (double)((t256 + ncu - 1) / ncu) * round_us(256, kmax / 4), cb = (double)((t128 + ncu - 1) / ncu) * round_us(128, kmax / 4);
    const int bm = gemm_env().bm ? gemm_env().bm : (cb < ca ? 128 : 256);
    GemmArgs all;
    all.n = 0;
    all.tile_start[0] = 0;
    for (int i = 1; i <= MAX_SUB; ++i) all.tile_start[i] = 0;
    for (int i = 0; i < n; ++i) plan_add(all, problems[i], 0, bm);
    const int t = all.tile_start[all.n];
    if (bm == 256) hipLaunchKernelGGL(lx_gemm_fp8_kernel<256>, dim3(t), dim3(NTHREADS), 0, s, all);
    else hipLaunchKernelGGL(lx_gemm_fp8_kernel<128>, dim3(t), dim3(NTHREADS), 0, s, all);
    LX_LAUNCH_CHECK("lx_gemm_bf16 (fp8)");
    return LX_OK;
  }
  const int NCU = device_cus() > 0 ? device_cus() : 256;   // one workgroup per CU: a launch runs in rounds of NCU tiles
  const GemmEnv& env = gemm_env();
  const int forced = env.bm;      // 256 | 128 | 0 = plan
  // lx_gemm4_kernel (one wave per SIMD, 256-row tiles only): the launches whose epilogue it has and whose tile count fills whole rounds.
  // LX_GEMM4 = 0 never | 1 (default) where the last round is at least 3/4 full or there are >= 8 rounds | 2 whenever the epilogue allows (tests).
  if (env.g4 && forced == 0 && (workspace || env.g4 == 2)) {     // (no workspace = the batch-size-invariant plans only)
    bool ok = true;
    for (int i = 0; i < n; ++i) {
      const lx_gemm_desc& p = problems[i];
      ok = ok && (p.epilogue & 0xff) <= LX_EPI_RESID_F32 && p.K / BK >= 2 && (env.g4_q8 || !((p.epilogue & LX_EPI_QKV) && p.qkv_q8));   // (e4m3 q/k/v images: built and tested here, measured 0.2 % slower per image at 1024 x 1024 than the 8-wave mixed plan -- 8-byte V^T stores per lane against 16 -- profiles/r04f_attnfp8_ab.txt; LX_GEMM4_Q8=1 turns it on)
      if (p.lora_t)        // the kernel's LoRA step: two ranks per 8-byte load, one 32-deep MFMA k-step, up to four K-split slabs
        ok = ok && p.lora_r <= 8 && p.lora_r % 2 == 0 && p.lora_nsplit >= 1 && p.lora_nsplit <= 4 && p.lora_ldt % 2 == 0 && p.lora_split_stride % 2 == 0 &&
             ((((uintptr_t)p.lora_t) | ((uintptr_t)p.lora_up)) & 7) == 0;
    }
    const long rounds = (t256 + NCU - 1) / NCU;
    bool fills = env.g4 == 2 || (t256 >= NCU && (rounds * NCU - t256 <= NCU / 4 || rounds >= 8));
    // split form (LX_GEMM4_SK = 1 default | 0 off): the tiles of a partial last round, or all tiles of a launch with <= 128 of them and
    // a long K, by two workgroups each (half of K), meeting through the caller's workspace. One K for the whole launch, >= 16 K tiles.
    // (launches of up to LX_GEMM4_SK_ROUNDS = 16 rounds: 7 until round 4 -- at batch 16 the N = 3072 projections are 1920 tiles = 7.5 rounds,
    //  whole tiles paid the half-empty eighth round: configs[2] 1.1318 / 1.1288 -> 1.1419 / 1.1418 images/s, profiles/r04aa_*)
    bool uniform_k4 = true;
    for (int i = 1; i < n; ++i) uniform_k4 = uniform_k4 && problems[i].K == problems[0].K;
    const long tail = t256 % NCU, full = t256 - tail;
    for (int i = 1; i < n; ++i) uniform_k4 = uniform_k4 && problems[i].k_segs == problems[0].k_segs;
    const int kt_all = kmax / BK;                      // K tiles of a tile, all segments of a split-bf16 problem counted
    const bool can_split = env.sk && workspace && ws_bytes >= SK_WS_BYTES && ((uintptr_t)workspace & 255) == 0 && uniform_k4 && kt_all >= 16 &&
                           tail > 0 && tail * 2 <= 256 && tail * 2 <= NCU && rounds <= env.sk_max_rounds;
    bool split_all = can_split && full == 0 && kt_all >= env.pair_min_kt;      // (the pair kernel's shapes)
    // a tail of up to a third of a round; up to half a round where the exchange is a small part of the tile: long K (>= 96 K tiles) and no
    // q/k/v epilogue in the launch (the 1024 x 1024 batch-4 N = 3072 projections, 1632 tiles = 6 rounds + 96: 0.2594 / 0.2584 -> 0.2651 /
    // 0.2650 images/s, profiles/r04z_*; the double blocks' q/k/v launch at batch 1, 104 tail tiles at K = 3072: 146 vs 136 us for the
    // 8-wave mixed plan's half-height tiles -- stays there). LX_GEMM4_TAIL_DIV = d forces tail * d <= CUs (A/B).
    const int tail_div = env.sk_tail_div > 0 ? env.sk_tail_div : (!qkv && kt_all >= 96 ? 2 : 3);
    bool split_tail = can_split && full > 0 && tail * tail_div <= NCU;      // (a tail of more than a third of a round: the 8-wave mixed plan's half-height tiles win -- the double blocks' q/k/v launch, 104 tail tiles: 154 vs 136 us)
    if (split) {
      // precise mode (two or three passes over K: the K-independent cost of a round and of the exchange weigh a third as much as on the
      // bf16 path; no q/k/v epilogue, no mixed plan to compete with): by cost -- measured slopes per K tile and round, 1.31 us for this
      // kernel (8.9 fixed, ~6 for an exchange), 1.55 / 1.06 for the 8-wave kernels at 256 / 128 rows (tools/gemm_ksweep.py)
      const double r4 = 8.9 + 1.31 * kt_all;
      const bool sk_tail = can_split && tail > 0;
      const double c4 = (double)(full / NCU) * r4 + (tail == 0 ? 0.0 : sk_tail ? 8.9 + 1.31 * (kt_all - kt_all / 2) + 6.0 : r4);
      const double c8a = (double)((t256 + NCU - 1) / NCU) * (5.8 + 1.55 * kt_all), c8b = (double)((t128 + NCU - 1) / NCU) * (10.5 + 1.06 * kt_all);
      fills = env.g4 == 2 || c4 < (c8a < c8b ? c8a : c8b);
      split_all = fills && sk_tail && full == 0;
      split_tail = fills && sk_tail && full > 0;
      if (!fills) split_all = split_tail = false;
    }
    if (ok && (fills || split_all || split_tail)) {
      GemmArgs all;
      all.n = 0;
      all.tile_start[0] = 0;
      for (int i = 1; i <= MAX_SUB; ++i) all.tile_start[i] = 0;
      for (int i = 0; i < n; ++i) plan_add(all, problems[i], 0, 256);
      if (split_all || split_tail) {
        float* slots = (float*)workspace;
        int* flags = (int*)((char*)workspace + (size_t)256 * SK_SLOT_FLOATS * sizeof(float));
        int* err = (int*)((char*)workspace + PAIR_OFF + (size_t)PAIR_MAX_WG * PAIR_SLOT_FLOATS * sizeof(float)) + PAIR_MAX_WG;
        if (split) hipLaunchKernelGGL(lx_gemm4_kernel<true>, dim3((unsigned)(full + 2 * tail)), dim3(G4_THREADS), 0, s, all, (int)full, env.g4_fault ? 3 : 2, slots, flags, err);
        else hipLaunchKernelGGL(lx_gemm4_kernel<false>, dim3((unsigned)(full + 2 * tail)), dim3(G4_THREADS), 0, s, all, (int)full, env.g4_fault ? 3 : 2, slots, flags, err);
      } else if (split)
        hipLaunchKernelGGL(lx_gemm4_kernel<true>, dim3((unsigned)t256), dim3(G4_THREADS), 0, s, all, (int)t256, 1, (float*)nullptr, (int*)nullptr, (int*)nullptr);
      else
        hipLaunchKernelGGL(lx_gemm4_kernel<false>, dim3((unsigned)t256), dim3(G4_THREADS), 0, s, all, (int)t256, 1, (float*)nullptr, (int*)nullptr, (int*)nullptr);
      LX_LAUNCH_CHECK("lx_gemm_bf16");
      return LX_OK;
    }
  }
  if (split) {   // precise mode on the 8-wave kernels: all 256-row tiles or all 128-row tiles (no mixed / pair plans)
    const double ca = (double)((t256 + NCU - 1) / NCU) * round_us(256, kmax), cb = (double)((t128 + NCU - 1) / NCU) * round_us(128, kmax);
    const int bm = forced ? forced : (cb < ca ? 128 : 256);
    GemmArgs all;
    all.n = 0;
    all.tile_start[0] = 0;
    for (int i = 1; i <= MAX_SUB; ++i) all.tile_start[i] = 0;
    for (int i = 0; i < n; ++i) plan_add(all, problems[i], 0, bm);
    return launch_plan(all, bm, s, true);
  }
  // Two workgroups per 256-row tile (lx_gemm_pair_kernel) when there are at most 128 such tiles: needs one K for the whole
  // group, a 256-CU device and the scratch slots. LX_GEMM_PAIR = 0 never | 1 (default) where it measures faster (K >= 6144:
  // the swap costs ~6 us, the better loop saves 0.12 us per K tile) | 2 whenever possible (tests).
  {
    const int pair_mode = env.pair;
    bool uniform_k = true;
    for (int i = 1; i < n; ++i) uniform_k = uniform_k && problems[i].K == problems[0].K;
    const int per_xcd = (int)(t256 / 8 + (t256 % 8 ? 1 : 0));
    const bool fits = per_xcd * 16 <= PAIR_MAX_WG && kmax / BK >= 2;
    const bool pays = pair_mode == 2 || kmax / BK >= env.pair_min_kt;
    if (workspace) LX_CHECK_ARG(ws_bytes >= SK_WS_BYTES && ((uintptr_t)workspace & 255) == 0, "lx_gemm_bf16_ws: workspace needs %zu bytes (lx_gemm_workspace_bytes()), 256-byte aligned", SK_WS_BYTES);
    if (workspace && pair_mode && forced == 0 && uniform_k && fits && pays && !qkv && NCU == PAIR_MAX_WG) {
      {
        float* slots = (float*)((char*)workspace + PAIR_OFF);
        int* flags = (int*)((char*)workspace + PAIR_OFF + (size_t)PAIR_MAX_WG * PAIR_SLOT_FLOATS * sizeof(float));
        GemmArgs all;
        all.n = 0;
        all.tile_start[0] = 0;
        for (int i = 1; i <= MAX_SUB; ++i) all.tile_start[i] = 0;
        for (int i = 0; i < n; ++i) plan_add(all, problems[i], 0, 256);
        hipLaunchKernelGGL(lx_gemm_pair_kernel, dim3(per_xcd * 16), dim3(NTHREADS), 0, s, all, slots, flags);
        LX_LAUNCH_CHECK("lx_gemm_bf16");
        return LX_OK;
      }
    }
  }
  // candidate schedules: all 256-row tiles, all 128-row tiles, or full rounds of 256-row tiles + a 128-row-tile tail
  const double c_a = (double)((t256 + NCU - 1) / NCU) * round_us(256, kmax);
  const double c_b = (double)((t128 + NCU - 1) / NCU) * round_us(128, kmax);
  GemmArgs big, tail;
  big.n = tail.n = 0;
  big.tile_start[0] = tail.tile_start[0] = 0;
  for (int i = 1; i <= MAX_SUB; ++i) big.tile_start[i] = tail.tile_start[i] = 0;
  double c_c = 1e30;
  const long full = (t256 / NCU) * NCU;
  if (forced == 0 && full > 0 && t256 > full) {
    // peel 256-row tile-rows off the ends of the problems (last problem first) until the big launch fits in full rounds
    long remain = t256;
    int keep_rows[LX_GEMM_MAX_GROUP];
    for (int i = 0; i < n; ++i) keep_rows[i] = problems[i].M;
    for (int i = n - 1; i >= 0 && remain > full; --i) {
      const int tn = (problems[i].N + BN - 1) / BN;
      while (remain > full && keep_rows[i] > 0) {
        const int last_rows = keep_rows[i] % 256 ? keep_rows[i] % 256 : 256;
        keep_rows[i] -= last_rows;
        remain -= tn;
      }
    }
    long tail_tiles = 0;
    for (int i = 0; i < n; ++i) {
      if (keep_rows[i] > 0) plan_add(big, sub_rows(problems[i], 0, keep_rows[i]), 0, 256);
      if (keep_rows[i] < problems[i].M) {
        const lx_gemm_desc q = sub_rows(problems[i], keep_rows[i], problems[i].M - keep_rows[i]);
        plan_add(tail, q, keep_rows[i], 128);
        tail_tiles += tiles_of(q, 128);
      }
    }
    c_c = (double)((remain + NCU - 1) / NCU) * round_us(256, kmax) + (double)((tail_tiles + NCU - 1) / NCU) * round_us(128, kmax) + 2.0;
  }
  int choice = forced == 256 ? 0 : forced == 128 ? 1 : (c_c < c_a && c_c < c_b) ? 2 : (c_b < c_a ? 1 : 0);
  if (choice == 2) {
    const bool one_grid = env.one_grid != 0;
    if (one_grid && big.n > 0 && tail.n > 0) {
      const int n_big = big.tile_start[big.n], n_big_pad = (n_big + 7) & ~7;
      hipLaunchKernelGGL(lx_gemm_mixed_kernel, dim3(n_big_pad + tail.tile_start[tail.n]), dim3(NTHREADS), 0, s, big, tail, n_big_pad);
      LX_LAUNCH_CHECK("lx_gemm_bf16");
      return LX_OK;
    }
    const int rc = launch_plan(big, 256, s);
    if (rc != LX_OK) return rc;
    return launch_plan(tail, 128, s);
  }
  GemmArgs all;
  all.n = 0;
  all.tile_start[0] = 0;
  for (int i = 1; i <= MAX_SUB; ++i) all.tile_start[i] = 0;
  for (int i = 0; i < n; ++i) plan_add(all, problems[i], 0, choice == 0 ? 256 : 128);
  return launch_plan(all, choice == 0 ? 256 : 128, s);
}
