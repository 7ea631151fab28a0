// gemm8.h -- the 8-wave GEMM kernels for gfx950 (MI355X): kernel TEMPLATES, instantiated by gemm.hip (bf16 operands), gemm_f16.hip
// (fp16 operands) and gemm_modes.hip (split-bf16 "precise" and e4m3 operands).
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
//  * v_mfma_f32_32x32x16_bf16 (_f16 with LX_OPERANDS_F16: same shapes, same register layout) with the weight tile as the MFMA "A"
//    operand, so the accumulator layout is lane = output row m, registers = 4 consecutive output columns n -> vector epilogue
//    loads/stores;
//  * per-wave tile (BM/2) x 64: 2 W-fragments + BM/64 X-fragments feed 2*BM/64 MFMAs per 16-deep k step;
//  * epilogue fuses bias, rank-r LoRA up-projection, GELU(tanh), and the gated residual accumulate
//    X += gate * y in fp32 (block.py:224-234,269-272,326-334);
//  * blockIdx -> tile map is XCD-aware: each of the 8 XCDs (private L2) gets a contiguous run of tiles,
//    ordered in 4-tile-tall column groups so co-resident tiles share A / W panels in that L2.
#pragma once
#include "gemm_common.h"

#ifndef LX_ACC_AGPR
#define LX_ACC_AGPR 0
#endif

namespace {

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


// ---- LX_EPI_QKV: RMSNorm(128) + RoPE on the k / q columns, V^T image for the v columns, inside the projection's epilogue --------
// (block.py:60-99: attn.norm_q / norm_k, apply_rotary_emb; replaces the qkv_prep pass over the projected buffer: one read + one
//  write of 3 D columns per token, 22 us x 57 launches per denoise step at S = 2560.) A 256-column tile is two whole heads of
// one kind (qkv_d % 256 == 0); a wave holds 64 columns, so the sum of squares of a head's row is the sum of two waves' partial
// sums, exchanged through LDS once per tile. Everything is computed in fp32 on the accumulators: one bf16 rounding instead of
// the two of the separate pass. (With fp16 operands the outputs are still bf16: they are the attention kernel's operands.)
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

template <int BM, int MI, bool SPLIT = false, bool FP8 = false, bool F16 = false>
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

  static_assert(!(F16 && (SPLIT || FP8)), "fp16 operands exclude the split-bf16 and e4m3 modes");
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
    float f16_mx = 0.f;                              // LX_OPERANDS_F16: max |x| of what this lane rounded to fp16 (pack_f16x2_sat)
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
          // (fp16 operands: the store is the next GEMM's A operand -- fp16, nearest even, saturated)
          u32x4 o = {pack_op16x2<F16>(v0[0], v0[1], f16_mx), pack_op16x2<F16>(v0[2], v0[3], f16_mx), pack_op16x2<F16>(v1[0], v1[1], f16_mx),
                     pack_op16x2<F16>(v1[2], v1[3], f16_mx)};
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
    if constexpr (F16) report_f16_overflow(f16_mx, P.f16_ovf);
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

// K tiles [kt0, kt1) of output tile (m0, n0) accumulated into acc (which the caller has cleared). `after_issue` runs between the
// issue of the prologue's operand DMA and the wait for it. `smem` = the workgroup's LDS
// buffer (gemm_lds_bytes<BM>() bytes, 1 KiB aligned). On return no wave reads the operand rings any more.
template <int BM, bool SPLIT = false, bool F16 = false, class F>
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
      // (fp16 operands: the fragments are the same 16 bytes per lane, the MFMA the same shape; only the element format differs)
      if constexpr (F16) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wf[j]), __builtin_bit_cast(f16x8, xf[i]), acc[j][i], 0, 0, 0);
      else if (LX_ACC_AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[j][i]) : "v"(wf[j]), "v"(xf[i]));
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
template <int BM, bool SPLIT = false, bool F16 = false>
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
  gemm_mainloop<BM, SPLIT, F16>(P, m0, n0, tn, 0, nkt, smem, acc, tid, [&]() {
    if (lora_early) {
      f32x4 t4[MI];
      lora_sum<MI>(P, 0, sv, t4);
      lora_apply<MI>(u4, t4, lhi, acc);
    }
  });
  gemm_epilogue<BM, MI, SPLIT, false, F16>(P, acc, smem, m0, n0, args.m_base[g], wave, wm, wn, lane, l31, lhi, lora_early);
}

template <int BM, bool F16 = false>
__global__ __launch_bounds__(NTHREADS) void lx_gemm_kernel(const GemmArgs args) {
  __shared__ __attribute__((aligned(1024))) char smem[gemm_lds_bytes<BM>()];
  gemm_tile<BM, false, F16>(args, blockIdx.x, smem);
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
template <bool F16 = false>
__global__ __launch_bounds__(NTHREADS) void lx_gemm_mixed_kernel(const GemmArgs big, const GemmArgs tail, const int n_big_pad) {
  __shared__ __attribute__((aligned(1024))) char smem[gemm_lds_bytes<128>() > gemm_lds_bytes<256>() ? gemm_lds_bytes<128>() : gemm_lds_bytes<256>()];
  const int bid = blockIdx.x;
  if (bid < big.tile_start[MAX_SUB]) gemm_tile<256, false, F16>(big, bid, smem);
  else if (bid >= n_big_pad) gemm_tile<128, false, F16>(tail, bid - n_big_pad, smem);
}


}  // namespace
