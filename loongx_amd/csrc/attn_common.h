// attn_common.h -- what the attention translation units (attn.hip: the 8-wave kernels; attn4.hip: the one-wave-per-SIMD kernel) share:
// the launch argument block, the K / V^T tile geometry and the row-per-lane epilogue store.
#pragma once
#include "common.h"

namespace {

// A value defined behind an empty asm INSIDE a rarely taken branch cannot be hoisted out of the loop around it. Without it hipcc's
// LICM turned the 32 key constants of the ragged-tile masks into registers held (or spilled) across the whole tile loop: 30 VGPRs of
// lx_attn_pipe_kernel (254 -> 224) and 33 spilled registers of lx_attn_fp8_pipe_kernel<true, true> (-> 0).
#ifdef LX_ATTN_NO_LICM_FIX
#define LX_PIN_IN_BRANCH(x)
#else
#define LX_PIN_IN_BRANCH(x) asm volatile("" : "+v"(x))
#endif

constexpr int DH = 128;
constexpr int KVBLK = 64;
constexpr int K_BYTES = KVBLK * DH * 2;   // 16 KiB
constexpr int V_BYTES = DH * KVBLK * 2;   // 16 KiB
constexpr int STAGE_BYTES = K_BYTES + V_BYTES;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// XCD-aware work order (guide T1; round 6). The dispatcher is observed to place workgroup w on XCD w % 8 and each XCD has its own 4-MiB L2.
// With head-minor numbering (head = w % BH) the 32 workgroups an XCD runs at a time are ~32 DIFFERENT (batch, head) pairs: every one streams
// its own K / V^T image and nothing is shared in that L2 -- at batch 16 (384 heads x 1.3 MB, 10 query tiles each) a launch pulls 5 GB
// through the fabric, 10x the images' size. Head-major numbering with a contiguous block of it per XCD makes an XCD's concurrent workgroups
// the query tiles of ~3 heads: a K / V^T tile is fetched once per XCD and served to the other query tiles from L2. A bijection on [0, n)
// for every n (XCD x owns (n / 8) + (x < n % 8) consecutive items); placement is a speed assumption only, never a correctness one.
#ifndef LX_ATTN_XCD_ORDER
#define LX_ATTN_XCD_ORDER 1
#endif
#ifndef LX_ATTN_XCD_QT
#define LX_ATTN_XCD_QT 12
#endif
__device__ __forceinline__ int lx_xcd_order(int w, int n) {
#if LX_ATTN_XCD_ORDER
  const int q = n >> 3, r = n & 7, x = w & 7, j = w >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
#else
  return w;
#endif
}
// item -> (query tile, batch-head): head-major under the XCD order (an XCD's consecutive items = the query tiles of one head), head-minor
// (the numbering of rounds 1-5) without it
__device__ __forceinline__ void lx_item_decode(int w, int n_items, int BH, int n_qt, int& qt, int& bh) {
#if LX_ATTN_XCD_ORDER
  // ... of a GROUP of G heads when a head has many query tiles (1024x1024: 34): 32 workgroups walking ONE K / V^T image in step ask the same
  // L2 lines at the same moment (measured: the e4m3 kernel 3.6 % slower than with no sharing at all); G heads x 32 / G query tiles keep the
  // sharing and spread the requests over G address streams. G = 1 up to LX_ATTN_XCD_QT query tiles per head, then doubling.
  const int L = lx_xcd_order(w, n_items);
  int G = 1;
  while (n_qt > LX_ATTN_XCD_QT * G && G < 8) G *= 2;
  const int per = G * n_qt, g = L / per, i = L - g * per, Gg = min(G, BH - g * G);
  qt = i / Gg;
  bh = g * G + (i - qt * Gg);
#else
  qt = w / BH;
  bh = w - qt * BH;
#endif
}

struct AttnArgs {
  lx_attn_desc d;
  int qt_start[4];   // prefix of (32*NW)-row query tiles per segment
  int wide_store;    // O rows and columns are 16-byte aligned: the epilogue stores 16 B per lane (lx_store_o)
};

// Epilogue store of one query row per lane pair: O[q, d] = O^T / l; lane (q = lane & 31, half = lane >> 5) holds
// d = db*32 + 8*rq + 4*half + (0..3), i.e. 8 bytes of bf16 per (db, rq), and the two halves of a row hold ADJACENT 8-byte groups.
// wide: for each pair of groups (rq, rq+1) one v_permlane32_swap per dword hands the lower half-wave the upper half's group rq and
// the upper half-wave the lower half's group rq+1: every lane then owns 16 contiguous bytes -> 8 dwordx4 stores per lane instead of
// 16 dwordx2, same bytes, same addresses (the store tail of a row-per-lane epilogue is store-ISSUE bound: guide T21).
// Output format (lx_attn_desc.flags & LX_ATTN_O_F16): bf16, or the fp16 operand image of an LX_OPERANDS_F16 projection (to_out / proj_out:
// nearest even, saturated; `mx` collects max |o| for the overflow report). The format is a per-value SELECT, not a branch: the accumulators
// are read at ONE place whatever the format. lx_attn4_kernel keeps O^T in AGPRs behind inline-asm MFMAs across its persistent item loop,
// and a second read site in another branch made hipcc reconcile AGPR assignments at the join -- moves its hazard recogniser cannot order
// against MFMAs it cannot see: the kernel turned nondeterministic (round 5). 64 extra conversions per row and item: not measurable.
__device__ __forceinline__ uint32_t lx_pack_o(float lo, float hi, bool f16, float& mx) {
  const uint32_t b = pack_bf16x2(lo, hi);
  const uint32_t h = pack_f16x2_sat(lo, hi, mx);
  return f16 ? h : b;
}
// (the swap inside exchanges data between the two half-waves of a row: both halves of a row must be valid or invalid together; `valid`
//  guards the stores of rows past the segment.) `mode` = lx_o_mode(args): bit 0 wide stores, bit 1 fp16 output; `ovf` = the overflow
//  word. Both are VALUES the kernel computes once, up front: no kernel-argument load inside lx_attn4_kernel's item loop.
__device__ __forceinline__ int lx_o_mode(const AttnArgs& args) { return (args.wide_store != 0 ? 1 : 0) | ((args.d.flags & LX_ATTN_O_F16) ? 2 : 0); }
__device__ __forceinline__ void lx_store_o(int mode, int* ovf, bool valid, uint16_t* row, const f32x16 (&oacc)[4], float inv, int lhi) {
  float mx = 0.f;
  const bool f16 = (mode & 2) != 0;
  if (valid) {
    if (mode & 1) {
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int rq = 0; rq < 4; rq += 2) {
          const uint32_t a0 = lx_pack_o(oacc[db][rq * 4 + 0] * inv, oacc[db][rq * 4 + 1] * inv, f16, mx);
          const uint32_t a1 = lx_pack_o(oacc[db][rq * 4 + 2] * inv, oacc[db][rq * 4 + 3] * inv, f16, mx);
          const uint32_t b0 = lx_pack_o(oacc[db][rq * 4 + 4] * inv, oacc[db][rq * 4 + 5] * inv, f16, mx);
          const uint32_t b1 = lx_pack_o(oacc[db][rq * 4 + 6] * inv, oacc[db][rq * 4 + 7] * inv, f16, mx);
          const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
          const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
          *(u32x4*)(row + db * 32 + 8 * (rq + lhi)) = u32x4{r0[0], r1[0], r0[1], r1[1]};
        }
    } else {
      uint16_t* op = row + 4 * lhi;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          u32x2 o;
          o[0] = lx_pack_o(oacc[db][rq * 4 + 0] * inv, oacc[db][rq * 4 + 1] * inv, f16, mx);
          o[1] = lx_pack_o(oacc[db][rq * 4 + 2] * inv, oacc[db][rq * 4 + 3] * inv, f16, mx);
          *(u32x2*)(op + db * 32 + rq * 8) = o;
        }
    }
  }
  if (f16) report_f16_overflow(mx, ovf);
}

}  // namespace

// attn4.hip: launches lx_attn4_kernel (one wave per SIMD; bounded-score contract only) on a validated AttnArgs block (256-row query tiles)
int lx_attn4_launch(const void* attn_args, int n_items, int mode, void* stream);
int lx_attn4_cus(void);   // compute units of the current device (one persistent workgroup each)
