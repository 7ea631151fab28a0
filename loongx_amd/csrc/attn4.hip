// attn4.hip -- lx_attn4_kernel: the joint attention of attn.hip (same contract, same K / V^T images, same bounded-score softmax) with
// ONE wave per SIMD. Replaces F.scaled_dot_product_attention + the mask / c_factor bias of attn_forward (src/flux/block.py:101-135)
// wherever the caller passes LX_ATTN_Q_LOG2 | LX_ATTN_BOUNDED (the engine's default for bf16 attention).
//
// Why another shape: in the 8-wave kernel every wave reads the whole 32-KiB K / V^T tile from LDS for its 32 query
// rows -- one 1-KiB fragment per MFMA, 16-28 % of the launch by elimination -- and the two waves of a SIMD add their issue streams. Here
// a workgroup is 4 waves x 64 query rows (two 32-row blocks per wave): every K / V^T fragment feeds TWO MFMAs, half the waves issue
// LDS-DMA, and a wave owns its SIMD's 512 registers: O^T (2 x 4 x 16) and the Q fragments (2 x 8 x 4) live in AGPRs; ONE score set
// (2 x 2 x 16), the P words (2 x 4 x 4) and a four-deep fragment ring in the architectural VGPRs. The matrix instructions and the
// softmax are inline asm in a fixed order (one fragment "slot" = wait, MFMA, vector fillers, MFMA, the ds_read for four slots ahead,
// fillers); hipcc allocates registers and emits the scalar code, the LDS-DMA pieces and the epilogue. The loop body is NOT expanded:
// with one copy of the body every value has one live range around the loop and hipcc's allocator has nothing to split (a first version
// with two score sets and a body expanded six times for immediate ring offsets spilled 962 registers; pinning its operands to physical
// registers turned the accumulators into VGPR-class values and spilled 1481).
//
// One frame T of the loop = 32 slots = 64 MFMAs = a modulo schedule over three tiles (the 64 keys of a tile are two 32-key blocks
// kb = 0 / 1 whose scores live in separate registers, and four 16-key slices s = 0..3 for P.V; slices 0, 1 belong to kb 0):
//   slots  0-1   P.V slice 2 of tile T-1, d blocks 2, 3        vector stream (160 instructions per frame, in order):
//   slots  2-5   P.V slice 3 of tile T-1                          softmax slices 0, 1, 2, 3 of tile T = per lane 16 exp2, 16 row-sum adds and
//   slots  6-13  scores of tile T, key block 1                    8 cvt_pk each; slices 0, 1 read the kb-0 scores (complete since slot 29 of the
//   slots 14-17  P.V slice 0 of tile T                            previous frame, overwritten from slot 22 on), slices 2, 3 the kb-1 scores
//   slots 18-21  P.V slice 1 of tile T                            (complete at slot 13, overwritten from slot 6 of the next frame)
//   slots 22-29  scores of tile T+1, key block 0
//   slots 30-31  P.V slice 2 of tile T, d blocks 0, 1
// K and V^T tiles sit in two three-deep rings (96 KiB): in frame T the kb-1 half of K(T), K(T+1), V^T(T-1) and V^T(T) are read, K(T+2)
// is staged behind the frame's barrier (slot 8; four 1-KiB LDS-DMA pieces per wave) and V^T(T+1) in slots 24-27; both are waited for at
// slot 8 of the NEXT frame, so staging latency is never exposed. Ring positions are run-time: the eight K and four V^T fragment
// addresses move by one position per frame (twelve v_add).
//
// Hazards the assembler does not see (inline asm is opaque to hipcc's hazard recogniser; gfx940-class rules):
//   MFMA result -> VALU read (11 wait states for an 8-pass MFMA): kb-0 scores are complete at slot 29 and first read in slot 0 of the next
//     frame, kb-1 scores complete at slot 13 and first read in slot 16 (static_asserts below): two slots = at least 14 instructions.
//   v_exp_f32 result -> the next VALU (trans forwarding, 1 wait state): the stream is [exp e(u), exp o(u), add e(u-1), add o(u-1),
//     cvt(u-1)]: nothing reads an exp2 result in the instruction after it.
//   MFMA result in AGPRs -> v_accvgpr_read (epilogue): s_nop 15 + s_nop 7 behind the loop.
#include "attn_common.h"

namespace {

constexpr int A4_KV = 16384;                 // one K tile [64 keys][128 d] or one V^T tile [128 d][64 keys], bf16
constexpr int A4_VB = 3 * A4_KV;             // the V^T ring starts behind the K ring
constexpr int A4_LDS = 6 * A4_KV;            // 96 KiB
static_assert(A4_LDS <= 160 * 1024, "LDS budget");

// ---- the frame's schedule (compile-time tables) ----
// slot -> matrix work: kind 0 = score fragment (ks = f, key block kb), kind 1 = P.V fragment f = s * 4 + db
constexpr int a4_kind(int i) { return (i >= 6 && i < 14) || (i >= 22 && i < 30) ? 0 : 1; }
constexpr int a4_kb(int i) { return i < 14 ? 1 : 0; }                                       // score slots: key block
constexpr int a4_ks(int i) { return (i < 14 ? i - 6 : i - 22) & 7; }                        // score slots: 16-wide d step
constexpr int a4_pvf(int i) { return (i < 6 ? 10 + i : i < 22 ? i - 14 : i - 22) & 15; }    // P.V slots: 0-1 -> 10, 11; 2-5 -> 12..15; 14-21 -> 0..7; 30-31 -> 8, 9
// Row sums: one v_add_f32 per probability, taken from the UNROUNDED p (fp32), summed per lane and across the two half-waves at the end.
// (Summing on the matrix pipe instead -- an all-ones MFMA per slice and query block -- was built and measured in round 4: 8 MFMAs = 256 pipe
// cycles per frame to save 53 exposed ones, profiles/r04a_attn4_ab.txt; removed in round 5.)
constexpr int A4_SL = 40;                        // vector instructions per 16-key slice (both query blocks): 16 exp2 + 16 add + 8 cvt_pk
// vector instructions per slot: none in the eight slots that carry an LDS-DMA piece
constexpr int a4_nslot(int i) {
  constexpr int n[32] = {7, 7, 6, 7, 7, 6, 7, 6, 0, 0, 0, 0, 7, 7, 7, 6, 7, 7, 6, 7, 7, 6, 7, 6, 0, 0, 0, 0, 7, 7, 7, 6};
  return n[i];
}
constexpr int a4_ngap(int g) { return (g & 1) ? a4_nslot(g >> 1) / 2 : (a4_nslot(g >> 1) + 1) / 2; }   // gap 2i: behind the slot's first MFMA
constexpr int a4_pos(int g) { int p = 0; for (int k = 0; k < g; ++k) p += a4_ngap(k); return p; }      // stream position at the start of gap g (0..64)
static_assert(a4_pos(64) == 4 * A4_SL, "four slices of vector instructions per frame");
static_assert(a4_pos(28) >= A4_SL && a4_pos(36) >= 2 * A4_SL && a4_pos(60) >= 3 * A4_SL, "a slice's P words are complete before its P.V slots (14, 18, 30)");
static_assert(a4_pos(32) <= 2 * A4_SL, "slices 2, 3 read the kb-1 scores: not before slot 16 (complete at slot 13 + 11 wait states)");
static_assert(a4_pos(44) >= 2 * A4_SL, "slices 0, 1 read the kb-0 scores: done before slot 22 overwrites them");
static_assert(a4_pos(12) <= 3 * A4_SL && a4_pos(4) <= 2 * A4_SL, "P words of slices 2 / 3 are rewritten only after the previous tile's P.V slots 0-1 / 2-5");
// The softmax of one 16-key slice is a stream of vector instructions over 8 pair units u = pair * 2 + query block:
//   exp2 e(0), o(0) | for u = 1..7: exp2 e(u), exp2 o(u), add e(u-1), add o(u-1), cvt_pk(u-1) | add e(7), add o(7), cvt_pk(7)    (40)
// kind: 0 exp2, 1 add, 2 cvt_pk
constexpr int a4_op_kind(int n) { return n < 2 ? 0 : n >= 37 ? (n == 39 ? 2 : 1) : ((n - 2) % 5 < 2 ? 0 : (n - 2) % 5 < 4 ? 1 : 2); }
constexpr int a4_op_unit(int n) { return n < 2 ? 0 : n >= 37 ? 7 : ((n - 2) % 5 < 2 ? (n - 2) / 5 + 1 : (n - 2) / 5); }
constexpr int a4_op_half(int n) { return n < 2 ? n : n >= 37 ? (n - 37) & 1 : ((n - 2) % 5 < 2 ? (n - 2) % 5 : ((n - 2) % 5 - 2) & 1); }
// Fragment ring: LX_A4_LOOK reads in flight (4 or 8: it has to divide 32)
#ifndef LX_A4_LOOK
#define LX_A4_LOOK 4
#endif
constexpr int A4_LOOK = LX_A4_LOOK;
static_assert(A4_LOOK == 4 || A4_LOOK == 8, "ring depth");
// lgkmcnt for position i of a fragment sequence of n_seq reads (-1: no wait here): min(LOOK, n_seq - i) reads are outstanding in front of it
constexpr int a4_wait(int i, int n_seq) {
  const int out = n_seq - i < A4_LOOK ? n_seq - i : A4_LOOK;
  return out - 1;
}
// LDS-DMA pieces: K(T+2) pieces 0-3 behind slots 8-11 (right behind the barrier), V^T(T+1) pieces 0-3 behind slots 24-27
constexpr int a4_kpiece(int i) { return i >= 8 && i < 12 ? i - 8 : -1; }
constexpr int a4_vpiece(int i) { return i >= 24 && i < 28 ? i - 24 : -1; }

template <int MODE>   // 1: p = exp2(s)   2: p = exp2(s + bias): the bias of the (query, key) segment pair is the srcC of the first score MFMA
__global__ __launch_bounds__(256, 1) void lx_attn4_kernel(const AttnArgs args, const int n_items) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int QBLK = 256;
  __shared__ __attribute__((aligned(1024))) char smem[A4_LDS];
  const lx_attn_desc& D = args.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int n_cta = gridDim.x;

  // descriptor fields -> scalars, once (a kernarg array indexed with a running index inside the loops is a dependent s_load)
  const int BH = D.B * D.H, NH = D.H;
  const int n_seg = D.n_seg;
  const int len0 = D.seg_len[0], len1 = D.seg_len[1], len2 = D.seg_len[2];
  const int row00 = D.seg_row0[0], row01 = D.seg_row0[1], row02 = D.seg_row0[2];
  const int vt00 = D.seg_vt0[0], vt01 = D.seg_vt0[1], vt02 = D.seg_vt0[2];
  const float b00 = D.bias[0][0], b01 = D.bias[0][1], b02 = D.bias[0][2], b10 = D.bias[1][0], b11 = D.bias[1][1], b12 = D.bias[1][2],
              b20 = D.bias[2][0], b21 = D.bias[2][1], b22 = D.bias[2][2];
  const int qs1 = args.qt_start[1], qs2 = args.qt_start[2];
  // the output format / store shape and the overflow word, as register VALUES before the item loop (see lx_store_o: no kernel-argument
  // load may appear inside that loop)
  int o_mode = __builtin_amdgcn_readfirstlane(lx_o_mode(args));
  uint32_t ovf_lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)D.f16_ovf), ovf_hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)D.f16_ovf >> 32));
  asm volatile("" : "+s"(o_mode), "+s"(ovf_lo), "+s"(ovf_hi));
  int* const o_ovf = (int*)(((uintptr_t)ovf_hi << 32) | (uintptr_t)ovf_lo);
  auto pick = [](int s, auto x0, auto x1, auto x2) { return s == 0 ? x0 : (s == 1 ? x1 : x2); };
  auto seg_len = [&](int s) { return pick(s, len0, len1, len2); };
  const int ldk = D.ldk, vt_ld = D.vt_ld, ldq = D.ldq;

  // ---- work items: one item = one 256-row query tile of one (batch, head). A workgroup walks items blockIdx.x, + gridDim.x, ... : the
  // launch is PERSISTENT when there are more items than CUs. Between items only O / l / Q change hands: the K / V^T tile stream, its rings
  // and the frame pipeline run on across the boundary (the generator below hands out the next item's tiles behind the last tile of this
  // one), so the next item's first tiles are staged under this item's last frames and its Q is fetched under the last frame. Measured
  // before this (profiles/r04a_attn4_cycles.txt): prologue + epilogue + dispatch = 18-20 k of a workgroup's 105-125 k cycles at 32-40 tiles. ----
  struct Item { int w, b, h, bh, sq, q_len, q_row0, q_tile0; };   // q_row0: first row of the item's query segment and batch; q_tile0: the tile's first row in it
  auto decode = [&](int w) {
    Item it;
    it.w = w;
    int qt;
    lx_item_decode(w, n_items, BH, args.qt_start[3], qt, it.bh);
    it.b = it.bh / NH;
    it.h = it.bh - it.b * NH;
    it.sq = (n_seg > 2 && qt >= qs2) ? 2 : ((n_seg > 1 && qt >= qs1) ? 1 : 0);
    it.q_len = seg_len(it.sq);
    it.q_row0 = pick(it.sq, row00, row01, row02) + it.b * it.q_len;
    it.q_tile0 = (qt - pick(it.sq, 0, qs1, qs2)) * QBLK;
    return it;
  };
  // Q fragments of an item: lane (q = l31, half = lhi) of query block qb holds d = ks*16 + lhi*8 .. +8
  auto load_q = [&](const Item& it, bf16x8 (&q)[2][8]) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const size_t q_row = (size_t)it.q_row0 + min(it.q_tile0 + wave * 64 + l31 + 32 * qb, it.q_len - 1);
      const __bf16* qp = (const __bf16*)D.Q + q_row * ldq + D.q_col + it.h * DH + lhi * 8;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) q[qb][ks] = *(const bf16x8*)(qp + ks * 16);
    }
  };

  // ---- staging: piece j (0..3) of this wave = 1 KiB = K rows (j*4 + wave)*4 .. +4 (256 B each) / V^T rows (j*4 + wave)*8 .. +8 (128 B).
  // The buffer descriptors cover the whole K column block / V^T image; the item's head and the tile's rows are the scalar offset. ----
  const lx_rsrc_t rs_k = lx_make_rsrc((const __bf16*)D.K + D.k_col);
  const lx_rsrc_t rs_v = lx_make_rsrc((const __bf16*)D.VT);
  uint32_t k_off[4], v_off[4];
  uint32_t k_slot_off;
  {
    const int key0 = wave * 4 + (lane >> 4);                   // + 16 j
    k_slot_off = (uint32_t)((((lane & 15) ^ (key0 & 15)) * 8) * 2);
    const int drow0 = wave * 8 + (lane >> 3);                  // + 32 j: (drow >> 1) & 7 does not depend on j
    const uint32_t v_slot_off = (uint32_t)((((lane & 7) ^ ((drow0 >> 1) & 7)) * 8) * 2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      k_off[j] = (uint32_t)((key0 + 16 * j) * ldk * 2) + k_slot_off;
      v_off[j] = (uint32_t)((drow0 + 32 * j) * vt_ld * 2) + v_slot_off;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(k_off[j]), "+v"(v_off[j]));   // (kept in registers: rematerialised they are the multiplies again)
  // K rows are clamped to the tile's last valid key (ragged segment tails; a tile that does not exist is staged from row 0 of the
  // previous one and never used): min(row, clamp) * ld + slot = min(row * ld + slot, clamp * ld + slot). ring_off: byte offset of the
  // ring position (0, 16384, 32768); ksoff / vsoff: byte offset of the tile's first key row (+ head) / first V^T column (+ batch, head)
  const int piece_lds = wave * 1024;
  auto kpiece = [&](int j, int ring_off, int ksoff, int nclamp) {
    const uint32_t off_ = min(k_off[j], (uint32_t)((nclamp - 1) * ldk * 2) + k_slot_off);
    lx_buf_to_lds(rs_k, (lptr_t)(smem + ring_off + piece_lds + j * 4096), off_, ksoff);
  };
  auto vpiece = [&](int j, int ring_off, int vsoff) {
    lx_buf_to_lds(rs_v, (lptr_t)(smem + A4_VB + ring_off + piece_lds + j * 4096), v_off[j], vsoff);
  };

  // ---- wave-uniform KV-tile descriptors, handed down a three-deep FIFO: tile T (mask), T+1 (bias of its scores, V^T staging), T+2 (K staging).
  // The key segments an item's query segment attends to are packed into up to three "runs" when the generator reaches the item; the common
  // step is two scalar adds and a min, a run / item switch is a rare branch. ----
  struct Tile { int ksoff, vsoff, nvalid, nclamp, w; float bl; };   // K / V^T byte offsets, keys in the tile (0 = none), staging clamp (>= 1), item, bias * log2 e
  // Generator state is (item, run, keys left, current tile) only: a run / item switch recomputes what it needs from the item index (rare
  // path, ~100 scalar instructions), so that the common path carries no table around the loop (loop-carried scalars that a rare branch
  // redefines cost a dozen s_mov per frame at hipcc's block merges).
  int gw = blockIdx.x;                       // the item whose tiles are being handed out
  int g_run = -1, g_left = 0;
  Tile g_cur = {0, 0, 0, 1, 0, 0.f};
  auto gen_next = [&]() {
    if (__builtin_expect(g_left > 0, 1)) {
      g_cur.ksoff += KVBLK * ldk * 2; g_cur.vsoff += KVBLK * 2;
    } else {
      __builtin_amdgcn_sched_barrier(0);
      int sq_ = 0, rl0 = 0, rl1 = 0, rl2 = 0, r = 3;
      Item it = {0, 0, 0, 0, 0, 0, 0, 0};
      auto runs_of = [&](int w) {            // key segments the item's query segment attends to: length, or 0 (masked / absent)
        it = decode(w);
        sq_ = it.sq;
        rl0 = pick(sq_, b00, b10, b20) > -1e37f ? len0 : 0;
        rl1 = (n_seg > 1 && pick(sq_, b01, b11, b21) > -1e37f) ? len1 : 0;
        rl2 = (n_seg > 2 && pick(sq_, b02, b12, b22) > -1e37f) ? len2 : 0;
      };
      auto next_run = [&](int p) { return (p < 0 && rl0 > 0) ? 0 : ((p < 1 && rl1 > 0) ? 1 : ((p < 2 && rl2 > 0) ? 2 : 3)); };
      if (gw < n_items) { runs_of(gw); r = next_run(g_run); }
      if (r == 3) {                          // this item's keys are exhausted: on to the workgroup's next item
        gw = gw < n_items ? gw + n_cta : gw;
        if (gw >= n_items) { gw = n_items; g_cur.nvalid = 0; g_cur.nclamp = 1; g_cur.w = n_items; g_left = 0; return; }   // offsets stay on the last real tile
        runs_of(gw);
        r = next_run(-1);                    // (lx_attn_fwd: every query segment attends to at least one key segment)
      }
      g_run = r;
      g_left = pick(r, rl0, rl1, rl2);
      g_cur.ksoff = (pick(r, row00, row01, row02) + it.b * g_left) * ldk * 2 + it.h * (DH * 2);
      g_cur.vsoff = pick(r, vt00, vt01, vt02) * 2 + it.bh * DH * vt_ld * 2;
      g_cur.bl = pick(sq_, pick(r, b00, b01, b02), pick(r, b10, b11, b12), pick(r, b20, b21, b22)) * 1.4426950408889634f;
      g_cur.w = gw;
    }
    g_cur.nvalid = min(g_left, KVBLK);
    g_cur.nclamp = g_cur.nvalid;
    g_left -= g_cur.nvalid;
  };
  Item C = decode(gw);                       // the item being computed
  gen_next(); Tile T0 = g_cur;
  gen_next(); Tile T1 = g_cur;
  gen_next(); Tile T2 = g_cur;

  bf16x8 qf[2][8];             // Q fragments of the item (AGPRs)
  bf16x8 qn[2][8];             // the next item's, fetched under this item's last frame
  load_q(C, qn);

  f32x16 oacc[2][4];
  float lsum[2][2];            // LSUM 0: per lane, its own 32 probabilities per tile [query block][even / odd value]

  // ---- fragment addresses: K 16-B slot ((2ks + lhi) ^ (key & 15)), V^T slot ((2s + lhi) ^ ((d >> 1) & 7)); key block / d block are immediates,
  // the ring position is part of the register: both start at ring position 0 ----
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
  uint32_t kaddr[8], vaddr[4];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) kaddr[ks] = lds0 + l31 * 256 + (((2 * ks + lhi) ^ (l31 & 15)) * 16);
#pragma unroll
  for (int s = 0; s < 4; ++s) vaddr[s] = lds0 + A4_VB + l31 * 128 + (((2 * s + lhi) ^ ((l31 >> 1) & 7)) * 16);
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+v"(kaddr[ks]));
#pragma unroll
  for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(vaddr[s]));

  bf16x8 ring[A4_LOOK];
  u32x4 pw[2][4];              // [query block][slice]: the slice's P^T fragment (8 bf16 per lane)
  f32x16 sc[2][2];             // [query block][key block]
  float pv[2][2];              // exp2 results in flight: [unit parity][even / odd value]
  f32x16 offv;                 // MODE 2: bias * log2 e of the tile whose scores are computed next, in all 16 registers
  float off_cur = 0.f;

#define A4_SB() __builtin_amdgcn_sched_barrier(0)
  // LX_A4_ELIM_*: timing experiments only (WRONG numbers): what one class of instructions costs the stream -- DSR the fragment reads, DMA the
  // LDS-DMA pieces, VALU the softmax stream, BAR the frame barrier, ADDR the twelve ring-position adds (tools/run_a4_elim.sh)
#define A4_RC "v"
#ifdef LX_A4_ELIM_DSR
#define A4_DSR(dst, addr, offs) asm volatile("" : "+" A4_RC(dst) : "v"(addr)); A4_SB()
#elif defined(LX_A4_ELIM_DSR_HALF)
#define A4_DSR(dst, addr, offs) if constexpr (((offs) / 4096) & 1) asm volatile("" : "+" A4_RC(dst) : "v"(addr)); else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=" A4_RC(dst) : "v"(addr), "n"(offs)); A4_SB()
#else
#define A4_DSR(dst, addr, offs) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=" A4_RC(dst) : "v"(addr), "n"(offs)); A4_SB()
#endif
#define A4_WAITR(n, reg) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(reg) : "n"(n) : "memory"); A4_SB()
  // the fragment read for slot j (j >= 32: slots 0-3 of the next frame, P.V fragments)
#define A4_RD(j)                                                                                                       \
  {                                                                                                                    \
    constexpr int j_ = (j) & 31;                                                                                       \
    if constexpr (a4_kind(j_) == 0) { A4_DSR(ring[j_ % A4_LOOK], kaddr[a4_ks(j_)], a4_kb(j_) * 8192); }                \
    else { A4_DSR(ring[j_ % A4_LOOK], vaddr[a4_pvf(j_) >> 2], (a4_pvf(j_) & 3) * 4096); }                              \
  }
  // an MFMA statement, with the wait for its fragment in front of it in the SAME statement when WN >= 0 (an asm that DEFINES the ring
  // register right in front of the asm that reads it makes hipcc put an s_nop between them: "assume inline asm has dst forwarding hazard")
#ifdef LX_A4_ELIM_WAIT
#define A4_WN(WN) -1
#else
#define A4_WN(WN) (WN)
#endif
#define A4_ASMW(WN, TEXT, OUT, ...)                                                                                    \
  if constexpr (A4_WN(WN) >= 0) asm volatile("s_waitcnt lgkmcnt(%[w])\n\t" TEXT : OUT : __VA_ARGS__, [w] "n"((WN) < 0 ? 0 : (WN)) : "memory"); \
  else asm volatile(TEXT : OUT : __VA_ARGS__ : "memory")
#define A4_MMQ(ks, kb, qb, R, WN)                                                                                      \
  {                                                                                                                    \
    if constexpr ((ks) == 0) {                                                                                         \
      if constexpr (MODE == 2) { A4_ASMW(WN, "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3", "=&v"(sc[qb][kb]), A4_RC(ring[R]), "a"(qf[qb][0]), "v"(offv)); } \
      else { A4_ASMW(WN, "v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0", "=&v"(sc[qb][kb]), A4_RC(ring[R]), "a"(qf[qb][0])); } \
    } else { A4_ASMW(WN, "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0", "+v"(sc[qb][kb]), A4_RC(ring[R]), "a"(qf[qb][ks])); } \
    A4_SB();                                                                                                           \
  }
#define A4_MMP(f, qb, R, WN)                                                                                           \
  {                                                                                                                    \
    A4_ASMW(WN, "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0", "+a"(oacc[qb][(f) & 3]), A4_RC(ring[R]), "v"(pw[qb][(f) >> 2])); \
    A4_SB();                                                                                                           \
  }
#define A4_MM(i, qb, WN)                                                                                               \
  if constexpr (a4_kind(i) == 0) { A4_MMQ(a4_ks(i), a4_kb(i), qb, (i) % A4_LOOK, WN) } else { A4_MMP(a4_pvf(i), qb, (i) % A4_LOOK, WN) }
  // vector stream position n (0..159): slice s = n / 40; value j = 2 * pair + half of query block qb is score register 8 * (s & 1) + j of sc[qb][s >> 1]
#ifdef LX_A4_ELIM_EXP
#define A4_EXP_OP "v_mov_b32"
#else
#define A4_EXP_OP "v_exp_f32"
#endif
#ifdef LX_A4_ELIM_ADD
#define A4_ADD_OP(l, p) asm volatile("" : "+v"(l) : "v"(p));
#else
#define A4_ADD_OP(l, p) asm volatile("v_add_f32 %0, %0, %1" : "+v"(l) : "v"(p));
#endif
#ifdef LX_A4_ELIM_CVT
#define A4_CVT_OP(dst, e, o) asm volatile("" : "+v"(dst) : "v"(e), "v"(o))
#else
#define A4_CVT_OP(dst, e, o) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(dst) : "v"(e), "v"(o))
#endif
#define A4_OP(n)                                                                                                       \
  {                                                                                                                    \
    constexpr int s_ = (n) / A4_SL, m_ = (n) % A4_SL, u_ = a4_op_unit(m_), hf_ = a4_op_half(m_), qb_ = u_ & 1, pr_ = u_ >> 1; \
    if constexpr (a4_op_kind(m_) == 0) asm volatile(A4_EXP_OP " %0, %1" : "=v"(pv[u_ & 1][hf_]) : "v"(sc[qb_][s_ >> 1][8 * (s_ & 1) + 2 * pr_ + hf_])); \
    else if constexpr (a4_op_kind(m_) == 1) { A4_ADD_OP(lsum[qb_][hf_], pv[u_ & 1][hf_]) }                             \
    else A4_CVT_OP(pw[qb_][s_][pr_], pv[u_ & 1][0], pv[u_ & 1][1]);                                                    \
  }
#define A4_OPK(g, k) if constexpr ((k) < a4_ngap(g)) A4_OP((a4_pos(g) + (k)) % (4 * A4_SL))
#ifdef LX_A4_ELIM_VALU
#define A4_VALU(g)
#else
#define A4_VALU(g) { A4_OPK(g, 0) A4_OPK(g, 1) A4_OPK(g, 2) A4_OPK(g, 3) } A4_SB();
#endif
  // One slot. DRAIN: the matrix work of slots 0-5 only (the last tile's P.V slices 2b, 3 behind the loop)
#ifdef LX_A4_ELIM_BAR
#define A4_FRAME_BARRIER()
#else
#define A4_FRAME_BARRIER() __builtin_amdgcn_s_barrier()
#endif
#ifdef LX_A4_ELIM_DMA
#define A4_KPIECE(j, pos, krow, nclamp)
#define A4_VPIECE(j, pos, vpos)
#else
#define A4_KPIECE(j, pos, krow, nclamp) kpiece(j, pos, krow, nclamp)
#define A4_VPIECE(j, pos, vpos) vpiece(j, pos, vpos)
#endif
#ifdef LX_A4_ELIM_ADDR
#define A4_ADDR_STEP(reg, d)
#else
#define A4_ADDR_STEP(reg, d) reg += d; asm volatile("" : "+v"(reg))
#endif
#define A4_SLOT(i, DRAIN)                                                                                              \
  if constexpr (!(DRAIN)) {                                                                                            \
    if constexpr ((i) == 8) {       /* pieces of the previous frame landed (this wave, then every wave); every wave is done with K(T-1), V^T(T-2) */ \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); A4_SB();                                                        \
      A4_FRAME_BARRIER(); A4_SB();                                                                                     \
    }                                                                                                                  \
    if constexpr ((i) == 16) {      /* ragged last tile of a segment, key block 1 (complete since slot 13): keys past the end get p = exp2(-1e30) = 0 */ \
      if (T0.nvalid < KVBLK) {                                                                                         \
        A4_SB();                                                                                                       \
        int lh4_ = 4 * lhi;                                                                                            \
        LX_PIN_IN_BRANCH(lh4_);                                                                                        \
        _Pragma("unroll") for (int qb = 0; qb < 2; ++qb) _Pragma("unroll") for (int r = 0; r < 16; ++r)                \
          if (lh4_ + 32 + 8 * (r >> 2) + (r & 3) >= T0.nvalid) sc[qb][1][r] = -1e30f;                                  \
      }                                                                                                                \
      A4_SB();                                                                                                         \
    }                                                                                                                  \
    if constexpr ((i) == 20 && MODE == 2) {      /* slot 22 starts the scores of tile T+1 from ITS bias */              \
      if (T1.bl != off_cur) {                                                                                          \
        A4_SB();                                                                                                       \
        off_cur = T1.bl;                                                                                               \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) offv[r] = off_cur;                                              \
        asm volatile("" : "+v"(offv));                                                                                 \
      }                                                                                                                \
      A4_SB();                                                                                                         \
    }                                                                                                                  \
  }                                                                                                                    \
  A4_MM(i, 0, (DRAIN) ? a4_wait(i, A4_LOOK == 4 ? 6 : 8) : a4_wait((i) & 1, 64))     /* (steady state: LOOK reads outstanding at every slot; a4_wait(0 / 1, .) by parity) */ \
  if constexpr (!(DRAIN)) { A4_VALU(2 * (i)) }                                                                         \
  A4_MM(i, 1, -1)                                                                                                      \
  if constexpr (!(DRAIN) || (i) + A4_LOOK < 6) { A4_RD((i) + A4_LOOK) }                                                \
  if constexpr (!(DRAIN)) {                                                                                            \
    A4_VALU(2 * (i) + 1)                                                                                               \
    if constexpr ((i) >= 2 && (i) < 6) { A4_ADDR_STEP(vaddr[((i) - 2) & 3], dv); A4_SB(); }     /* V^T reads move on to ring position T (the reads of slots <= 5 are issued by slot 1) */ \
    if constexpr ((i) >= 10 && (i) < 18) { A4_ADDR_STEP(kaddr[((i) - 10) & 7], dk); A4_SB(); } /* K reads move on to position T+1 (the kb-1 reads of K(T) are issued by slot 9) */ \
    if constexpr (a4_kpiece(i) >= 0) { A4_KPIECE(a4_kpiece(i) & 3, pos_prev, T2.ksoff, T2.nclamp); A4_SB(); }             \
    if constexpr (a4_vpiece(i) >= 0) { A4_VPIECE(a4_vpiece(i) & 3, pos_next, T1.vsoff); A4_SB(); }                        \
    if constexpr ((i) == 12) {      /* the item's last frame: the next item's Q (its tiles are being staged already); behind slot 8's vmcnt(0) */ \
      if (T1.w != C.w && T1.nvalid != 0) { A4_SB(); const Item nx_ = decode(T1.w); load_q(nx_, qn); }                     \
      A4_SB();                                                                                                         \
    }                                                                                                                  \
    if constexpr ((i) == 27) { T0 = T1; T1 = T2; A4_SB(); }     /* (nothing reads T1 / T2 behind the V^T pieces: the scalar bookkeeping sits under queued MFMAs) */ \
    if constexpr ((i) == 28) { gen_next(); T2 = g_cur; A4_SB(); }                                                      \
  }
#define A4_SLOT4(i, DRAIN) A4_SLOT(i, DRAIN) A4_SLOT((i) + 1, DRAIN) A4_SLOT((i) + 2, DRAIN) A4_SLOT((i) + 3, DRAIN)

  // ---- prologue: K(0) -> ring position 0, V^T(0) -> 0, K(1) -> 1 ----
#pragma unroll
  for (int j = 0; j < 4; ++j) kpiece(j, 0, T0.ksoff, T0.nclamp);
#pragma unroll
  for (int j = 0; j < 4; ++j) vpiece(j, 0, T0.vsoff);
#pragma unroll
  for (int j = 0; j < 4; ++j) kpiece(j, A4_KV, T1.ksoff, T1.nclamp);
  A4_SB();
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // Q and K(0) of this wave (V^T(0) and K(1) land under the first tile's kb-0 scores) ...
  A4_SB();
  __builtin_amdgcn_s_barrier();                          // ... and of every wave
  A4_SB();
  int pos_cur = 0;                                       // byte offset of ring position T % 3
  bool first_frame = true;                               // the V^T reads of the very first frame stay on position 0 (V^T(0) itself stands in for "tile -1")
  while (true) {                                         // ---- items ----
    // the item's Q into the AGPRs (the loads are complete: the prologue's wait / the wait behind the previous item's drain)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) { qf[qb][ks] = qn[qb][ks]; asm volatile("" : "+a"(qf[qb][ks])); }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[qb][i][r] = 0.f;
      lsum[qb][0] = lsum[qb][1] = 0.f;
      // "tile -1" of the item: P = 0 against whatever finite tile the V^T reads still point at (0 x finite = 0)
      pw[qb][2] = u32x4{0, 0, 0, 0}; pw[qb][3] = u32x4{0, 0, 0, 0};
#ifdef LX_A4_ELIM_VALU
      pw[qb][0] = u32x4{1, 1, 1, 1}; pw[qb][1] = u32x4{1, 1, 1, 1}; pv[qb][0] = pv[qb][1] = 0.f;
#endif
    }
    if constexpr (MODE == 2) {
      off_cur = T0.bl;
#pragma unroll
      for (int r = 0; r < 16; ++r) offv[r] = off_cur;
      asm volatile("" : "+v"(offv));
    }
    A4_SB();
    // kb-0 scores of the item's first tile (K reads point at its ring position: the tile stream is continuous across items)
#define A4_PRD(ks) A4_DSR(ring[(ks) % A4_LOOK], kaddr[ks], 0)
#define A4_PG(ks)                                                                                                      \
  A4_MMQ(ks, 0, 0, (ks) % A4_LOOK, a4_wait(ks, 8)) A4_MMQ(ks, 0, 1, (ks) % A4_LOOK, -1)                                \
  if constexpr ((ks) + A4_LOOK < 8) { A4_PRD(((ks) + A4_LOOK) & 7); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    A4_SB();
    A4_PRD(0); A4_PRD(1); A4_PRD(2); A4_PRD(3);
    if constexpr (A4_LOOK == 8) { A4_PRD(4); A4_PRD(5); A4_PRD(6); A4_PRD(7); }
    A4_PG(0) A4_PG(1) A4_PG(2) A4_PG(3) A4_PG(4) A4_PG(5) A4_PG(6) A4_PG(7)
#undef A4_PG
#undef A4_PRD
    A4_SB();
    // the first item: V^T(0) and K(1) of this wave, then of every wave (later items: nothing is outstanding here, the tile stream's own
    // waits and barriers have covered their first tiles); the kb-0 scores -> their first vector read: 11 wait states, with room
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    A4_SB();
    __builtin_amdgcn_s_barrier();
    A4_SB();
    asm volatile("s_nop 7" ::: "memory");
    A4_SB();
    A4_RD(32) A4_RD(33) A4_RD(34) A4_RD(35)              // the ring: the first slots of the item's first frame (0-5: "tile -1", P = 0)
    if constexpr (A4_LOOK == 8) { A4_RD(36) A4_RD(37) A4_RD(38) A4_RD(39) }
    while (true) {                                       // ---- frames ----
      const int pos_next = pos_cur == 2 * A4_KV ? 0 : pos_cur + A4_KV;      // position of T+1: V^T(T+1) is staged there, the K reads move there
      const int pos_prev = pos_cur == 0 ? 2 * A4_KV : pos_cur - A4_KV;      // position of T-1 = T+2: K(T+2) is staged there, the V^T reads come from there
      const int dk = pos_next - pos_cur, dv = first_frame ? 0 : pos_cur - pos_prev;
      if (T0.nvalid < KVBLK) {        // ragged last tile of a segment, key block 0 (complete since slot 29 of the previous frame)
        A4_SB();
        int lh4_ = 4 * lhi;
        LX_PIN_IN_BRANCH(lh4_);
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (lh4_ + 8 * (r >> 2) + (r & 3) >= T0.nvalid) sc[qb][0][r] = -1e30f;
      }
      A4_SB();
      A4_SLOT4(0, false) A4_SLOT4(4, false) A4_SLOT4(8, false) A4_SLOT4(12, false)
      A4_SLOT4(16, false) A4_SLOT4(20, false) A4_SLOT4(24, false) A4_SLOT4(28, false)
      pos_cur = pos_next;
      first_frame = false;
      A4_SB();
      if (T0.w != C.w) break;         // (T0 is the NEXT tile by now: the FIFO moves behind slot 27)
    }
    // ---- drain: P.V slice 2 (d blocks 2, 3) and slice 3 of the item's last tile ----
    {
      const int dv = 0, dk = 0, pos_prev = 0, pos_next = 0;
      (void)dv; (void)dk; (void)pos_prev; (void)pos_next;
      A4_SLOT4(0, true) A4_SLOT(4, true) A4_SLOT(5, true)
    }
    // MFMA results in AGPRs -> v_accvgpr_read: 18 wait states by hand; the ring reads issued for a frame that does not follow have landed
    asm volatile("s_nop 15\n s_nop 7\n s_waitcnt lgkmcnt(0)" ::: "memory");
    A4_SB();
    const bool more = T0.nvalid != 0;
    // ---- epilogue: O[q, d] = O^T / l ----
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const float l_lane = lsum[qb][0] + lsum[qb][1];
      const float l_tot = l_lane + __shfl_xor(l_lane, 32, 64);
      const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
      const int q_in_seg = C.q_tile0 + wave * 64 + l31 + 32 * qb;
      const size_t q_row = (size_t)C.q_row0 + min(q_in_seg, C.q_len - 1);
      lx_store_o(o_mode, o_ovf, q_in_seg < C.q_len, (uint16_t*)D.O + q_row * D.ldo + D.o_col + C.h * DH, oacc[qb], inv, lhi);
    }
    if (!more) break;
    C = decode(T0.w);
    A4_SB();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the next item's Q (fetched under the last frame); hipcc's own wait in front of the copy counts the same
    A4_SB();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every LDS-DMA piece of this wave (tiles that do not exist are staged too) has landed before the wave ends
#undef A4_SLOT4
#undef A4_FRAME_BARRIER
#undef A4_KPIECE
#undef A4_VPIECE
#undef A4_ADDR_STEP
#undef A4_SLOT
#undef A4_VALU
#undef A4_OPK
#undef A4_OP
#undef A4_EXP_OP
#undef A4_ADD_OP
#undef A4_CVT_OP
#undef A4_MM
#undef A4_ASMW
#undef A4_MMP
#undef A4_MMQ
#undef A4_RD
#undef A4_WAITR
#undef A4_DSR
#undef A4_SB
#endif
}

}  // namespace

// mode 1: no bias on any attended pair; 2: biases. The caller (lx_attn_fwd) has validated the descriptor and decided that the
// bounded-score contract holds and that every byte offset fits 31 bits.
int lx_attn4_cus(void) {
  static const int n_cu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  return n_cu;
}

int lx_attn4_launch(const void* attn_args, int n_items, int mode, void* stream) {
  const AttnArgs& a = *(const AttnArgs*)attn_args;
  // one workgroup per CU (96 KiB of LDS, ~400 registers per lane): more items than CUs -> a persistent launch
  const int grid = n_items > lx_attn4_cus() ? lx_attn4_cus() : n_items;
  if (mode == 2) hipLaunchKernelGGL((lx_attn4_kernel<2>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a, n_items);
  else hipLaunchKernelGGL((lx_attn4_kernel<1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a, n_items);
  return 0;
}
