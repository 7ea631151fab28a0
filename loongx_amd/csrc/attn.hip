// attn.hip -- joint (text | image | condition) flash attention forward for gfx950, head_dim 128, bf16.
//
// Replaces F.scaled_dot_product_attention and the mask / c_factor bias construction of the reference's
// attn_forward (src/flux/block.py:101-135): non-causal softmax(Q K^T / sqrt(dh) + bias) V over the
// concatenation of up to three token segments, with the bias constant per (query segment, key segment)
// pair (0, log(c_factor), or -inf for the union_cond_attn=False / independent_condition masks), so the
// [S,S] mask tensor is never materialised.
//
// Structure (one workgroup = NW waves = 32*NW query rows of one (batch, head); lane = query row; NW = 8 by default: one
// K/V tile staged per 256 query rows halves the LDS-DMA pieces per flop against NW = 4):
//   S^T[key, q] = K_tile . Q^T      v_mfma_f32_32x32x16_bf16, K fragments from LDS, Q resident in VGPRs
//   online softmax in registers     each lane owns 32 scores of ITS query row; the other 32 sit in lane^32
//   O^T[d, q]  += V^T_tile . P^T    P fragments are 8 consecutive accumulator registers (no shuffles):
//                                   the key order inside every 16-key group is interleaved as
//                                   [0-3, 8-11, 4-7, 12-15] on BOTH operands, which lx_qkv_prep bakes
//                                   into the V^T image it writes, so V^T fragments are plain ds_read_b128.
//   K tile [64 keys][128] and V^T tile [128 d][64 keys] are staged with global_load_lds (16 B/lane),
//   double buffered, one barrier per tile; 16-B slots are XOR-swizzled on the source address
//   (K: slot ^= key&15, V^T: slot ^= (d>>1)&7) so fragment reads are bank-conflict free.
#include "attn_common.h"
#include <algorithm>

#ifdef LX_ATTN_PROBE
// Measurement build only (tools/attn_probe.py through tools/build_variant.sh -DLX_ATTN_PROBE=1): per wave of lx_attn_pipe_kernel,
// shader-clock cycles spent in the end-of-iteration `s_waitcnt vmcnt(0)` (this wave's LDS-DMA pieces) and `s_barrier` (the other
// waves), in the iterations as a whole, and in the kernel from entry to exit. s_memtime is a scalar memory read: it is taken only
// where the iteration has just waited lgkmcnt down to 0, and costs a round trip each time -- the SPLIT is what the numbers are for.
__device__ unsigned long long lx_attn_probe_buf[4096 * 8 * 4];
extern "C" int lx_attn_probe_read(unsigned long long* host, size_t n_u64) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(lx_attn_probe_buf), n_u64 * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif

namespace {

// LX_ATTN_DEFER: rescale threshold in log2 units for the deferred running-max update (0 = always rescale).
// While no row of the wave sees its tile maximum grow past the max in use by more than this, the stale max keeps being
// used (P <= 2^thr), so the 64 accumulator multiplies and the l rescale are skipped; O and l carry the same factor, the
// final O / l is unchanged.
constexpr float DEFER_THR = 8.0f;

// (The plain 8-wave kernel this file started with -- both waves of a SIMD alternating QK -> softmax -> PV, 695-710 TFLOP/s -- lived here until
//  round 5 as an A/B arm (LX_ATTN_PIPE=0); the pipelined kernel below has been the only default since round 1: DESIGN 3.3.)


// ---------------------------------------------------------------------------------------------------------------
// Software-pipelined variant (default): one instruction stream per wave that interleaves matrix and vector work.
//
// Measured on gfx950 (tools/ubench/coexec): an MFMA stream on one wave and a VALU stream on the wave sharing its SIMD do
// NOT overlap -- their times add (651 + 1405 -> 1921 us), whatever s_setprio says. In lx_attn_kernel both waves of a
// SIMD alternate QK -> softmax -> PV, so a KV tile costs (2 x 1024 MFMA + 2 x ~1100 VALU) cycles per SIMD: 5300 measured,
// MfmaUtil 43 %. Skewing the two waves by a phase (tried: X = {PV, QK} / Y = {softmax} with an odd s_barrier) changes
// nothing for the same reason. What does overlap is a wave's OWN vector instructions issued in the shadow of its own
// MFMAs. So every iteration t runs 32 "gaps", each = {wait for the operand, one MFMA, the ds_read for six gaps ahead,
// a few VALU instructions}:
//     matrix work : QK(t+1) (scores of the NEXT tile)  and  PV(t)
//     vector work : softmax(t): row max (before gap 0, covering the first LDS round trip), then 32 one-value half-units
//                   {fma, exp2, row-sum[, cvt_pk]}, about one per gap, placed so that P slice s (16 keys) is complete
//                   just before PV slice s.
// Scores live in two register sets that swap roles every iteration (the loop body is expanded twice). The last
// iteration's QK works on a stale buffer and is discarded. One barrier per tile; K(t+2) and V(t+1) are staged (four 1-KiB
// LDS-DMA pieces per wave, spread over gaps) into the slots K(t) / V(t-1) vacated in the previous iteration.
// Gap schedule of one iteration (32 gaps): the row max / rescale decision of tile t occupies the vector slots of gaps 0-3, the
// 32 softmax half-units gaps 4-27 (slice s = halves 8s..8s+7), and PV slice s starts only after its slice is complete.
//   MFMAs : gaps 0-13 QK fragments 0-13 | 14-17 PV slice 0 | 18-19 QK 14-15 | 20-23 PV 1 | 24-27 PV 2 | 28-31 PV 3
constexpr int pipe_is_q(int g) { return g < 14 || g == 18 || g == 19; }
constexpr int pipe_idx(int g) {   // index of the QK fragment (ks*2+kb) or PV fragment (s*4+db) consumed at gap g
  return g < 14 ? g : g < 18 ? g - 14 : g < 20 ? g - 4 : g - 16;
}
constexpr int pipe_half(int g, int k) {   // k-th (0/1) softmax half-unit (slice*8 + value) issued behind gap g, or -1
  constexpr int at[32] = {4, 5, 6, 7, 8, 9, 10, 11, 10, 11, 12, 13, 14, 15, 16, 17, 16, 17, 18, 19, 20, 21, 22, 23, 20, 21, 22, 23, 24, 25, 26, 27};
  int seen = 0;
  for (int j = 0; j < 32; ++j)
    if (at[j] == g) { if (seen == k) return j; ++seen; }
  return -1;
}

// MODE 0: online softmax with a running maximum (the general contract). MODE 1 / 2 (LX_ATTN_Q_LOG2 | LX_ATTN_BOUNDED, include/lx.h):
// q carries scale * log2 e and the caller bounds every score, so there is no running maximum at all -- the row-max chunks of gaps 0-3
// and the per-score v_fma go: p = exp2(s) (MODE 1: every unmasked segment pair has bias 0) or exp2(s + bias) (MODE 2). Measured with
// elimination builds (profiles/r03s_attn_elim_*.txt): -5.8 % at S = 2560, -6.9 % at S = 8704.
template <bool DEFER, int MODE>
__global__ __launch_bounds__(512, 1) void lx_attn_pipe_kernel(const AttnArgs args) {
  constexpr int NW = 8, QBLK = 256;
  __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE_BYTES];
  const lx_attn_desc& D = args.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;

  const int BH = D.B * D.H;
  int qt, bh;
  lx_item_decode((int)blockIdx.x, (int)gridDim.x, BH, args.qt_start[3], qt, bh);
  const int b = bh / D.H, h = bh % D.H;
  int sq = 0;
#pragma unroll
  for (int s = 1; s < 3; ++s)
    if (s < D.n_seg && qt >= args.qt_start[s]) sq = s;
  // Descriptor fields are copied into scalars ONCE: indexing the kernarg arrays with the running segment index inside
  // the tile loop costs a dependent s_load (~150 cycles each) and scalar loads would break the counted lgkmcnt below.
  const int n_seg = D.n_seg;
  const int len0 = D.seg_len[0], len1 = D.seg_len[1], len2 = D.seg_len[2];
  const int row00 = D.seg_row0[0], row01 = D.seg_row0[1], row02 = D.seg_row0[2];
  const int vt00 = D.seg_vt0[0], vt01 = D.seg_vt0[1], vt02 = D.seg_vt0[2];
  const float bia0 = D.bias[sq][0], bia1 = D.bias[sq][1], bia2 = D.bias[sq][2];
  auto pick = [](int s, auto x0, auto x1, auto x2) { return s == 0 ? x0 : (s == 1 ? x1 : x2); };
  auto seg_len = [&](int s) { return pick(s, len0, len1, len2); };

  const int q_len = seg_len(sq);
  const int q_in_seg = (qt - args.qt_start[sq]) * QBLK + wave * 32 + l31;
  const bool q_valid = q_in_seg < q_len;
  const size_t q_row = (size_t)pick(sq, row00, row01, row02) + (size_t)b * q_len + min(q_in_seg, q_len - 1);

  bf16x8 qf[8];
  {
    const __bf16* qp = (const __bf16*)D.Q + q_row * D.ldq + D.q_col + h * DH + lhi * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
  }
  const float c2 = (D.flags & LX_ATTN_Q_LOG2) ? 1.0f : D.scale * 1.4426950408889634f;

  f32x16 oacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;   // (LX_ATTN_LSUM_MFMA: l_run is unused, the row sums live in lacc)

  const __bf16* Kbase = (const __bf16*)D.K + D.k_col + h * DH;
  const __bf16* Vbase = (const __bf16*)D.VT + (size_t)bh * DH * D.vt_ld;
  const int ldk = D.ldk, vt_ld = D.vt_ld;
  // One 1-KiB LDS-DMA piece: j = 0,1 K rows, j = 2,3 V^T rows (each wave moves four pieces per tile). Branch-free, so
  // that the pieces can sit inside the MFMA stream without splitting it into basic blocks: source = wave-uniform tile
  // origin (SGPR pair) + per-lane 32-bit byte offset. K rows are clamped to the tile's last valid key (ragged segment
  // tails; a tile that does not exist is staged from row 0 of the previous one and never read).
  const int k_key = wave * 4 + (lane >> 4);                                              // + 32 for the second piece
  const uint32_t k_slot_off = (uint32_t)((((lane & 15) ^ (k_key & 15)) * 8) * 2);
  const uint32_t v_lane_off = (uint32_t)(((wave * 8 + (lane >> 3)) * vt_ld + (((lane & 7) ^ (((wave * 8 + (lane >> 3)) >> 1) & 7)) * 8)) * 2);
  // (buffer addressing: the SRSRC is rebuilt from the wave-uniform origin with scalar instructions; in the GEMM loop this form
  // measured 260 fewer stall cycles per 16 pieces than flat-global addresses)
  uint32_t k_off0 = (uint32_t)(k_key * ldk * 2) + k_slot_off, k_off1 = (uint32_t)((k_key + 32) * ldk * 2) + k_slot_off;
  asm volatile("" : "+v"(k_off0), "+v"(k_off1));     // (kept in registers: rematerialised they are the multiply again)
  auto piece = [&](int j, int krow, int vpos, int nclamp, int slot) {
    char* base = smem + slot * STAGE_BYTES;
    if (j < 2) {
      const lx_rsrc_t rs = lx_make_rsrc(Kbase + (size_t)krow * ldk);                                     // wave-uniform
      // min(row, clamp) * ld + slot = min(row * ld + slot, clamp * ld + slot): the products are loop-invariant registers / one scalar
      // multiply, so a piece costs v_add + v_min instead of v_min + v_mad_u64_u32 (hipcc's only 32 x 32 + 32 form, not full rate)
      const uint32_t off_ = min(j ? k_off1 : k_off0, (uint32_t)((nclamp - 1) * ldk * 2) + k_slot_off);
      lx_buf_to_lds(rs, (lptr_t)(base + (j * NW + wave) * 1024), off_, 0);
    } else {
      const int jj = j - 2;
      const lx_rsrc_t rs = lx_make_rsrc(Vbase + (size_t)(jj * NW * 8) * vt_ld + vpos);                   // wave-uniform
      lx_buf_to_lds(rs, (lptr_t)(base + K_BYTES + (jj * NW + wave) * 1024), v_lane_off, 0);
    }
  };
  // Wave-uniform KV-tile descriptors. `gen` walks (segment, tile) over the segments this query segment may attend to; the
  // common step is three scalar adds, a segment switch is a rare branch. Each iteration needs the descriptors of tiles t
  // (softmax: bias, valid keys), t+1 (V staging) and t+2 (K staging): they are handed down a three-deep FIFO.
  struct Tile { int krow, vpos, nvalid, nclamp; float bl; };   // first key row, first V^T column, keys in the tile (0 = none),
                                                               // row clamp for staging (>= 1 always), bias*log2e
  int g_seg = -1, g_left = 0;                               // generator state: segment, keys of it not yet handed out
  Tile g_cur = {0, 0, 0, 1, 0.f};
  auto gen_next = [&]() {
    if (g_left > 0) {
      g_cur.krow += KVBLK; g_cur.vpos += KVBLK;
    } else {
      __builtin_amdgcn_sched_barrier(0);
      do { ++g_seg; } while (g_seg < n_seg && !(pick(g_seg, bia0, bia1, bia2) > -1e37f));
      if (g_seg >= n_seg) { g_cur.nvalid = 0; g_cur.nclamp = 1; g_left = 0; return; }   // krow / vpos stay on the last real tile
      g_left = seg_len(g_seg);
      g_cur.krow = pick(g_seg, row00, row01, row02) + b * g_left;
      g_cur.vpos = pick(g_seg, vt00, vt01, vt02);
      g_cur.bl = pick(g_seg, bia0, bia1, bia2) * 1.4426950408889634f;
    }
    g_cur.nvalid = min(g_left, KVBLK);
    g_cur.nclamp = g_cur.nvalid;
    g_left -= g_cur.nvalid;
  };
  gen_next();
  Tile t0 = g_cur;                  // tile t: being soft-maxed and multiplied into O
  gen_next();
  Tile t1 = g_cur;                  // tile t+1: its scores are being computed, its V staged
  gen_next();
  Tile t2 = g_cur;                  // tile t+2: its K is being staged

  // K: 16-B slot ((2ks + lhi) ^ (key & 15)), V^T: slot ((2s + lhi) ^ ((d >> 1) & 7)). The tiles are 1 KiB aligned and
  // 2ks / 2s only touch bits the row offset leaves clear, so address(ks) = address(0) ^ (ks * 32): two address registers.
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
  const uint32_t kaddr0 = lds0 + l31 * 256 + ((lhi ^ (l31 & 15)) * 16);
  const uint32_t vaddr0 = lds0 + K_BYTES + l31 * 128 + ((lhi ^ ((l31 >> 1) & 7)) * 16);

#ifndef LX_ATTN_LOOK
#define LX_ATTN_LOOK 5
#endif
#ifndef LX_ATTN_PG0                 // gaps behind which the four LDS-DMA pieces of an iteration are issued (A/B knobs; tools/attn_ab.py)
#define LX_ATTN_PG0 1
#define LX_ATTN_PG1 3
#define LX_ATTN_PG2 5
#define LX_ATTN_PG3 7
#endif
  constexpr int LOOK = LX_ATTN_LOOK;           // fragments in flight ahead of the MFMA that consumes them (<= 6)
  bf16x8 ring[LOOK];
#ifdef LX_ATTN_ELIM_SOFT
  u32x4 pfw[4] = {{1, 1, 1, 1}, {1, 1, 1, 1}, {1, 1, 1, 1}, {1, 1, 1, 1}};
#else
  u32x4 pfw[4];
#endif
  f32x16 sA[2], sB[2];
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float mx[4], t_new = 0.f;
  float off = 0.f, p_even[2] = {0.f, 0.f};   // (slices s and s+1 overlap in time: one pending even value per slice parity)
  uint32_t bq = 0, bv = 0;          // LDS byte offsets of the K buffer read by QK and the V buffer read by PV

#define LX_FENCE() asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0)
#define LX_BARRIER() LX_FENCE(); __builtin_amdgcn_s_barrier(); LX_FENCE()
  // LX_ATTN_ELIM_*: timing experiments only (WRONG numbers): what one class of instructions costs the stream -- DSR the fragment reads,
  // DMA the LDS-DMA pieces, EXP the exp2, SOFT the whole softmax half-units, BAR the end-of-iteration barrier (tools/run_r03u.sh)
#ifdef LX_ATTN_ELIM_DSR
#define LX_DSR(dst, addr, offs) asm volatile("" : "+v"(dst) : "v"(addr))
#else
#define LX_DSR(dst, addr, offs) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(offs))
#endif
#define LX_WAITL(n) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory"); __builtin_amdgcn_sched_barrier(0)
  // read of the fragment consumed at gap j (gaps >= last do not exist); QONLY: prologue stream of the 16 K fragments
#define LX_RD(j, last, QONLY)                                                                                          \
  if ((j) < (last)) {                                                                                                  \
    if ((QONLY) || pipe_is_q(j)) {                                                                                     \
      constexpr int f_ = (QONLY) ? (j) : pipe_idx(j);                                                                  \
      const uint32_t a_ = (kaddr0 + bq) ^ ((f_ >> 1) * 32);                                                            \
      LX_DSR(ring[(j) % LOOK], a_, (f_ & 1) * 8192);                                                                   \
    } else {                                                                                                           \
      constexpr int f_ = pipe_idx(j);                                                                                  \
      const uint32_t a_ = (vaddr0 + bv) ^ ((f_ >> 2) * 32);                                                            \
      LX_DSR(ring[(j) % LOOK], a_, (f_ & 3) * 4096);                                                                   \
    }                                                                                                                  \
  }                                                                                                                    \
  __builtin_amdgcn_sched_barrier(0)
#define LX_RDP(j, last, QONLY) LX_RD(j, (j) < LOOK ? (last) : 0, QONLY)   /* the LOOK reads that prime the ring */
#define LX_LSUM(f_)
#define LX_LADD(p_) l_run += p_;
#define LX_LPIN , "+v"(l_run)
#define LX_MM(g, SC, SN, QONLY)                                                                                        \
  if ((QONLY) || pipe_is_q(g)) {                                                                                       \
    constexpr int f_ = (QONLY) ? (g) : pipe_idx(g);                                                                    \
    SN[f_ & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[(g) % LOOK], qf[f_ >> 1], f_ < 2 ? zero16 : SN[f_ & 1], 0, 0, 0); \
  } else {                                                                                                             \
    constexpr int f_ = pipe_idx(g);                                                                                    \
    oacc[f_ & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[(g) % LOOK], __builtin_bit_cast(bf16x8, pfw[f_ >> 2]), oacc[f_ & 3], 0, 0, 0); \
    LX_LSUM(f_)                                                                                                        \
  }                                                                                                                    \
  __builtin_amdgcn_sched_barrier(0)
  // softmax half-unit hu = slice*8 + value: one score -> one probability and the row sum; odd values also pack the bf16 pair.
  // The empty asm keeps it in ITS gap (hipcc sinks the arithmetic to the first use otherwise).
#ifdef LX_ATTN_ELIM_EXP
#define LX_EXP2(x) (x)
#else
#define LX_EXP2(x) __builtin_amdgcn_exp2f(x)
#endif
#define LX_SCORE_ARG(x) (MODE == 0 ? fmaf(x, c2, off) : MODE == 1 ? (x) : (x) + off)
#define LX_HALF(hu, SC)                                                                                                \
  {                                                                                                                    \
    constexpr int s_ = (hu) >> 3, r_ = 8 * (s_ & 1) + ((hu) & 7);                                                      \
    const float p_ = LX_EXP2(LX_SCORE_ARG(SC[s_ >> 1][r_]));                                                           \
    LX_LADD(p_)                                                                                                        \
    if ((hu) & 1) { pfw[s_][((hu) & 7) >> 1] = pack_bf16x2(p_even[s_ & 1], p_); asm volatile("" : "+v"(pfw[s_][((hu) & 7) >> 1]) LX_LPIN); } \
    else { p_even[s_ & 1] = p_; asm volatile("" : "+v"(p_even[s_ & 1]) LX_LPIN); }                                     \
  }
  // row max of tile t and the rescale decision, as vector fillers of gaps 0-3 (they used to run serially at the head of the
  // iteration: 8 us of a 90 us launch with nothing to overlap); four independent v_max3 chains, not one 16-deep chain
#define LX_LSCALE(a_) l_run *= a_
#define LX_MCHUNK(g, SC)                                                                                               \
  if ((g) == 0) {                                                                                                      \
    _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                                    \
      const int kb = c >> 1, r0 = (c & 1) * 8;                                                                         \
      mx[c] = __builtin_fmaxf(__builtin_fmaxf(SC[kb][r0], SC[kb][r0 + 1]), SC[kb][r0 + 2]);                            \
      mx[c] = __builtin_fmaxf(__builtin_fmaxf(mx[c], SC[kb][r0 + 3]), SC[kb][r0 + 4]);                                 \
    }                                                                                                                  \
    asm volatile("" : "+v"(mx[0]), "+v"(mx[1]), "+v"(mx[2]), "+v"(mx[3]));                                             \
  } else if ((g) == 1) {                                                                                               \
    _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                                    \
      const int kb = c >> 1, r0 = (c & 1) * 8;                                                                         \
      mx[c] = __builtin_fmaxf(__builtin_fmaxf(mx[c], SC[kb][r0 + 5]), SC[kb][r0 + 6]);                                 \
      mx[c] = __builtin_fmaxf(mx[c], SC[kb][r0 + 7]);                                                                  \
    }                                                                                                                  \
    mx[0] = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(mx[0], mx[1]), mx[2]), mx[3]);                             \
    asm volatile("" : "+v"(mx[0]));                                                                                    \
  } else if ((g) == 2) {                                                                                               \
    const auto sw_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx[0]), __float_as_uint(mx[0]), false, false);   \
    t_new = fmaxf(__uint_as_float(sw_[0]), __uint_as_float(sw_[1])) * c2 + bl;                                         \
    asm volatile("" : "+v"(t_new));                                                                                    \
  } else if ((g) == 3) {                                                                                               \
    bool rescale = true;                                                                                               \
    if (DEFER) rescale = __builtin_amdgcn_ballot_w64(t_new - m_run > DEFER_THR) != 0;                                  \
    if (rescale) {                                                                                                     \
      const float m_new = fmaxf(m_run, t_new);                                                                         \
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);                                                       \
      LX_LSCALE(alpha);                                                                                                \
      m_run = m_new;                                                                                                   \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha; \
    }                                                                                                                  \
    off = bl - m_run;                                                                                                  \
    asm volatile("" : "+v"(off));                                                                                      \
  }
  // the wait of gap g hands the fragment register on ("+v"): the MFMA that consumes it then depends on the WAIT, not only on the
  // ds_read that was issued five gaps earlier -- otherwise nothing but luck keeps hipcc from hoisting the MFMA above its wait
// (Wave-priority variants -- a static s_setprio for the younger half, priority traded inside every iteration -- were measured in round 3,
// profiles/r03m_*: zero-sum between the two waves of a SIMD; removed in round 5.)
#define LX_WAITR(n, reg) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(reg) : "n"(n) : "memory"); __builtin_amdgcn_sched_barrier(0)
#define LX_MCHUNK_MAYBE(g, SC) if constexpr (MODE == 0) { LX_MCHUNK(g, SC) }
#define LX_WAIT_GAP(g) LX_WAITR((31 - (g)) < (LOOK - 1) ? (31 - (g)) : (LOOK - 1), ring[(g) % LOOK]);
#ifdef LX_ATTN_ELIM_SOFT
#define LX_HALVES(g, SC)
#else
#define LX_HALVES(g, SC)                                                                                               \
  if (pipe_half(g, 0) >= 0) { constexpr int h_ = pipe_half(g, 0) < 0 ? 0 : pipe_half(g, 0); LX_HALF(h_, SC); }         \
  if (pipe_half(g, 1) >= 0) { constexpr int h_ = pipe_half(g, 1) < 0 ? 0 : pipe_half(g, 1); LX_HALF(h_, SC); }
#endif
#ifdef LX_ATTN_ELIM_DMA
#define LX_PIECES(g)
#else
#define LX_PIECES(g)                                                                                                   \
  if ((g) == LX_ATTN_PG0) piece(0, t2.krow, 0, t2.nclamp, ks_slot);                                                    \
  if ((g) == LX_ATTN_PG1) piece(1, t2.krow, 0, t2.nclamp, ks_slot);                                                    \
  if ((g) == LX_ATTN_PG2) piece(2, 0, t1.vpos, 0, vs_slot);                                                            \
  if ((g) == LX_ATTN_PG3) piece(3, 0, t1.vpos, 0, vs_slot);
#endif
#define LX_GAP(g, SC, SN)                                                                                              \
  LX_WAIT_GAP(g)                                                                                                       \
  LX_MM(g, SC, SN, false);                                                                                             \
  LX_RD((g) + LOOK, 32, false);                                                                                        \
  LX_MCHUNK_MAYBE(g, SC)                                                                                               \
  LX_HALVES(g, SC)                                                                                                     \
  LX_PIECES(g)                                                                                                         \
  __builtin_amdgcn_sched_barrier(0)
#define LX_GAP4(g, SC, SN) LX_GAP(g, SC, SN); LX_GAP((g) + 1, SC, SN); LX_GAP((g) + 2, SC, SN); LX_GAP((g) + 3, SC, SN)

  // One iteration: softmax(t) + PV(t) + QK(t+1).  SC = scores of tile t (complete), SN = scores of tile t+1 (written).
#ifdef LX_ATTN_ELIM_BAR
#define LX_ITER_BARRIER()
#else
#define LX_ITER_BARRIER() LX_BARRIER()
#endif
#define LX_ITER(SC, SN)                                                                                                \
  {                                                                                                                    \
    const int ks_slot = t & 1, vs_slot = (t + 1) & 1;                                                                  \
    bq = ((t + 1) & 1) * STAGE_BYTES;                                                                                  \
    bv = (t & 1) * STAGE_BYTES;                                                                                        \
    LX_WAITL(0);                                                                                                       \
    LX_RDP(0, 32, false); LX_RDP(1, 32, false); LX_RDP(2, 32, false); LX_RDP(3, 32, false); LX_RDP(4, 32, false); LX_RDP(5, 32, false); \
    /* ---- ragged last tile of the segment: mask keys past its end (rare, before the stream) ---- */                  \
    const float bl = t0.bl;                                                                                            \
    if constexpr (MODE == 2) { off = bl; asm volatile("" : "+v"(off)); }   /* the segment pair's bias is the whole offset */ \
    if (t0.nvalid < KVBLK) {                                                                                           \
      __builtin_amdgcn_sched_barrier(0); /* keeps this a wave-uniform branch: if-converted it is ~100 VALU on every tile */ \
      int lh4_ = 4 * lhi;                                                                                              \
      LX_PIN_IN_BRANCH(lh4_);     /* defined inside the rare branch: LICM otherwise hoists the 32 key constants into registers held across the loop */ \
      _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) _Pragma("unroll") for (int r = 0; r < 16; ++r) {                \
        const int key = lh4_ + kb * 32 + 8 * (r >> 2) + (r & 3);                                                       \
        if (key >= t0.nvalid) SC[kb][r] = -1e30f;                                                                      \
      }                                                                                                                \
    }                                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
    LX_GAP4(0, SC, SN); LX_GAP4(4, SC, SN); LX_GAP4(8, SC, SN); LX_GAP4(12, SC, SN);                                   \
    LX_GAP4(16, SC, SN); LX_GAP4(20, SC, SN); LX_GAP4(24, SC, SN); LX_GAP4(28, SC, SN);                                \
    LX_PROBE_A();                                                                                                      \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                   \
    LX_PROBE_B();                                                                                                      \
    LX_ITER_BARRIER();                                                                                                 \
    LX_PROBE_C();                                                                                                      \
    t0 = t1; t1 = t2;                                                                                                  \
    gen_next();                                                                                                        \
    t2 = g_cur;                                                                                                        \
    ++t;                                                                                                               \
  }

#ifdef LX_ATTN_PROBE
  unsigned long long pr_t0 = __builtin_amdgcn_s_memtime(), pr_a = 0, pr_b = 0, pr_c = 0, pr_vm = 0, pr_bar = 0, pr_loop0 = 0;
#define LX_PROBE_A() pr_a = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define LX_PROBE_B() pr_b = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); pr_vm += pr_b - pr_a
#define LX_PROBE_C() pr_c = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); pr_bar += pr_c - pr_b
#else
#define LX_PROBE_A()
#define LX_PROBE_B()
#define LX_PROBE_C()
#endif
  int t = 0;
  if (t0.nvalid > 0) {
    piece(0, t0.krow, 0, t0.nclamp, 0); piece(1, t0.krow, 0, t0.nclamp, 0);
    piece(2, 0, t0.vpos, 0, 0); piece(3, 0, t0.vpos, 0, 0);
    piece(0, t1.krow, 0, t1.nclamp, 1); piece(1, t1.krow, 0, t1.nclamp, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    LX_BARRIER();
    // prologue: scores of tile 0
    LX_WAITL(0);
    LX_RDP(0, 16, true); LX_RDP(1, 16, true); LX_RDP(2, 16, true); LX_RDP(3, 16, true); LX_RDP(4, 16, true); LX_RDP(5, 16, true);
#define LX_PG(g) LX_WAITR((15 - (g)) < (LOOK - 1) ? (15 - (g)) : (LOOK - 1), ring[(g) % LOOK]); LX_MM(g, sA, sA, true); LX_RD((g) + LOOK, 16, true)
    LX_PG(0); LX_PG(1); LX_PG(2); LX_PG(3); LX_PG(4); LX_PG(5); LX_PG(6); LX_PG(7);
    LX_PG(8); LX_PG(9); LX_PG(10); LX_PG(11); LX_PG(12); LX_PG(13); LX_PG(14); LX_PG(15);
#undef LX_PG
    // Every wave must be done READING K(0) before any wave stages K(2) over it (gap 1 of the first iteration): without this
    // barrier a wave that runs ~14 gaps behind its workgroup (it shares its SIMD with an older wave) could fetch fragments of tile 2
    // for its tile-0 scores -- seen as run-to-run differences of single 32-row groups, ~1e-6 per workgroup, only under load
    // (tools/det_block.py). The iterations themselves end in a barrier; the prologue did not.
    LX_BARRIER();
#ifdef LX_ATTN_PROBE
    pr_loop0 = __builtin_amdgcn_s_memtime();
#endif
    while (true) {
      LX_ITER(sA, sB);
      if (t0.nvalid == 0) break;
      LX_ITER(sB, sA);
      if (t0.nvalid == 0) break;
    }
  }
#undef LX_ITER
#undef LX_PROBE_A
#undef LX_PROBE_B
#undef LX_PROBE_C
#undef LX_GAP4
#undef LX_GAP
#undef LX_MCHUNK
#undef LX_MCHUNK_MAYBE
#undef LX_WAIT_GAP
#undef LX_HALVES
#undef LX_PIECES
#undef LX_ITER_BARRIER
#undef LX_EXP2
#undef LX_SCORE_ARG
#undef LX_LSCALE
#undef LX_LSUM
#undef LX_LADD
#undef LX_LPIN
#undef LX_HALF
#undef LX_MM
#undef LX_RD
#undef LX_RDP
#undef LX_WAITL
#undef LX_WAITR
#undef LX_DSR
#undef LX_BARRIER
#undef LX_FENCE

#ifdef LX_ATTN_PROBE
  {
    const unsigned long long pr_end = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x < 4096) {
      unsigned long long* o = lx_attn_probe_buf + ((size_t)blockIdx.x * 8 + wave) * 4;
      o[0] = pr_end - pr_t0; o[1] = pr_end - pr_loop0; o[2] = pr_vm; o[3] = pr_bar;
    }
  }
#endif
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  lx_store_o(lx_o_mode(args), (int*)D.f16_ovf, q_valid, (uint16_t*)D.O + q_row * D.ldo + D.o_col + h * DH, oacc, inv, lhi);
}


// ---------------------------------------------------------------------------------------------------------------
// fp8 (OCP e4m3) attention: BASELINE configs[4] names an "fp8 MFMA attention path". Q, K and V^T come as byte images written
// by lx_qkv_prep_fp8_segs; both products run on v_mfma_f32_32x32x64_f8f6f4 (64-deep, twice the bf16 rate): a 64-key tile is
// 4 MFMAs for S^T = K.Q^T (two key blocks x two 64-wide halves of d) and 4 for O^T += V^T.P^T (four d blocks, the whole tile
// as the k dimension) instead of 16 + 16. Softmax, running max / sum and the O accumulators stay fp32; P is rounded to e4m3
// (values <= 2^DEFER_THR = 256 < 448). Operand convention of the f8f6f4 MFMA (checked on hardware, tools/ubench/fp8_mfma):
// lane (row = lane % 32, g = lane / 32) supplies 32 bytes, and byte p of group g is the same k index on both operands. So the
// K / Q fragments simply take d = half*64 + g*32 + p, the P fragment is the lane's own 32 probabilities in register order
// (p = kb*16 + r), and the V^T image stores its keys in exactly that order (qkv_prep_fp8_kernel, vt8_key()).
// Structure: the plain 8-wave kernel (two waves per SIMD); stage = 8 KiB K (64 keys x 128 B, slot ^= (key>>1)&7) + 8 KiB V^T
// (128 d x 64 B, slot ^= (d>>2)&3), double buffered, one 1-KiB LDS-DMA piece per wave and operand.
typedef int i32x8 __attribute__((ext_vector_type(8)));
constexpr int K8_BYTES = KVBLK * DH;        // 8 KiB
constexpr int V8_BYTES = DH * KVBLK;        // 8 KiB
constexpr int STAGE8_BYTES = K8_BYTES + V8_BYTES;

// (lx_attn_fp8_kernel, the plain e4m3 kernel -- 1147 / 1197 TFLOP/s at S = 2560 / 8704 against 1390 / 1568 for the pipelined one below --
//  was kept as LX_ATTN_FP8_PIPE=0 until round 5.)


// ---------------------------------------------------------------------------------------------------------------
// fp8 attention, software-pipelined (default; LX_ATTN_FP8_PIPE=0 selects the plain kernel above): the principle of
// lx_attn_pipe_kernel -- a wave's vector instructions run in the shadow of ITS OWN MFMAs -- on the 8-MFMA fp8 tile, whose
// bottleneck is the vector pipe (the plain kernel: ~170 vector instructions per key tile and wave against 8 MFMAs of 64 cycles).
// One iteration t = eight gaps, each {LDS fragment reads for the next gap, a slice of vector work, one MFMA}:
//   gaps 0-3 : K.Q^T of tile t+1 (scores of the NEXT tile)  beside  exp2 / e4m3 packing of tile t (8 scores per gap)
//   gaps 4-7 : V^T.P^T of tile t                             beside  row max and rescale decision of tile t+1
// and what the vector pipe no longer does at all:
//   * row sums: a ninth MFMA per tile multiplies P^T by a fragment whose row 0 is all ones (e4m3 1.0): lacc[0] of lanes 0-31
//     accumulates sum_k P[k, q] of the ROUNDED probabilities (the values P.V sees) -- 32 v_add per tile less;
//   * POW2 (the softmax scale x log2 e x operand descale is an exact power of two 2^-k: ops.py picks the q scale that way): the
//     score MFMAs run with the MX block scale 2^-k (E8M0 operand of v_mfma_scale_f32_32x32x64_f8f6f4: exact, free) and start from
//     an accumulator that already holds `bias - running max` (offv: 16 registers, rewritten only when the running max moves), so
//     a finished score IS the exp2 argument -- 32 v_fma per tile less. The running max a tile is computed against is the one
//     BEFORE that tile (its own maximum is not known when its MFMAs start); the deferred-rescale rule absorbs that: a tile whose
//     maximum exceeds the reference by more than 2^DEFER_THR moves the reference and has its 32 scores corrected (rare).
// Score registers: kb = 1 of tile t+1 is produced (gaps 0-1) while kb = 0 of tile t is consumed, kb = 0 of tile t+1 (gaps 2-3) goes
// into the registers kb = 0 of tile t just left: three 16-register sets, not four; the other two swap roles every iteration (the
// loop body is expanded twice). The O rescale of a tile is applied at the head of the iteration that multiplies it in, K(t+2)
// and V^T(t+1) are staged into the slots K(t) / V^T(t-1) vacated (one 1-KiB LDS-DMA piece per wave and operand), one barrier per
// tile. Results differ from lx_attn_fp8_kernel only in rounding (row sums of rounded P, reference max one tile later).
// Round 6 (POW2 form): the probabilities carry a constant factor 2^P8_OFF (it cancels in O / l): a score at the reference becomes 2^6, the
// deferred-rescale threshold drops from 8 to P8_THR = 2 (P <= 2^8 = 256 < 448 as before), and e4m3's normal range then reaches 12 octaves
// BELOW the reference instead of 6 (flush to zero under 2^-15.5 of the reference instead of 2^-9.5): with thousands of keys per row the mass
// that lives there is not negligible (tools/p_loglin.py: peaked rows lose 2-3x less).
// LOGLIN: the vector pipe is this kernel's bottleneck and 32 v_exp_f32 per tile (quarter rate: 16 cycles each) + 16 v_cvt_pk_fp8_f32 are
// half of its vector time. An e4m3 byte IS a piecewise-linear code of log2: byte = 8 * (exponent + 7) + mantissa, so for a normal value
// byte ~ 8 log2(p) + 56. The score MFMAs therefore deliver y = 8 (s - reference) + 8 P8_OFF + 56 directly (MX block scale 2^(3-k), the
// accumulator they start from holds the rest) and ONE v_cvt_pk_u8_f32 per score (saturating at 0: masked and far keys become zero) writes the
// probability byte: p = 2^n (1 + m / 8) with m = the 3 leading bits of the score's fraction -- the chord of 2^f instead of its rounding
// (at most 0.69 of a mantissa step above the true value, +0.46 on average: a constant factor, which cancels). No exp2, no fp8 conversion.
constexpr float P8_OFF = 6.0f, P8_THR = 2.0f;
template <bool DEFER, bool POW2, bool LOGLIN = false>
__global__ __launch_bounds__(512, 1) void lx_attn_fp8_pipe_kernel(const AttnArgs args, float qk_descale, float v_descale, int e8m0) {
  static_assert(POW2 || !LOGLIN, "the log-linear probability bytes need the scores as they come out of the matrix pipe (POW2)");
  constexpr int QBLK = 256;
  // POW2: a finished score y = YU * (s - reference, log2 units) + YZ
  constexpr float YU = LOGLIN ? 8.0f : 1.0f, YZ = LOGLIN ? 8.0f * P8_OFF + 56.0f : P8_OFF;      // (v_cvt_pk_u8_f32 rounds to nearest even: tools/ubench/cvt_u8)
  __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE8_BYTES];      // K slot s at s*STAGE8, V^T slot s at s*STAGE8 + K8
  const lx_attn_desc& D = args.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int BH = D.B * D.H;
  int qt, bh;
  lx_item_decode((int)blockIdx.x, (int)gridDim.x, BH, args.qt_start[3], qt, bh);
  const int b = bh / D.H, h = bh % D.H;
  int sq = 0;
#pragma unroll
  for (int s = 1; s < 3; ++s)
    if (s < D.n_seg && qt >= args.qt_start[s]) sq = s;
  const int n_seg = D.n_seg;
  const int len0 = D.seg_len[0], len1 = D.seg_len[1], len2 = D.seg_len[2];
  const int row00 = D.seg_row0[0], row01 = D.seg_row0[1], row02 = D.seg_row0[2];
  const int vt00 = D.seg_vt0[0], vt01 = D.seg_vt0[1], vt02 = D.seg_vt0[2];
  const float bia0 = D.bias[sq][0], bia1 = D.bias[sq][1], bia2 = D.bias[sq][2];
  auto pick = [](int s, auto x0, auto x1, auto x2) { return s == 0 ? x0 : (s == 1 ? x1 : x2); };
  auto seg_len = [&](int s) { return pick(s, len0, len1, len2); };
  const int q_len = seg_len(sq);
  const int q_in_seg = (qt - args.qt_start[sq]) * QBLK + wave * 32 + l31;
  const bool q_valid = q_in_seg < q_len;
  const size_t q_row = (size_t)pick(sq, row00, row01, row02) + (size_t)b * q_len + min(q_in_seg, q_len - 1);

  i32x8 qf[2];
  {
    const uint8_t* qp = (const uint8_t*)D.Q + q_row * D.ldq + D.q_col + h * DH + lhi * 32;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const u32x4 lo = *(const u32x4*)(qp + hf * 64), hi = *(const u32x4*)(qp + hf * 64 + 16);
      qf[hf] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
    }
  }
  const float c2 = D.scale * qk_descale * 1.4426950408889634f;      // (!POW2: multiplies every score; POW2: it is 2^(e8m0 - 127))
  f32x16 oacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  float m_run = -1e30f;
  f32x16 lacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) lacc[r] = 0.f;
  const int one_w = l31 == 0 ? 0x38383838 : 0;
  const i32x8 ones_frag = {one_w, one_w, one_w, one_w, one_w, one_w, one_w, one_w};

  const uint8_t* Kbase = (const uint8_t*)D.K + D.k_col + h * DH;
  const uint8_t* Vbase = (const uint8_t*)D.VT + (size_t)bh * DH * D.vt_ld;
  const int ldk = D.ldk, vt_ld = D.vt_ld;
  struct Tile { int krow, vpos, nvalid, nclamp; float bl; };
  int g_seg = -1, g_left = 0;
  Tile g_cur = {0, 0, 0, 1, 0.f};
  auto gen_next = [&]() {
    if (g_left > 0) {
      g_cur.krow += KVBLK; g_cur.vpos += KVBLK;
    } else {
      __builtin_amdgcn_sched_barrier(0);
      do { ++g_seg; } while (g_seg < n_seg && !(pick(g_seg, bia0, bia1, bia2) > -1e37f));
      if (g_seg >= n_seg) { g_cur.nvalid = 0; g_cur.nclamp = 1; g_left = 0; return; }   // krow / vpos stay on the last real tile
      g_left = seg_len(g_seg);
      g_cur.krow = pick(g_seg, row00, row01, row02) + b * g_left;
      g_cur.vpos = pick(g_seg, vt00, vt01, vt02);
      g_cur.bl = pick(g_seg, bia0, bia1, bia2) * 1.4426950408889634f;
    }
    g_cur.nvalid = min(g_left, KVBLK);
    g_cur.nclamp = g_cur.nvalid;
    g_left -= g_cur.nvalid;
  };
  gen_next();
  Tile t0 = g_cur;                  // tile t: exp2 / packing / P.V
  gen_next();
  Tile t1 = g_cur;                  // tile t+1: scores, row max, its V^T staged
  gen_next();
  Tile t2 = g_cur;                  // tile t+2: its K staged

  // this wave's staging pieces: K = 8 key rows of 128 B (lane -> row lane>>3, slot lane&7), V^T = 16 d rows of 64 B (row lane>>2, slot lane&3)
  const int k_key = wave * 8 + (lane >> 3);
  const int k_lslot = (lane & 7) ^ ((k_key >> 1) & 7);
  const int v_drow = wave * 16 + (lane >> 2);
  const int v_lslot = (lane & 3) ^ ((v_drow >> 2) & 3);
#ifdef LX_ATTN_FP8_FLAT_DMA      /* A/B: flat-global addresses, 64-bit per-lane address arithmetic per piece */
  auto stage_k = [&](const Tile& T, int slot) {      // rows past the tile's last valid key are clamped (loaded, masked later)
    const uint8_t* src = Kbase + (size_t)(T.krow + min(k_key, T.nclamp - 1)) * ldk + k_lslot * 16;
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + slot * STAGE8_BYTES + wave * 1024), 16, 0, 0);
  };
  auto stage_v = [&](const Tile& T, int slot) {
    const uint8_t* src = Vbase + (size_t)v_drow * vt_ld + T.vpos + v_lslot * 16;
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + slot * STAGE8_BYTES + K8_BYTES + wave * 1024), 16, 0, 0);
  };
#else
  // buffer addressing as in lx_attn_pipe_kernel: wave-uniform tile origin in the SRSRC (scalar instructions), per-lane 32-bit offsets
  // that are loop-invariant registers; the row clamp is min(row * ld + slot, clamp * ld + slot): v_add + v_min per K piece, nothing per V^T piece
  uint32_t k_off8 = (uint32_t)(k_key * ldk + k_lslot * 16), v_off8 = (uint32_t)(v_drow * vt_ld + v_lslot * 16);
  asm volatile("" : "+v"(k_off8), "+v"(v_off8));
  auto stage_k = [&](const Tile& T, int slot) {      // rows past the tile's last valid key are clamped (loaded, masked later)
    const lx_rsrc_t rs = lx_make_rsrc(Kbase + (size_t)T.krow * ldk);
    const uint32_t off_ = min(k_off8, (uint32_t)((T.nclamp - 1) * ldk) + (uint32_t)(k_lslot * 16));
    lx_buf_to_lds(rs, (lptr_t)(smem + slot * STAGE8_BYTES + wave * 1024), off_, 0);
  };
  auto stage_v = [&](const Tile& T, int slot) {
    const lx_rsrc_t rs = lx_make_rsrc(Vbase + T.vpos);
    lx_buf_to_lds(rs, (lptr_t)(smem + slot * STAGE8_BYTES + K8_BYTES + wave * 1024), v_off8, 0);
  };
#endif
  const int ksw = (l31 >> 1) & 7, vsw = (l31 >> 2) & 3;
  auto frag = [&](const char* p0, int slot_a, int sw) {
    const u32x4 lo = *(const u32x4*)(p0 + ((slot_a ^ sw) * 16)), hi = *(const u32x4*)(p0 + (((slot_a + 1) ^ sw) * 16));
    return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
  };
  // fragment of gap g: 0-1 = K(t+1) key block 1 (d half g), 2-3 = K(t+1) key block 0 (d half g-2), 4-7 = V^T(t) d block g-4
#ifdef LX8_ELIM_DSR            /* timing experiments only (WRONG numbers), as LX_ATTN_ELIM_* of the bf16 kernel: tools/run_r03ac.sh */
  i32x8 fr_stale = {1, 2, 3, 4, 5, 6, 7, 8};
  auto frag_of = [&](int g, const char* kbuf, const char* vbuf) { asm volatile("" : "+v"(fr_stale) : "v"(kbuf), "v"(vbuf)); return fr_stale; };
#else
  auto frag_of = [&](int g, const char* kbuf, const char* vbuf) {
    if (g < 4) return frag(kbuf + ((g < 2 ? 32 : 0) + l31) * 128, (g & 1) * 4 + lhi * 2, ksw);
    return frag(vbuf + ((g - 4) * 32 + l31) * 64, lhi * 2, vsw);
  };
#endif
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 s0, s1a, s1b;              // scores: key block 0 (in place), key block 1 of the current / next tile (swap roles)
  f32x16 offv = zero16;             // POW2: bias - running max in all 16 slots = the accumulator the score MFMAs start from
#ifdef LX8_ELIM_SOFT
  int pfw[8] = {0x38302c34, 0x2c383430, 0x34383c30, 0x30343828, 0x38302c34, 0x2c383430, 0x34383c30, 0x30343828};   // (non-zero operands: zeros would run at a higher clock)
#else
  int pfw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  float off = 0.f, alpha = 1.f, mx[4], t_new = 0.f;
  bool resc = false;                 // wave-uniform: O must be multiplied by alpha before the next P.V
  auto qk = [&](const i32x8& a, const i32x8& q, const f32x16& c) {
    if constexpr (POW2) return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, q, c, 0, 0, 0, e8m0, 0, 127);
    else return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, q, c, 0, 0, 0, 0, 0, 0);
  };

#define LX8_FENCE() __builtin_amdgcn_sched_barrier(0)
#ifdef LX8_ELIM_BAR            /* timing experiment (RACY, wrong numbers): no end-of-tile barrier -- the bound on what fewer barriers could buy */
#define LX8_BARRIER()
#else
#define LX8_BARRIER() __builtin_amdgcn_s_barrier()
#endif
  // hipcc linearises pure arithmetic freely inside a basic block (sched_barrier only holds what already sits on either side of it):
  // every slice takes its inputs through an empty asm at the head of its gap and leaves its results through one at the end, and
  // the gap's MFMA takes its fragment through one in front of it and hands its result through one behind it -- that is what keeps
  // {reads, slice, MFMA} in this order.
  // exp2 / packing of eight scores of tile t: chunk c -> key block c>>1 (SCk), register quads 2*(c&1), 2*(c&1)+1
#ifdef LX8_ELIM_EXP
#define LX8_EXP2(x) (x)
#else
#define LX8_EXP2(x) __builtin_amdgcn_exp2f(x)
#endif
#ifdef LX8_ELIM_CVT        /* two f32 -> one v_perm-free integer op instead of v_cvt_pk_fp8_f32 */
#define LX8_CVT(a, b, old, hi) ((int)(__float_as_uint(a) ^ (__float_as_uint(b) >> 3)) + (old))
#else
#define LX8_CVT(a, b, old, hi) __builtin_amdgcn_cvt_pk_fp8_f32(a, b, old, hi)
#endif
#define LX8_SOFT(c, SCk)                                                                                               \
  {                                                                                                                    \
    if constexpr (!POW2) { asm volatile("" : "+v"(off)); }                                                             \
    _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_) {                                                                 \
      constexpr int kb_ = ((c) >> 1) & 1;                                                                              \
      const int rq_ = ((c) & 1) * 2 + q_;                                                                              \
      if constexpr (LOGLIN) {          /* the score IS the byte: one saturating conversion per score, all four bytes of the word rewritten */ \
        unsigned w_ = (unsigned)pfw[kb_ * 4 + rq_];                                                                    \
        _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) w_ = __builtin_amdgcn_cvt_pk_u8_f32(SCk[rq_ * 4 + e_], e_, w_); \
        pfw[kb_ * 4 + rq_] = (int)w_;                                                                                  \
      } else {                                                                                                         \
        float pv_[4];                                                                                                  \
        _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) {                                                             \
          if constexpr (POW2) { float x_ = SCk[rq_ * 4 + e_]; asm volatile("" : "+v"(x_)); pv_[e_] = LX8_EXP2(x_); } \
          else pv_[e_] = __builtin_amdgcn_exp2f(fmaf(SCk[rq_ * 4 + e_], c2, off));                                     \
        }                                                                                                              \
        int w_ = LX8_CVT(pv_[0], pv_[1], pfw[kb_ * 4 + rq_], false);   /* (old value: both halves are overwritten; a literal 0 costs a v_mov) */ \
        w_ = LX8_CVT(pv_[2], pv_[3], w_, true);                                                                        \
        pfw[kb_ * 4 + rq_] = w_;                                                                                       \
      }                                                                                                                \
      asm volatile("" : "+v"(pfw[kb_ * 4 + rq_]));                                                                     \
    }                                                                                                                  \
  }
  // 8-score v_max3 chains of one key block (4 instructions per 8 scores)
#define LX8_MAX8(dst, S, h_)                                                                                           \
  {                                                                                                                    \
    float m_ = __builtin_fmaxf(__builtin_fmaxf(S[8 * (h_)], S[8 * (h_) + 1]), S[8 * (h_) + 2]);                        \
    m_ = __builtin_fmaxf(__builtin_fmaxf(m_, S[8 * (h_) + 3]), S[8 * (h_) + 4]);                                       \
    m_ = __builtin_fmaxf(__builtin_fmaxf(m_, S[8 * (h_) + 5]), S[8 * (h_) + 6]);                                       \
    dst = __builtin_fmaxf(m_, S[8 * (h_) + 7]);                                                                        \
  }
  // row max of tile t+1 (SN0 = key block 0, SN1 = key block 1), rescale decision, reference of tile t+2 -- pieces 0..3 behind gaps 4..7
#define LX8_MAX(c, SN0, SN1)                                                                                           \
  if ((c) == 0) {                                                                                                      \
    LX8_MAX8(mx[2], SN1, 0) LX8_MAX8(mx[3], SN1, 1)                                                                    \
    asm volatile("" : "+v"(mx[2]), "+v"(mx[3]));                                                                       \
  } else if ((c) == 1) {                                                                                               \
    LX8_MAX8(mx[0], SN0, 0) LX8_MAX8(mx[1], SN0, 1)                                                                    \
    asm volatile("" : "+v"(mx[0]), "+v"(mx[1]));                                                                       \
  } else if ((c) == 2) {                                                                                               \
    asm volatile("" : "+v"(mx[0]), "+v"(mx[1]), "+v"(mx[2]), "+v"(mx[3]));                                             \
    float m_ = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(mx[0], mx[1]), mx[2]), mx[3]);                          \
    const auto sw_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(m_), __float_as_uint(m_), false, false);         \
    if constexpr (POW2) t_new = (fmaxf(__uint_as_float(sw_[0]), __uint_as_float(sw_[1])) - YZ) * (1.0f / YU);      /* excess over the reference, log2 units */ \
    else t_new = fmaxf(__uint_as_float(sw_[0]), __uint_as_float(sw_[1])) * c2 + t1.bl;                                 \
    asm volatile("" : "+v"(t_new));                                                                                    \
  } else {                                                                                                             \
    asm volatile("" : "+v"(t_new));                                                                                    \
    if (t1.nvalid > 0) {              /* a tile past the end contributes nothing and must not move the running max */ \
      if constexpr (POW2) {                                                                                            \
        bool r_ = __builtin_amdgcn_ballot_w64(t_new > (DEFER ? P8_THR : 0.f)) != 0;                                    \
        if (r_) {                     /* move the reference up by this row's excess and re-reference the tile's scores */ \
          const float d_ = fmaxf(t_new, 0.f);                                                                          \
          alpha = __builtin_amdgcn_exp2f(-d_);                                                                         \
          m_run += d_;                                                                                                 \
          const float dy_ = d_ * YU;                                                                                   \
          _Pragma("unroll") for (int r2_ = 0; r2_ < 16; ++r2_) { SN0[r2_] -= dy_; SN1[r2_] -= dy_; }                   \
          resc = true;                                                                                                 \
        }                                                                                                              \
        if (r_ || t2.bl != t1.bl) {   /* the accumulator tile t+2's scores start from */                              \
          const float o_ = (t2.bl - m_run) * YU + YZ;                                                                  \
          _Pragma("unroll") for (int r2_ = 0; r2_ < 16; ++r2_) offv[r2_] = o_;                                         \
        }                                                                                                              \
      } else {                                                                                                         \
        bool r_ = true;                                                                                                \
        if (DEFER) r_ = __builtin_amdgcn_ballot_w64(t_new - m_run > DEFER_THR) != 0;                                   \
        if (r_) {                                                                                                      \
          const float m_new = fmaxf(m_run, t_new);                                                                     \
          alpha = __builtin_amdgcn_exp2f(m_run - m_new);                                                               \
          m_run = m_new;                                                                                               \
          resc = true;                                                                                                 \
        }                                                                                                              \
        off = t1.bl - m_run;                                                                                           \
      }                                                                                                                \
    }                                                                                                                  \
    if constexpr (!POW2) { asm volatile("" : "+v"(off)); }                                                             \
  }
  // score MFMA of gap g (0-1: key block 1 -> SN1, 2-3: key block 0 -> SN0) / P.V MFMA of gap g (4-7)
#define LX8_MM(g, SN0, SN1)                                                                                            \
  asm volatile("" : "+v"(fr[(g) & 1]));                                                                                \
  if ((g) < 2) {                                                                                                       \
    SN1 = qk(fr[(g) & 1], qf[(g) & 1], ((g) & 1) ? SN1 : (POW2 ? offv : zero16));                                      \
    asm volatile("" : "+v"(SN1));     /* (an MFMA is pure arithmetic to hipcc too: without this it sinks to its first use) */ \
  } else if ((g) < 4) {                                                                                                \
    SN0 = qk(fr[(g) & 1], qf[(g) & 1], ((g) & 1) ? SN0 : (POW2 ? offv : zero16));                                      \
    asm volatile("" : "+v"(SN0));                                                                                      \
  } else {                                                                                                             \
    const i32x8 pf_ = {pfw[0], pfw[1], pfw[2], pfw[3], pfw[4], pfw[5], pfw[6], pfw[7]};                                \
    oacc[((g) - 4) & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fr[(g) & 1], pf_, oacc[((g) - 4) & 3], 0, 0, 0, 0, 0, 0); \
    asm volatile("" : "+v"(oacc[((g) - 4) & 3]));                                                                      \
  }
#ifdef LX8_ELIM_SOFT
#define LX8_SOFT_MAYBE(c, SCk) { asm volatile("" : "+v"(SCk)); }      /* (the scores stay live: the score MFMAs are not dead code) */
#else
#define LX8_SOFT_MAYBE(c, SCk) LX8_SOFT(c, SCk)
#endif
#ifdef LX8_ELIM_MAX
#define LX8_MAX_MAYBE(c, SN0, SN1) { asm volatile("" : "+v"(SN0), "+v"(SN1)); }
#else
#define LX8_MAX_MAYBE(c, SN0, SN1) LX8_MAX(c, SN0, SN1)
#endif
#ifdef LX8_ELIM_DMA
#define LX8_STAGE_MAYBE(g, PAR)
#else
#ifndef LX8_STAGE_K_GAP          /* measurement builds move the two LDS-DMA pieces of a wave to other gaps (tools/build_variant.sh) */
#define LX8_STAGE_K_GAP 1
#endif
#ifndef LX8_STAGE_V_GAP
#define LX8_STAGE_V_GAP 3          /* (round 6: 5 -> 3, -2.5 % per launch at S = 8704: the piece has two more gaps to land before the end-of-tile wait) */
#endif
#define LX8_STAGE_MAYBE(g, PAR)                                                                                        \
  if ((g) == LX8_STAGE_K_GAP) stage_k(t2, PAR);                                                                        \
  if ((g) == LX8_STAGE_V_GAP) stage_v(t1, (PAR) ^ 1);
#endif
  // one gap: reads of the NEXT gap's fragment, the vector slice, this gap's MFMA.  S0 = key block 0 (tile t, then t+1 in place),
  // S1C / S1N = key block 1 of tile t / t+1
#define LX8_GAP(g, S0, S1C, S1N, PAR)                                                                                  \
  if ((g) < 7) fr[((g) + 1) & 1] = frag_of((g) + 1, kbuf, vbuf);                                                       \
  LX8_FENCE();                                                                                                         \
  if ((g) < 2) { LX8_SOFT_MAYBE((g) & 1, S0) } else if ((g) < 4) { LX8_SOFT_MAYBE(2 + ((g) & 1), S1C) } else { LX8_MAX_MAYBE(((g) - 4) & 3, S0, S1N) } \
  LX8_STAGE_MAYBE(g, PAR)                                                                                              \
  LX8_FENCE();                                                                                                         \
  LX8_MM(g, S0, S1N)                                                                                                   \
  LX8_FENCE();
#define LX8_ITER(S0, S1C, S1N, PAR)    /* PAR = t & 1, a literal: the loop body is expanded once per parity */            \
  {                                                                                                                    \
    const char* kbuf = smem + ((PAR) ^ 1) * STAGE8_BYTES;                   /* K(t+1) */                               \
    const char* vbuf = smem + (PAR) * STAGE8_BYTES + K8_BYTES;              /* V^T(t) */                               \
    i32x8 fr[2];                                                                                                       \
    fr[0] = frag_of(0, kbuf, vbuf);                                                                                    \
    if (resc) {                        /* O of tiles < t was accumulated against the old running max */                \
      _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) oacc[i_][r_] *= alpha; \
      lacc[0] *= alpha;                                                                                                \
      resc = false;                                                                                                    \
    }                                                                                                                  \
    LX8_FENCE();                                                                                                       \
    LX8_GAP(0, S0, S1C, S1N, PAR) LX8_GAP(1, S0, S1C, S1N, PAR) LX8_GAP(2, S0, S1C, S1N, PAR) LX8_GAP(3, S0, S1C, S1N, PAR) \
    if (t1.nvalid > 0 && t1.nvalid < KVBLK) {     /* ragged last tile of a segment: mask keys past its end (rare) */  \
      LX8_FENCE();                                                                                                     \
      int lh4_ = 4 * lhi;                                                                                              \
      LX_PIN_IN_BRANCH(lh4_);    /* defined inside the rare branch: LICM otherwise hoists the 32 key constants into registers (spilled) across the loop */ \
      _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) {                                                              \
        const int key_ = lh4_ + 8 * (r_ >> 2) + (r_ & 3);                                                              \
        if (key_ >= t1.nvalid) S0[r_] = -1e30f;                                                                        \
        if (key_ + 32 >= t1.nvalid) S1N[r_] = -1e30f;                                                                  \
      }                                                                                                                \
    }                                                                                                                  \
    LX8_FENCE();                                                                                                       \
    LX8_GAP(4, S0, S1C, S1N, PAR) LX8_GAP(5, S0, S1C, S1N, PAR) LX8_GAP(6, S0, S1C, S1N, PAR) LX8_GAP(7, S0, S1C, S1N, PAR) \
    {                                   /* row sums of the rounded probabilities of tile t */                          \
      const i32x8 pf_ = {pfw[0], pfw[1], pfw[2], pfw[3], pfw[4], pfw[5], pfw[6], pfw[7]};                              \
      lacc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ones_frag, pf_, lacc, 0, 0, 0, 0, 0, 0);                  \
      asm volatile("" : "+v"(lacc));                                                                                   \
    }                                                                                                                  \
    LX8_PROBE_A();                                                                                                     \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                   \
    LX8_PROBE_B();                                                                                                     \
    LX8_FENCE(); LX8_BARRIER(); LX8_FENCE();                                                                           \
    LX8_PROBE_C();                                                                                                     \
    t0 = t1; t1 = t2;                                                                                                  \
    gen_next();                                                                                                        \
    t2 = g_cur;                                                                                                        \
  }

#ifdef LX_ATTN_PROBE
  unsigned long long pr_t0 = __builtin_amdgcn_s_memtime(), pr_a = 0, pr_b = 0, pr_c = 0, pr_vm = 0, pr_bar = 0, pr_loop0 = 0;
#define LX8_PROBE_A() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); pr_a = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define LX8_PROBE_B() pr_b = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); pr_vm += pr_b - pr_a
#define LX8_PROBE_C() pr_c = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); pr_bar += pr_c - pr_b
#else
#define LX8_PROBE_A()
#define LX8_PROBE_B()
#define LX8_PROBE_C()
#endif
  if (t0.nvalid > 0) {
    stage_k(t0, 0); stage_v(t0, 0); stage_k(t1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // prologue: scores, row max and reference of tile 0 (the running state is empty: no rescale)
    {
      const char* kb0 = smem;
      s0 = qk(frag(kb0 + l31 * 128, 0 + lhi * 2, ksw), qf[0], zero16);
      s0 = qk(frag(kb0 + l31 * 128, 4 + lhi * 2, ksw), qf[1], s0);
      s1a = qk(frag(kb0 + (32 + l31) * 128, 0 + lhi * 2, ksw), qf[0], zero16);
      s1a = qk(frag(kb0 + (32 + l31) * 128, 4 + lhi * 2, ksw), qf[1], s1a);
      if (t0.nvalid < KVBLK) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = 4 * lhi + 8 * (r >> 2) + (r & 3);
          if (key >= t0.nvalid) s0[r] = -1e30f;
          if (key + 32 >= t0.nvalid) s1a[r] = -1e30f;
        }
      }
      float tmax = fmaxf(s0[0], s1a[0]);
#pragma unroll
      for (int r = 1; r < 16; ++r) tmax = __builtin_fmaxf(__builtin_fmaxf(tmax, s0[r]), s1a[r]);
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      if constexpr (POW2) {
        m_run = tmax * (1.0f / YU) + t0.bl;                // log2 units (the MX block scale carries YU)
        const float o0 = (t0.bl - m_run) * YU + YZ;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] += o0; s1a[r] += o0; }
        const float o1 = (t1.bl - m_run) * YU + YZ;
#pragma unroll
        for (int r = 0; r < 16; ++r) offv[r] = o1;
      } else {
        m_run = tmax * c2 + t0.bl;
        off = t0.bl - m_run;
      }
    }
    // every wave is done reading K(0) before the first iteration stages K(2) over it (see lx_attn_pipe_kernel)
    LX8_FENCE(); __builtin_amdgcn_s_barrier(); LX8_FENCE();
#ifdef LX_ATTN_PROBE
    pr_loop0 = __builtin_amdgcn_s_memtime();
#endif
#ifdef LX8_YOUNG_PRIO          /* measurement build: static priority for the second-dispatched half (guide: two waves per SIMD, item 4) */
    if (wave >= 4) __builtin_amdgcn_s_setprio(LX8_YOUNG_PRIO);
#endif
    while (true) {
      LX8_ITER(s0, s1a, s1b, 0);
      if (t0.nvalid == 0) break;
      LX8_ITER(s0, s1b, s1a, 1);
      if (t0.nvalid == 0) break;
    }
  }
#undef LX8_ITER
#undef LX8_PROBE_A
#undef LX8_PROBE_B
#undef LX8_PROBE_C
#undef LX8_GAP
#undef LX8_MM
#undef LX8_MAX
#undef LX8_MAX8
#undef LX8_SOFT
#undef LX8_FENCE

#ifdef LX_ATTN_PROBE
  {
    const unsigned long long pr_end = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x < 4096) {
      unsigned long long* o = lx_attn_probe_buf + ((size_t)blockIdx.x * 8 + wave) * 4;
      o[0] = pr_end - pr_t0; o[1] = pr_end - pr_loop0; o[2] = pr_vm; o[3] = pr_bar;
    }
  }
#endif
  const float l_tot = __shfl(lacc[0], l31, 64);          // lanes 0-31 hold the sum of query l31
  const float inv = l_tot > 0.f ? v_descale / l_tot : 0.f;
  lx_store_o(lx_o_mode(args), (int*)D.f16_ovf, q_valid, (uint16_t*)D.O + q_row * D.ldo + D.o_col + h * DH, oacc, inv, lhi);
}

}  // namespace

// 16-byte epilogue stores need 16-byte aligned output rows and head columns
static int lx_attn_wide_store(const lx_attn_desc* d) { return d->ldo % 8 == 0 && d->o_col % 8 == 0 && ((uintptr_t)d->O & 15) == 0; }

// which segments have queries: qseg_mask when given, else the first n_qseg (0: all)
static int lx_attn_qmask(const lx_attn_desc* d) {
  if (d->qseg_mask != 0) return d->qseg_mask;
  const int nq = d->n_qseg > 0 ? d->n_qseg : d->n_seg;
  return (1 << nq) - 1;
}

static thread_local int lx_attn_last = LX_ATTN_KERNEL_NONE;
extern "C" int lx_attn_last_kernel(void) { return lx_attn_last; }

extern "C" int lx_attn_fwd(const lx_attn_desc* d, void* stream) {
  LX_CHECK_ARG(d && d->Q && d->K && d->VT && d->O, "lx_attn_fwd: NULL operand");
  LX_CHECK_ARG(d->n_seg >= 1 && d->n_seg <= 3, "lx_attn_fwd: n_seg=%d must be 1..3", d->n_seg);
  LX_CHECK_ARG(d->B >= 1 && d->H >= 1, "lx_attn_fwd: bad B/H");
  LX_CHECK_ARG(d->ldq % 8 == 0 && d->ldk % 8 == 0 && d->ldo % 4 == 0 && d->vt_ld % 64 == 0, "lx_attn_fwd: ldq/ldk %% 8, ldo %% 4, vt_ld %% 64 required");
  LX_CHECK_ARG(d->q_col % 8 == 0 && d->k_col % 8 == 0 && d->o_col % 4 == 0, "lx_attn_fwd: column offsets must be 16-byte aligned");
  const int qblk = 256;                        // query rows per workgroup (8 waves x 32 / 4 waves x 64)
  LX_CHECK_ARG(d->n_qseg >= 0 && d->n_qseg <= d->n_seg, "lx_attn_fwd: n_qseg=%d must be 0..n_seg", d->n_qseg);
  LX_CHECK_ARG(d->qseg_mask >= 0 && d->qseg_mask < (1 << d->n_seg), "lx_attn_fwd: qseg_mask=%d names a segment >= n_seg", d->qseg_mask);
  const int qmask = lx_attn_qmask(d);
  AttnArgs a;
  a.d = *d;
  a.wide_store = lx_attn_wide_store(d);
  int t = 0;
  for (int s = 0; s < 3; ++s) {
    a.qt_start[s] = t;
    if (s < d->n_seg) {
      LX_CHECK_ARG(d->seg_len[s] >= 1, "lx_attn_fwd: empty segment %d", s);
      LX_CHECK_ARG(d->seg_vt0[s] % 64 == 0, "lx_attn_fwd: seg_vt0 must be 64-aligned");
      bool any = false;
      for (int k = 0; k < d->n_seg; ++k) any |= d->bias[s][k] > -1e37f;
      if (!((qmask >> s) & 1)) continue;     // a segment without queries (n_qseg / qseg_mask): keys / values only
      LX_CHECK_ARG(any, "lx_attn_fwd: query segment %d is masked from every key segment", s);
      t += (d->seg_len[s] + qblk - 1) / qblk;
    }
  }
  a.qt_start[3] = t;
  const int grid = t * d->B * d->H;
  hipStream_t st = (hipStream_t)stream;
  LX_CHECK_ARG((d->flags & ~(LX_ATTN_Q_LOG2 | LX_ATTN_BOUNDED | LX_ATTN_INVARIANT | LX_ATTN_O_F16 | LX_ATTN_PREFER_4WAVE)) == 0 &&
                   (!(d->flags & LX_ATTN_BOUNDED) || (d->flags & LX_ATTN_Q_LOG2)) && !((d->flags & LX_ATTN_INVARIANT) && (d->flags & LX_ATTN_PREFER_4WAVE)),
               "lx_attn_fwd: flags=%d: unknown bit, LX_ATTN_BOUNDED without LX_ATTN_Q_LOG2, or LX_ATTN_INVARIANT with LX_ATTN_PREFER_4WAVE", d->flags);
  bool any_bias = false;                       // a finite non-zero bias on a pair that is attended to
  for (int s = 0; s < d->n_seg; ++s)
    for (int k = 0; k < d->n_seg; ++k) any_bias |= ((qmask >> s) & 1) && d->bias[s][k] > -1e37f && d->bias[s][k] != 0.f;
  static const bool nomax_ok = [] { const char* e = getenv("LX_ATTN_NOMAX"); return e ? atoi(e) != 0 : true; }();
  // lx_attn4_kernel (attn4.hip: one wave per SIMD, every K / V^T fragment feeds two MFMAs, persistent over the query tiles) serves the
  // bounded-score contract; its staging addresses a tile as buffer base + 32-bit byte offsets, so the K column block and the V^T image
  // have to lie within 2 GiB each. LX_ATTN_PREFER_4WAVE: whenever it can; LX_ATTN_INVARIANT: never; otherwise where it measured faster than the 8-wave kernel on
  // MI355X (profiles/r04a_attn4_ab.txt): launches of at least two rounds of workgroups whose items are at most 64 key tiles long
  // (B = 16, S = 2560: +1.4 %; 16 x 64 x 2048: +1.7 %) -- one round (B = 1: -2.7 % at S = 2560) and long items (S = 8704: -1.4 %) stay
  // on the 8-wave kernel.
  if ((d->flags & LX_ATTN_BOUNDED) && nomax_ok && !(d->flags & LX_ATTN_INVARIANT)) {
    long long max_row = 0, key_tiles = 0;
    for (int s = 0; s < d->n_seg; ++s) {
      max_row = std::max(max_row, (long long)d->seg_row0[s] + (long long)d->B * d->seg_len[s]);
      key_tiles += (d->seg_len[s] + 63) / 64;
    }
    const bool fits = ((max_row + 64) * (long long)d->ldk + (long long)d->H * 128) * 2 < (1ll << 31) &&
                      ((long long)d->B * d->H + 1) * 128 * d->vt_ld * 2 < (1ll << 31);
    const bool wanted = (d->flags & LX_ATTN_PREFER_4WAVE) || (grid >= 2 * lx_attn4_cus() && key_tiles <= 64);
    if (fits && wanted) {
      lx_attn4_launch(&a, grid, any_bias ? 2 : 1, stream);
      LX_LAUNCH_CHECK("lx_attn_fwd (lx_attn4_kernel)");
      lx_attn_last = LX_ATTN_KERNEL_4WAVE;
      return LX_OK;
    }
  }
  // one kernel per contract: bounded scores without / with a bias (no running maximum), or the max-tracking form with the deferred rescale
  if ((d->flags & LX_ATTN_BOUNDED) && nomax_ok) {
    if (any_bias) hipLaunchKernelGGL((lx_attn_pipe_kernel<true, 2>), dim3(grid), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((lx_attn_pipe_kernel<true, 1>), dim3(grid), dim3(512), 0, st, a);
  } else {
    hipLaunchKernelGGL((lx_attn_pipe_kernel<true, 0>), dim3(grid), dim3(512), 0, st, a);
  }
  LX_LAUNCH_CHECK("lx_attn_fwd");
  lx_attn_last = LX_ATTN_KERNEL_8WAVE;
  return LX_OK;
}

extern "C" int lx_attn_fwd_fp8(const lx_attn_desc* d, float qk_descale, float v_descale, void* stream) {
  LX_CHECK_ARG(d && d->Q && d->K && d->VT && d->O, "lx_attn_fwd_fp8: NULL operand");
  LX_CHECK_ARG(d->n_seg >= 1 && d->n_seg <= 3, "lx_attn_fwd_fp8: n_seg=%d must be 1..3", d->n_seg);
  LX_CHECK_ARG(d->B >= 1 && d->H >= 1, "lx_attn_fwd_fp8: bad B/H");
  LX_CHECK_ARG(d->ldq % 16 == 0 && d->ldk % 16 == 0 && d->ldo % 4 == 0 && d->vt_ld % 64 == 0, "lx_attn_fwd_fp8: ldq/ldk %% 16 (bytes), ldo %% 4, vt_ld %% 64 required");
  LX_CHECK_ARG(d->q_col % 16 == 0 && d->k_col % 16 == 0 && d->o_col % 4 == 0, "lx_attn_fwd_fp8: column offsets must be 16-byte aligned");
  LX_CHECK_ARG(qk_descale > 0.f && v_descale > 0.f, "lx_attn_fwd_fp8: descale factors must be positive");
  LX_CHECK_ARG((d->flags & ~(LX_ATTN_O_F16 | LX_ATTN_P_EXP2)) == 0, "lx_attn_fwd_fp8: the flags are LX_ATTN_O_F16 and LX_ATTN_P_EXP2 (the e4m3 kernels fold their own scales)");
  LX_CHECK_ARG(d->n_qseg >= 0 && d->n_qseg <= d->n_seg, "lx_attn_fwd_fp8: n_qseg=%d must be 0..n_seg", d->n_qseg);
  LX_CHECK_ARG(d->qseg_mask >= 0 && d->qseg_mask < (1 << d->n_seg), "lx_attn_fwd_fp8: qseg_mask=%d names a segment >= n_seg", d->qseg_mask);
  const int qmask = lx_attn_qmask(d);
  AttnArgs a;
  a.d = *d;
  a.wide_store = lx_attn_wide_store(d);
  int t = 0;
  for (int s = 0; s < 3; ++s) {
    a.qt_start[s] = t;
    if (s < d->n_seg) {
      LX_CHECK_ARG(d->seg_len[s] >= 1, "lx_attn_fwd_fp8: empty segment %d", s);
      LX_CHECK_ARG(d->seg_vt0[s] % 64 == 0, "lx_attn_fwd_fp8: seg_vt0 must be 64-aligned");
      bool any = false;
      for (int k = 0; k < d->n_seg; ++k) any |= d->bias[s][k] > -1e37f;
      if (!((qmask >> s) & 1)) continue;
      LX_CHECK_ARG(any, "lx_attn_fwd_fp8: query segment %d is masked from every key segment", s);
      t += (d->seg_len[s] + 255) / 256;
    }
  }
  a.qt_start[3] = t;
  const int grid = t * d->B * d->H;
  // softmax scale x log2(e) x operand descale an exact power of two 2^-k (ops.py chooses the q scale so)? Then the score MFMAs
  // apply it as an MX block scale and the scores come out as exp2 arguments (lx_attn_fp8_pipe_kernel, POW2); any other combination of
  // scales runs the generic form (one fma per score)
  const double c2 = (double)d->scale * (double)qk_descale * 1.4426950408889634;
  const int k = (int)lround(-log2(c2));
  const bool pow2 = k >= 1 && k <= 60 && fabs(c2 * ldexp(1.0, k) - 1.0) < 1e-5;
  hipStream_t st = (hipStream_t)stream;
  if (pow2 && !(d->flags & LX_ATTN_P_EXP2)) hipLaunchKernelGGL((lx_attn_fp8_pipe_kernel<true, true, true>), dim3(grid), dim3(512), 0, st, a, qk_descale, v_descale, 127 - k + 3);
  else if (pow2) hipLaunchKernelGGL((lx_attn_fp8_pipe_kernel<true, true, false>), dim3(grid), dim3(512), 0, st, a, qk_descale, v_descale, 127 - k);
  else hipLaunchKernelGGL((lx_attn_fp8_pipe_kernel<true, false>), dim3(grid), dim3(512), 0, st, a, qk_descale, v_descale, 127);
  LX_LAUNCH_CHECK("lx_attn_fwd_fp8");
  return LX_OK;
}
