// attn.hip -- joint (text | image | condition) flash attention forward for gfx950, head_dim 128, bf16.
//
// Replaces F.scaled_dot_product_attention and the mask / c_factor bias construction of the reference's
// attn_forward (src/flux/block.py:101-135): non-causal softmax(Q K^T / sqrt(dh) + bias) V over the
// concatenation of up to three token segments, with the bias constant per (query segment, key segment)
// pair (0, log(c_factor), or -inf for the union_cond_attn=False / independent_condition masks), so the
// [S,S] mask tensor is never materialised.
//
// Structure (one workgroup = 4 waves = 128 query rows of one (batch, head); lane = query row):
//   S^T[key, q] = K_tile . Q^T      v_mfma_f32_32x32x16_bf16, K fragments from LDS, Q resident in VGPRs
//   online softmax in registers     each lane owns 32 scores of ITS query row; the other 32 sit in lane^32
//   O^T[d, q]  += V^T_tile . P^T    P fragments are 8 consecutive accumulator registers (no shuffles):
//                                   the key order inside every 16-key group is interleaved as
//                                   [0-3, 8-11, 4-7, 12-15] on BOTH operands, which lx_qkv_prep bakes
//                                   into the V^T image it writes, so V^T fragments are plain ds_read_b128.
//   K tile [64 keys][128] and V^T tile [128 d][64 keys] are staged with global_load_lds (16 B/lane),
//   double buffered, one barrier per tile; 16-B slots are XOR-swizzled on the source address
//   (K: slot ^= key&15, V^T: slot ^= (d>>1)&7) so fragment reads are bank-conflict free.
#include "common.h"

namespace {

constexpr int DH = 128;
constexpr int QBLK = 128;
constexpr int KVBLK = 64;
constexpr int NTHREADS = 256;
constexpr int K_BYTES = KVBLK * DH * 2;   // 16 KiB
constexpr int V_BYTES = DH * KVBLK * 2;   // 16 KiB
constexpr int STAGE_BYTES = K_BYTES + V_BYTES;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct AttnArgs {
  lx_attn_desc d;
  int qt_start[4];   // prefix of 128-row query tiles per segment
};

__global__ __launch_bounds__(NTHREADS, 2) void lx_attn_kernel(const AttnArgs args) {
  __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE_BYTES];
  const lx_attn_desc& D = args.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;

  // block -> (query tile, batch*head); all query tiles of one (b,h) share an XCD when B*H % 8 == 0
  const int BH = D.B * D.H;
  const int bh = blockIdx.x % BH;
  const int qt = blockIdx.x / BH;
  const int b = bh / D.H, h = bh % D.H;
  int sq = 0;
#pragma unroll
  for (int s = 1; s < 3; ++s)
    if (s < D.n_seg && qt >= args.qt_start[s]) sq = s;
  const int q_in_seg = (qt - args.qt_start[sq]) * QBLK + wave * 32 + l31;
  const int q_len = D.seg_len[sq];
  const bool q_valid = q_in_seg < q_len;
  const size_t q_row = (size_t)D.seg_row0[sq] + (size_t)b * q_len + min(q_in_seg, q_len - 1);

  // Q fragments: lane (q = l31, half = lhi) holds d = ks*16 + lhi*8 .. +8 for ks = 0..7
  bf16x8 qf[8];
  {
    const __bf16* qp = (const __bf16*)D.Q + q_row * D.ldq + D.q_col + h * DH + lhi * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
  }

  const float c2 = D.scale * 1.4426950408889634f;  // scores are kept in log2 units

  f32x16 oacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  // ---- staging --------------------------------------------------------------------------------------
  const __bf16* Kbase = (const __bf16*)D.K + D.k_col + h * DH;
  const __bf16* Vbase = (const __bf16*)D.VT + (size_t)bh * DH * D.vt_ld;
  auto stage = [&](int sk, int kt, int buf) {
    char* base = smem + buf * STAGE_BYTES;
    const int klen = D.seg_len[sk];
    const size_t krow0 = (size_t)D.seg_row0[sk] + (size_t)b * klen;
    // K: one instruction = 4 key rows of 256 B; lane -> (row = lane>>4, slot = lane&15)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int key = (j * 4 + wave) * 4 + (lane >> 4);
      const int lslot = (lane & 15) ^ (key & 15);
      const int kin = min(kt * KVBLK + key, klen - 1);
      const __bf16* src = Kbase + (krow0 + kin) * D.ldk + lslot * 8;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(base + (j * 4 + wave) * 1024), 16, 0, 0);
    }
    // V^T: one instruction = 8 d rows of 128 B; lane -> (row = lane>>3, slot = lane&7)
    const int vpos = D.seg_vt0[sk] + kt * KVBLK;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int drow = (j * 4 + wave) * 8 + (lane >> 3);
      const int lslot = (lane & 7) ^ ((drow >> 1) & 7);
      const __bf16* src = Vbase + (size_t)drow * D.vt_ld + vpos + lslot * 8;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(base + K_BYTES + (j * 4 + wave) * 1024), 16, 0, 0);
    }
  };
  // iteration over (key segment, key tile), skipping fully masked segment pairs
  auto seg_ok = [&](int s) { return D.bias[sq][s] > -1e37f; };
  auto advance = [&](int& sk, int& kt) {
    ++kt;
    while (sk < D.n_seg && (kt * KVBLK >= D.seg_len[sk] || !seg_ok(sk))) { ++sk; kt = 0; }
  };
  int sk = 0, kt = -1;
  advance(sk, kt);

  const int ksw = l31 & 15;            // K rows: slot ^= key & 15
  const int vsw = (l31 >> 1) & 7;      // V^T rows: slot ^= (d >> 1) & 7
  const int k_row_off = l31 * 256;
  const int v_row_off = K_BYTES + l31 * 128;

  if (sk < D.n_seg) stage(sk, kt, 0);
  int buf = 0;
  while (sk < D.n_seg) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int nsk = sk, nkt = kt;
    advance(nsk, nkt);
    if (nsk < D.n_seg) stage(nsk, nkt, buf ^ 1);
    const char* sb = smem + buf * STAGE_BYTES;

    // ---- S^T = K . Q^T ---------------------------------------------------------------------------
    f32x16 sacc[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const bf16x8 kf = *(const bf16x8*)(sb + kb * 32 * 256 + k_row_off + (((ks * 2 + lhi) ^ ksw) * 16));
        sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sacc[kb], 0, 0, 0);
      }
    }
    // ---- online softmax (log2 domain) ------------------------------------------------------------
    const float bl = D.bias[sq][sk] * 1.4426950408889634f;
    const int klen = D.seg_len[sk];
    const int kbase = kt * KVBLK + 4 * lhi;
    if (kt * KVBLK + KVBLK > klen) {   // ragged last tile of the segment: mask keys past its end
      __builtin_amdgcn_sched_barrier(0);   // keep this a (wave-uniform) branch: if-converted it costs 64 VALU on every tile
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kbase + kb * 32 + 8 * (r >> 2) + (r & 3);
          if (key >= klen) sacc[kb][r] = -1e30f;
        }
    }
    float tmax = sacc[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sacc[kb][r]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax * c2 + bl);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    const float off = bl - m_new;
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(sacc[kb][r], c2, off));
        sacc[kb][r] = p;
        psum += p;
      }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
    // ---- P fragments: step s uses accumulator registers [8*(s&1), +8) of sacc[s>>1] -------------
    bf16x8 pf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      u32x4 w;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        w[i] = pack_bf16x2(sacc[s >> 1][8 * (s & 1) + 2 * i], sacc[s >> 1][8 * (s & 1) + 2 * i + 1]);
      pf[s] = __builtin_bit_cast(bf16x8, w);
    }
    // ---- O^T += V^T . P^T ------------------------------------------------------------------------
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const bf16x8 vf = *(const bf16x8*)(sb + v_row_off + db * 32 * 128 + (((s * 2 + lhi) ^ vsw) * 16));
        oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[s], oacc[db], 0, 0, 0);
      }
    }
    sk = nsk;
    kt = nkt;
    buf ^= 1;
  }

  // ---- epilogue: O[q, d] = O^T / l ; lane holds d = db*32 + 8*(r>>2) + 4*lhi + (r&3) --------------
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  if (q_valid) {
    uint16_t* op = (uint16_t*)D.O + q_row * D.ldo + D.o_col + h * DH + 4 * lhi;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        u32x2 o;
        o[0] = pack_bf16x2(oacc[db][rq * 4 + 0] * inv, oacc[db][rq * 4 + 1] * inv);
        o[1] = pack_bf16x2(oacc[db][rq * 4 + 2] * inv, oacc[db][rq * 4 + 3] * inv);
        *(u32x2*)(op + db * 32 + rq * 8) = o;
      }
  }
}

}  // namespace

extern "C" int lx_attn_fwd(const lx_attn_desc* d, void* stream) {
  LX_CHECK_ARG(d && d->Q && d->K && d->VT && d->O, "lx_attn_fwd: NULL operand");
  LX_CHECK_ARG(d->n_seg >= 1 && d->n_seg <= 3, "lx_attn_fwd: n_seg=%d must be 1..3", d->n_seg);
  LX_CHECK_ARG(d->B >= 1 && d->H >= 1, "lx_attn_fwd: bad B/H");
  LX_CHECK_ARG(d->ldq % 8 == 0 && d->ldk % 8 == 0 && d->ldo % 4 == 0 && d->vt_ld % 64 == 0, "lx_attn_fwd: ldq/ldk %% 8, ldo %% 4, vt_ld %% 64 required");
  LX_CHECK_ARG(d->q_col % 8 == 0 && d->k_col % 8 == 0 && d->o_col % 4 == 0, "lx_attn_fwd: column offsets must be 16-byte aligned");
  AttnArgs a;
  a.d = *d;
  int t = 0;
  for (int s = 0; s < 3; ++s) {
    a.qt_start[s] = t;
    if (s < d->n_seg) {
      LX_CHECK_ARG(d->seg_len[s] >= 1, "lx_attn_fwd: empty segment %d", s);
      LX_CHECK_ARG(d->seg_vt0[s] % 64 == 0, "lx_attn_fwd: seg_vt0 must be 64-aligned");
      bool any = false;
      for (int k = 0; k < d->n_seg; ++k) any |= d->bias[s][k] > -1e37f;
      LX_CHECK_ARG(any, "lx_attn_fwd: query segment %d is masked from every key segment", s);
      t += (d->seg_len[s] + QBLK - 1) / QBLK;
    }
  }
  a.qt_start[3] = t;
  const int grid = t * d->B * d->H;
  hipLaunchKernelGGL(lx_attn_kernel, dim3(grid), dim3(NTHREADS), 0, (hipStream_t)stream, a);
  LX_LAUNCH_CHECK("lx_attn_fwd");
  return LX_OK;
}
