// precise.hip -- the fp32-class ("precise") variant of the DiT step for gfx950: what runs when the model is configured the way
// the reference ships it (train/config/seed_512.yaml:2, dtype float32) and parity to ~1e-4 matters more than throughput.
//
//   * GEMMs stay on the bf16 MFMA (gemm.hip, lx_gemm_split_kernel): every activation operand is carried as a bf16 hi/lo pair
//     (x = hi + lo to 16 mantissa bits) and multiplied in 2 (bf16-exact weights) or 3 K-segments into one fp32 accumulation;
//   * this file holds the producers of those pairs -- fp32 -> [hi | lo] split, AdaLN LayerNorm + modulation with a split output --
//     the fp32 per-head RMSNorm + RoPE on fp32 q / k, and the fp32 attention on the f32-input MFMA
//     (v_mfma_f32_32x32x2_f32: exact fp32 products and accumulation, 1/16 of the bf16 rate), which writes its output as a pair.
//
// Reference arithmetic: src/flux/block.py:7-176 (attn_forward), :192-207 / :301-305 (AdaLN), diffusers apply_rotary_emb / RMSNorm.
#include "common.h"

namespace {

__device__ __forceinline__ void split2(float x, uint16_t& hi, uint16_t& lo) {
  hi = f32_to_bf16(x);
  lo = f32_to_bf16(x - bf16_to_f32(hi));
}

// ---- fp32 [M, K] -> bf16 [M, ...]: hi at column k, lo at column lo_off + k ---------------------------------------------------
__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ src, int lds, uint16_t* __restrict__ dst, int ldd, int lo_off,
                                                         int M, int K) {
  const int kq = K >> 2;
  const size_t n = (size_t)M * kq, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int m = (int)(i / kq), k = (int)(i % kq) * 4;
    const f32x4 v = *(const f32x4*)(src + (size_t)m * lds + k);
    uint16_t h[4], l[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) split2(v[c], h[c], l[c]);
    uint16_t* d = dst + (size_t)m * ldd + k;
    *(u32x2*)d = u32x2{(uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16)};
    *(u32x2*)(d + lo_off) = u32x2{(uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16)};
  }
}

// ---- LayerNorm (no affine) + AdaLN modulation, fp32 in -> bf16 hi/lo pair out. One wave per row. --------------------------------
struct LnSegsP {
  int n;
  int row0[3], n_rows[3], rows_per_batch[3];
  const float* shift[3];
  const float* scale[3];
};

__global__ __launch_bounds__(256) void ln_modulate_split_kernel(const float* __restrict__ X, int ldx, const LnSegsP segs, int mod_ld,
                                                                uint16_t* __restrict__ Y, int ldy, int lo_off, int M, int D, float eps) {
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  int sg = 0, acc_rows = 0;
  while (sg < segs.n - 1 && row >= acc_rows + segs.n_rows[sg]) { acc_rows += segs.n_rows[sg]; ++sg; }
  const int rin = row - acc_rows;
  row = segs.row0[sg] + rin;
  const int lane = threadIdx.x & 63;
  const float* xr = X + (size_t)row * ldx;
  float s = 0.f;
  for (int c = lane * 4; c < D; c += 256) {
    const f32x4 v = *(const f32x4*)(xr + c);
    s += (v[0] + v[1]) + (v[2] + v[3]);
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
  for (int c = lane * 4; c < D; c += 256) {
    const f32x4 v = *(const f32x4*)(xr + c);
#pragma unroll
    for (int k = 0; k < 4; ++k) q += (v[k] - mean) * (v[k] - mean);
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
  const int b = rin / segs.rows_per_batch[sg];
  const float* sh = segs.shift[sg] + (size_t)b * mod_ld;
  const float* sc = segs.scale[sg] + (size_t)b * mod_ld;
  uint16_t* yr = Y + (size_t)row * ldy;
  for (int c = lane * 4; c < D; c += 256) {
    const f32x4 v = *(const f32x4*)(xr + c);
    const f32x4 a = *(const f32x4*)(sc + c);
    const f32x4 bsh = *(const f32x4*)(sh + c);
    uint16_t h[4], l[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) split2((v[k] - mean) * rstd * (1.0f + a[k]) + bsh[k], h[k], l[k]);
    *(u32x2*)(yr + c) = u32x2{(uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16)};
    *(u32x2*)(yr + lo_off + c) = u32x2{(uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16)};
  }
}

// ---- per-head RMSNorm(weight) + interleaved-pair RoPE on fp32 q and k, in place -------------------------------------------------
// Block = 64 positions x 1 head x 1 batch; 16 lanes own one (row, head) vector of 128 floats (8 each = 4 rotary pairs).
struct QkvSegsP {
  int n;
  int row0[3], rows_per_batch[3], tile0[4];
  const float* wq[3]; const float* wk[3]; const float* cos_tab[3]; const float* sin_tab[3];
};

__global__ __launch_bounds__(256) void qkv_prep_f32_kernel(float* __restrict__ QKV, int ld, int q_col, int k_col, const QkvSegsP segs, float eps) {
  int sg = 0;
  while (sg < segs.n - 1 && (int)blockIdx.x >= segs.tile0[sg + 1]) ++sg;
  const int p0 = ((int)blockIdx.x - segs.tile0[sg]) * 64;
  const int rows_per_batch = segs.rows_per_batch[sg];
  const float* __restrict__ cos_tab = segs.cos_tab[sg];
  const float* __restrict__ sin_tab = segs.sin_tab[sg];
  const int h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const int sub = tid & 15, rloc = tid >> 4;
  const size_t rbase = (size_t)segs.row0[sg] + (size_t)b * rows_per_batch;
#pragma unroll 1
  for (int pass = 0; pass < 4; ++pass) {
    const int p = p0 + pass * 16 + rloc;
    const bool valid = p < rows_per_batch;
    float* rowp = QKV + (rbase + (valid ? p : 0)) * ld + h * 128 + sub * 8;
    f32x4 c0 = {1.f, 1.f, 1.f, 1.f}, c1 = c0, s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    if (cos_tab && valid) {
      const float* ct = cos_tab + (size_t)p * 128 + sub * 8;
      const float* st = sin_tab + (size_t)p * 128 + sub * 8;
      c0 = *(const f32x4*)ct; c1 = *(const f32x4*)(ct + 4);
      s0 = *(const f32x4*)st; s1 = *(const f32x4*)(st + 4);
    }
#pragma unroll
    for (int which = 0; which < 2; ++which) {   // 0: q, 1: k
      const float* wn = which ? segs.wk[sg] : segs.wq[sg];
      float* ptr = rowp + (which ? k_col : q_col);
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
      if (valid) { a0 = *(const f32x4*)ptr; a1 = *(const f32x4*)(ptr + 4); }
      float x[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
      if (wn) {
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) ss += x[i] * x[i];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        const float r = rsqrtf(ss * (1.0f / 128.0f) + eps);
        const f32x4 w0 = *(const f32x4*)(wn + sub * 8), w1 = *(const f32x4*)(wn + sub * 8 + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { x[i] = x[i] * r * w0[i]; x[4 + i] = x[4 + i] * r * w1[i]; }
      }
      float y[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {   // pairs (2i, 2i+1): out = x*cos + rot*sin, rot = (-x_odd, x_even)
        const float ce = i < 2 ? c0[2 * i] : c1[2 * i - 4], co = i < 2 ? c0[2 * i + 1] : c1[2 * i - 3];
        const float se = i < 2 ? s0[2 * i] : s1[2 * i - 4], so = i < 2 ? s0[2 * i + 1] : s1[2 * i - 3];
        y[2 * i] = x[2 * i] * ce - x[2 * i + 1] * se;
        y[2 * i + 1] = x[2 * i + 1] * co + x[2 * i] * so;
      }
      if (valid) {
        *(f32x4*)ptr = f32x4{y[0], y[1], y[2], y[3]};
        *(f32x4*)(ptr + 4) = f32x4{y[4], y[5], y[6], y[7]};
      }
    }
  }
}

// ---- fp32 joint attention over up to 3 token segments on v_mfma_f32_32x32x2_f32 -----------------------------------------------------
// Workgroup = 4 waves x 32 queries of one (batch, head, query segment). Per 32-key tile:
//   S^T[key][query] = K . Q^T   (A = K tile from LDS, row stride 129 floats: conflict-free column reads; B = the lane's own
//                                query row, 64 values of its d-parity held in registers, pre-multiplied by scale * log2(e)):
//                                the lane (query j = lane % 32, half = lane / 32) ends with the 16 keys 8*(r/4) + 4*half + r%4;
//   online softmax in registers (fp32, log2 units), the two halves of a query exchange their maxima with one shuffle;
//   O^T[d][query] += V^T . P^T  (A = V tile from LDS, B = the probability registers AS THEY ARE: k-step r pairs exactly the two
//                                keys the two half-waves hold in register r -- no shuffles, no LDS round trip for P).
// Keys past a segment's end are masked; query rows past the end are computed on a clamped row and not stored.
struct AttnF32Args {
  const float* QKV; int ld, q_col, k_col, v_col;
  uint16_t* O; int ldo, o_col, o_lo_off;
  int B, H, n_seg;
  int seg_row0[3], seg_len[3], qtile0[4];
  float bias[3][3];       // log2 units; -INFINITY masks the pair
  float scale_log2e;
};

constexpr int KS_LD = 129;

__global__ __launch_bounds__(256) void attn_f32_kernel(const AttnF32Args a) {
  __shared__ float Ks[32 * KS_LD];
  __shared__ __attribute__((aligned(16))) float Vs[32 * 128];
  int sq = 0;
  while (sq < a.n_seg - 1 && (int)blockIdx.x >= a.qtile0[sq + 1]) ++sq;
  const int h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int q_pos = ((int)blockIdx.x - a.qtile0[sq]) * 128 + wave * 32 + l31;
  const bool q_valid = q_pos < a.seg_len[sq];
  const size_t q_row = (size_t)a.seg_row0[sq] + (size_t)b * a.seg_len[sq] + (q_valid ? q_pos : 0);
  // the lane's query row: the 64 values of its d-parity, scaled
  float qv[64];
  {
    const float* qp = a.QKV + q_row * a.ld + a.q_col + h * 128;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const f32x4 v = *(const f32x4*)(qp + 4 * j);
      qv[2 * j] = (hi ? v[1] : v[0]) * a.scale_log2e;
      qv[2 * j + 1] = (hi ? v[3] : v[2]) * a.scale_log2e;
    }
  }
  f32x16 o[4];
#pragma unroll
  for (int dblk = 0; dblk < 4; ++dblk)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dblk][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  for (int sk = 0; sk < a.n_seg; ++sk) {
    const float bias = a.bias[sq][sk];
    if (bias == -INFINITY) continue;                                  // wave-uniform: the whole pair is masked
    const int klen = a.seg_len[sk];
    const size_t k_row0 = (size_t)a.seg_row0[sk] + (size_t)b * klen;
    for (int k0 = 0; k0 < klen; k0 += 32) {
      __syncthreads();                                                // every wave is done with the previous tile
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int key = pass * 8 + (tid >> 5), j = tid & 31;
        f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = kv;
        if (k0 + key < klen) {
          const float* rp = a.QKV + (k_row0 + k0 + key) * a.ld + h * 128 + 4 * j;
          kv = *(const f32x4*)(rp + a.k_col);
          vv = *(const f32x4*)(rp + a.v_col);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) Ks[key * KS_LD + 4 * j + c] = kv[c];
        *(f32x4*)&Vs[key * 128 + 4 * j] = vv;
      }
      __syncthreads();
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
      const float* kp = Ks + l31 * KS_LD + hi;
#pragma unroll
      for (int st = 0; st < 64; ++st) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[2 * st], qv[st], s, 0, 0, 0);
      // mask + bias, tile maximum
      float mt = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = 8 * (r >> 2) + 4 * hi + (r & 3);
        s[r] = (k0 + key < klen) ? s[r] + bias : -INFINITY;
        mt = fmaxf(mt, s[r]);
      }
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      const float m_new = fmaxf(m_run, mt);
      const float m_use = m_new == -INFINITY ? 0.f : m_new;          // (a fully masked row so far: exp2(-inf - 0) = 0 everywhere)
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = __builtin_amdgcn_exp2f(s[r] - m_use);
        ps += s[r];
      }
      l_run = l_run * alpha + ps;
      m_run = m_new;
#pragma unroll
      for (int dblk = 0; dblk < 4; ++dblk)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dblk][r] *= alpha;
#pragma unroll
      for (int dblk = 0; dblk < 4; ++dblk) {
        const float* vp = Vs + dblk * 32 + l31 + 4 * hi * 128;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dblk] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[(8 * (r >> 2) + (r & 3)) * 128], s[r], o[dblk], 0, 0, 0);
      }
    }
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  if (!q_valid) return;
  uint16_t* op = a.O + q_row * a.ldo + a.o_col + h * 128;
#pragma unroll
  for (int dblk = 0; dblk < 4; ++dblk)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      uint16_t hh[4], ll[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) split2(o[dblk][rq * 4 + c] * inv, hh[c], ll[c]);
      const int d = dblk * 32 + 8 * rq + 4 * hi;
      *(u32x2*)(op + d) = u32x2{(uint32_t)hh[0] | ((uint32_t)hh[1] << 16), (uint32_t)hh[2] | ((uint32_t)hh[3] << 16)};
      if (a.o_lo_off) *(u32x2*)(op + a.o_lo_off + d) = u32x2{(uint32_t)ll[0] | ((uint32_t)ll[1] << 16), (uint32_t)ll[2] | ((uint32_t)ll[3] << 16)};
    }
}

// ====================================================================================================================================
// Precise-mode attention on the bf16 matrix pipe (round 3). attn_f32_kernel above runs both products on v_mfma_f32_32x32x2_f32,
// 1/16 of the bf16 MFMA rate: half of a precise denoise step. The GEMMs of this mode already use the split-bf16 trick
// (lx_gemm_split_kernel); here attention does the same: x = x_hi + x_lo with x_hi = bf16(x), x_lo = bf16(x - x_hi) (16 mantissa
// bits), and a product a.b is evaluated as a_hi.b_hi + a_hi.b_lo + a_lo.b_hi on v_mfma_f32_32x32x16_bf16 into ONE fp32
// accumulation (the dropped a_lo.b_lo term is 2^-16 x 2^-16 relative): 3/16 of the fp32-MFMA cost per product.
//   S^T = K.Q^T   : K_hi.Q_hi + K_hi.Q_lo + K_lo.Q_hi          (q, k: fp32 RMSNorm + RoPE first, lx_qkv_prep_split_segs)
//   softmax       : fp32, exact running maximum (no deferred rescale: p <= 1, so p_hi + p_lo carries p to 2^-17)
//   O^T = V^T.P^T : V_hi.P_hi + V_lo.P_hi + V_hi.P_lo
// Structure = the plain 8-wave bf16 kernel of attn.hip (lane = query row, swapped QK^T so softmax is in-register, P fragments are
// 8 consecutive accumulator registers thanks to the 16-key interleave of the V^T images), with four 16-KiB operand tiles per
// stage (K_hi, K_lo, V^T_hi, V^T_lo), double buffered = 128 KiB of LDS, eight 1-KiB LDS-DMA pieces per wave and tile.
// ====================================================================================================================================
typedef const __attribute__((address_space(1))) void* sp_gptr_t;
typedef __attribute__((address_space(3))) void* sp_lptr_t;

__device__ __forceinline__ int sp_vt_interleave(int key) {   // within every 16 keys: [0-3, 8-11, 4-7, 12-15] (= rowops.hip / gemm.hip)
  return (key & ~15) | (((key >> 2) & 1) << 3) | (((key >> 3) & 1) << 2) | (key & 3);
}

struct QkvSplitArgs {
  const float* QKV; int ld, q_col, k_col, v_col;
  uint16_t* QK2; int ld2, q2_col, k2_col, lo_off;
  uint16_t* VT2; int vt_ld; long long vt_lo_off;
  int vt_pos0[3];
  int H;
  float eps;
};

// Block = 64 positions x 1 head x 1 batch (as qkv_prep_f32_kernel); q / k: RMSNorm + RoPE in fp32, written as bf16 hi / lo pairs;
// v: hi / lo pairs of the raw projection into the two V^T images (transposed through LDS, 16-key interleave).
__global__ __launch_bounds__(256) void qkv_prep_split_kernel(const QkvSplitArgs a, const QkvSegsP segs) {
  __shared__ float vt_s[64][128 + 4];
  int sg = 0;
  while (sg < segs.n - 1 && (int)blockIdx.x >= segs.tile0[sg + 1]) ++sg;
  const int p0 = ((int)blockIdx.x - segs.tile0[sg]) * 64;
  const int rows_per_batch = segs.rows_per_batch[sg];
  const float* __restrict__ cos_tab = segs.cos_tab[sg];
  const float* __restrict__ sin_tab = segs.sin_tab[sg];
  const int h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const int sub = tid & 15, rloc = tid >> 4;
  const size_t rbase = (size_t)segs.row0[sg] + (size_t)b * rows_per_batch;
#pragma unroll 1
  for (int pass = 0; pass < 4; ++pass) {
    const int p = p0 + pass * 16 + rloc;
    const bool valid = p < rows_per_batch;
    const size_t grow = rbase + (valid ? p : 0);
    const float* rowp = a.QKV + grow * a.ld + h * 128 + sub * 8;
    f32x4 c0 = {1.f, 1.f, 1.f, 1.f}, c1 = c0, s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    if (cos_tab && valid) {
      const float* ct = cos_tab + (size_t)p * 128 + sub * 8;
      const float* st = sin_tab + (size_t)p * 128 + sub * 8;
      c0 = *(const f32x4*)ct; c1 = *(const f32x4*)(ct + 4);
      s0 = *(const f32x4*)st; s1 = *(const f32x4*)(st + 4);
    }
#pragma unroll
    for (int which = 0; which < 2; ++which) {   // 0: q, 1: k
      const float* wn = which ? segs.wk[sg] : segs.wq[sg];
      const float* ptr = rowp + (which ? a.k_col : a.q_col);
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
      if (valid) { a0 = *(const f32x4*)ptr; a1 = *(const f32x4*)(ptr + 4); }
      float x[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
      if (wn) {
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) ss += x[i] * x[i];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        const float r = rsqrtf(ss * (1.0f / 128.0f) + a.eps);
        const f32x4 w0 = *(const f32x4*)(wn + sub * 8), w1 = *(const f32x4*)(wn + sub * 8 + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { x[i] = x[i] * r * w0[i]; x[4 + i] = x[4 + i] * r * w1[i]; }
      }
      float y[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {   // pairs (2i, 2i+1): out = x*cos + rot*sin, rot = (-x_odd, x_even)
        const float ce = i < 2 ? c0[2 * i] : c1[2 * i - 4], co = i < 2 ? c0[2 * i + 1] : c1[2 * i - 3];
        const float se = i < 2 ? s0[2 * i] : s1[2 * i - 4], so = i < 2 ? s0[2 * i + 1] : s1[2 * i - 3];
        y[2 * i] = x[2 * i] * ce - x[2 * i + 1] * se;
        y[2 * i + 1] = x[2 * i + 1] * co + x[2 * i] * so;
      }
      if (valid) {
        uint16_t hh[8], ll[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) split2(y[i], hh[i], ll[i]);
        uint16_t* op = a.QK2 + grow * a.ld2 + (which ? a.k2_col : a.q2_col) + h * 128 + sub * 8;
        u32x4 oh, ol;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          oh[i] = (uint32_t)hh[2 * i] | ((uint32_t)hh[2 * i + 1] << 16);
          ol[i] = (uint32_t)ll[2 * i] | ((uint32_t)ll[2 * i + 1] << 16);
        }
        *(u32x4*)op = oh;
        *(u32x4*)(op + a.lo_off) = ol;
      }
    }
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
    if (valid) { v0 = *(const f32x4*)(rowp + a.v_col); v1 = *(const f32x4*)(rowp + a.v_col + 4); }
    *(f32x4*)&vt_s[pass * 16 + rloc][sub * 8] = v0;
    *(f32x4*)&vt_s[pass * 16 + rloc][sub * 8 + 4] = v1;
  }
  __syncthreads();
  // V^T rows: thread -> (d, 8 consecutive slots); slot -> source key via the (involutive) interleave
  uint16_t* vtb = a.VT2 + ((size_t)(b * a.H + h) * 128) * a.vt_ld + a.vt_pos0[sg] + p0;
  // (eight lanes per d: the eight key groups sit a multiple of 32 banks apart -- half of this kernel's LDS cycles are conflicts -- but
  //  the lanes of a d write 128 contiguous bytes; lanes along d instead (conflict-free reads, one 16-byte store per V^T row and lane)
  //  measured 47.9 vs 42.4 us)
  for (int item = tid; item < 128 * 8; item += 256) {
    const int d = item >> 3, g = item & 7;
    u32x4 oh, ol;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint16_t h0, l0, h1, l1;
      split2(vt_s[sp_vt_interleave(g * 8 + 2 * i)][d], h0, l0);
      split2(vt_s[sp_vt_interleave(g * 8 + 2 * i + 1)][d], h1, l1);
      oh[i] = (uint32_t)h0 | ((uint32_t)h1 << 16);
      ol[i] = (uint32_t)l0 | ((uint32_t)l1 << 16);
    }
    *(u32x4*)(vtb + (size_t)d * a.vt_ld + g * 8) = oh;
    *(u32x4*)(vtb + a.vt_lo_off + (size_t)d * a.vt_ld + g * 8) = ol;
  }
}

struct AttnSplitArgs {
  lx_attn_desc d;
  int qt_start[4];
  int qk_lo_off;          // columns from the hi to the lo image of q and k
  long long vt_lo_off;    // elements from the hi to the lo V^T image
  int o_lo_off;           // columns from the hi to the lo half of the output pair (0: hi only)
};

constexpr int SP_KV = 64, SP_DH = 128;
constexpr int SP_TILE = SP_KV * SP_DH * 2;     // one operand tile: 16 KiB
constexpr int SP_STAGE = 4 * SP_TILE;          // K_hi | K_lo | V^T_hi | V^T_lo

// BOUNDED (LX_ATTN_Q_LOG2 | LX_ATTN_BOUNDED, include/lx.h): q carries scale * log2 e and the caller bounds the scores: no running
// maximum, no rescale of O -- p = exp2(s + bias)
template <bool BOUNDED>
__global__ __launch_bounds__(512, 1) void attn_split_kernel(const AttnSplitArgs args) {
  constexpr int NW = 8, QBLK = 256, PIECES = 2;
  __shared__ __attribute__((aligned(1024))) char smem[2 * SP_STAGE];
  const lx_attn_desc& D = args.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
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

  bf16x8 qh[8], ql[8];
  {
    const __bf16* qp = (const __bf16*)D.Q + q_row * D.ldq + D.q_col + h * SP_DH + lhi * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      qh[ks] = *(const bf16x8*)(qp + ks * 16);
      ql[ks] = *(const bf16x8*)(qp + args.qk_lo_off + ks * 16);
    }
  }
  const float c2 = (D.flags & LX_ATTN_Q_LOG2) ? 1.0f : D.scale * 1.4426950408889634f;   // scores are kept in log2 units
  f32x16 oacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  const __bf16* Kbase = (const __bf16*)D.K + D.k_col + h * SP_DH;
  const __bf16* Vbase = (const __bf16*)D.VT + (size_t)bh * SP_DH * D.vt_ld;
  auto stage = [&](int sk, int kt, int buf) {
    char* base = smem + buf * SP_STAGE;
    const int klen = D.seg_len[sk];
    const size_t krow0 = (size_t)D.seg_row0[sk] + (size_t)b * klen;
#pragma unroll
    for (int img = 0; img < 2; ++img) {
      // K: one instruction = 4 key rows of 256 B; lane -> (row = lane>>4, slot = lane&15), slot ^= key & 15
#pragma unroll
      for (int j = 0; j < PIECES; ++j) {
        const int key = (j * NW + wave) * 4 + (lane >> 4);
        const int lslot = (lane & 15) ^ (key & 15);
        const int kin = min(kt * SP_KV + key, klen - 1);
        const __bf16* src = Kbase + (krow0 + kin) * D.ldk + img * args.qk_lo_off + lslot * 8;
        __builtin_amdgcn_global_load_lds((sp_gptr_t)src, (sp_lptr_t)(base + img * SP_TILE + (j * NW + wave) * 1024), 16, 0, 0);
      }
      // V^T: one instruction = 8 d rows of 128 B; lane -> (row = lane>>3, slot = lane&7), slot ^= (d >> 1) & 7
      const int vpos = D.seg_vt0[sk] + kt * SP_KV;
#pragma unroll
      for (int j = 0; j < PIECES; ++j) {
        const int drow = (j * NW + wave) * 8 + (lane >> 3);
        const int lslot = (lane & 7) ^ ((drow >> 1) & 7);
        const __bf16* src = Vbase + img * args.vt_lo_off + (size_t)drow * D.vt_ld + vpos + lslot * 8;
        __builtin_amdgcn_global_load_lds((sp_gptr_t)src, (sp_lptr_t)(base + (2 + img) * SP_TILE + (j * NW + wave) * 1024), 16, 0, 0);
      }
    }
  };
  auto seg_ok = [&](int s) { return D.bias[sq][s] > -1e37f; };
  auto advance = [&](int& sk, int& kt) {
    ++kt;
    while (sk < D.n_seg && (kt * SP_KV >= D.seg_len[sk] || !seg_ok(sk))) { ++sk; kt = 0; }
  };
  int sk = 0, kt = -1;
  advance(sk, kt);
  const int ksw = l31 & 15;
  const int vsw = (l31 >> 1) & 7;
  const int k_row_off = l31 * 256;
  const int v_row_off = 2 * SP_TILE + l31 * 128;

  if (sk < D.n_seg) stage(sk, kt, 0);
  int buf = 0;
  while (sk < D.n_seg) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int nsk = sk, nkt = kt;
    advance(nsk, nkt);
    if (nsk < D.n_seg) stage(nsk, nkt, buf ^ 1);
    const char* sb = smem + buf * SP_STAGE;
    // ---- S^T = K . Q^T : the two small cross terms first, then hi.hi -------------------------------
    f32x16 sacc[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int off = kb * 32 * 256 + k_row_off + (((ks * 2 + lhi) ^ ksw) * 16);
        const bf16x8 kfh = *(const bf16x8*)(sb + off);
        const bf16x8 kfl = *(const bf16x8*)(sb + SP_TILE + off);
        sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfl, qh[ks], sacc[kb], 0, 0, 0);
        sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfh, ql[ks], sacc[kb], 0, 0, 0);
        sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfh, qh[ks], sacc[kb], 0, 0, 0);
      }
    }
    // ---- online softmax (fp32, log2 domain, exact running maximum) ---------------------------------
    const float bl = D.bias[sq][sk] * 1.4426950408889634f;
    const int klen = D.seg_len[sk];
    const int kbase = kt * SP_KV + 4 * lhi;
    if (kt * SP_KV + SP_KV > klen) {   // ragged last tile of the segment: mask keys past its end
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kbase + kb * 32 + 8 * (r >> 2) + (r & 3);
          if (key >= klen) sacc[kb][r] = -1e30f;
        }
    }
    float off = bl;
    if constexpr (!BOUNDED) {
      float tmax = fmaxf(sacc[0][0], sacc[0][1]);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = (kb == 0 ? 2 : 0); r < 16; r += 2) tmax = __builtin_fmaxf(__builtin_fmaxf(tmax, sacc[kb][r]), sacc[kb][r + 1]);
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      const float m_new = fmaxf(m_run, tmax * c2 + bl);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
      m_run = m_new;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
      off = bl - m_run;
    }
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(BOUNDED ? sacc[kb][r] + off : fmaf(sacc[kb][r], c2, off));
        sacc[kb][r] = p;
        psum += p;
      }
    l_run += psum;
    // ---- O^T += V^T . P^T; the P fragment of step s = accumulator registers [8*(s&1), +8) of sacc[s>>1], as a hi / lo pair ----
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      u32x4 wh, wl;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float p0 = sacc[s >> 1][8 * (s & 1) + 2 * i], p1 = sacc[s >> 1][8 * (s & 1) + 2 * i + 1];
        const uint16_t h0 = f32_to_bf16(p0), h1 = f32_to_bf16(p1);
        wh[i] = (uint32_t)h0 | ((uint32_t)h1 << 16);
        wl[i] = pack_bf16x2(p0 - bf16_to_f32(h0), p1 - bf16_to_f32(h1));
      }
      const bf16x8 ph = __builtin_bit_cast(bf16x8, wh), pl = __builtin_bit_cast(bf16x8, wl);
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const int off2 = v_row_off + db * 32 * 128 + (((s * 2 + lhi) ^ vsw) * 16);
        const bf16x8 vfh = *(const bf16x8*)(sb + off2);
        const bf16x8 vfl = *(const bf16x8*)(sb + SP_TILE + off2);
        oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfl, ph, oacc[db], 0, 0, 0);
        oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfh, pl, oacc[db], 0, 0, 0);
        oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfh, ph, oacc[db], 0, 0, 0);
      }
    }
    sk = nsk;
    kt = nkt;
    buf ^= 1;
  }
  // ---- epilogue: O[q, d] = O^T / l as a bf16 hi / lo pair; lane holds d = db*32 + 8*(r>>2) + 4*lhi + (r&3) ----
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  if (q_valid) {
    uint16_t* op = (uint16_t*)D.O + q_row * D.ldo + D.o_col + h * SP_DH + 4 * lhi;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        uint16_t hh[4], ll[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) split2(oacc[db][rq * 4 + c] * inv, hh[c], ll[c]);
        *(u32x2*)(op + db * 32 + rq * 8) = u32x2{(uint32_t)hh[0] | ((uint32_t)hh[1] << 16), (uint32_t)hh[2] | ((uint32_t)hh[3] << 16)};
        if (args.o_lo_off)
          *(u32x2*)(op + args.o_lo_off + db * 32 + rq * 8) = u32x2{(uint32_t)ll[0] | ((uint32_t)ll[1] << 16), (uint32_t)ll[2] | ((uint32_t)ll[3] << 16)};
      }
  }
}

}  // namespace

extern "C" int lx_split_bf16(const float* src, int lds, void* dst, int ldd, int lo_off, int M, int K, void* stream) {
  LX_CHECK_ARG(src && dst && M > 0 && K > 0, "lx_split_bf16: bad arguments");
  LX_CHECK_ARG(K % 4 == 0 && lds % 4 == 0 && ldd % 4 == 0 && lo_off % 4 == 0 && lo_off >= K && ldd >= lo_off + K, "lx_split_bf16: K, lds, ldd, lo_off must be multiples of 4 and K <= lo_off, lo_off + K <= ldd");
  LX_CHECK_ARG(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 7) == 0, "lx_split_bf16: misaligned operand");
  const size_t n = (size_t)M * (K / 4);
  const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(split_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, lds, (uint16_t*)dst, ldd, lo_off, M, K);
  LX_LAUNCH_CHECK("lx_split_bf16");
  return LX_OK;
}

extern "C" int lx_ln_modulate_split_segs(const float* X, int ldx, const lx_ln_seg* seg, int n_seg, int mod_ld, void* Y, int ldy, int y_lo_off,
                                         int D, float eps, void* stream) {
  LX_CHECK_ARG(seg && n_seg >= 1 && n_seg <= 3, "lx_ln_modulate_split_segs: 1..3 segments");
  LX_CHECK_ARG(X && Y && D > 0 && D % 4 == 0, "lx_ln_modulate_split_segs: bad arguments");
  LX_CHECK_ARG(ldx % 4 == 0 && ldy % 4 == 0 && mod_ld % 4 == 0 && y_lo_off % 4 == 0 && y_lo_off >= D && ldy >= y_lo_off + D,
               "lx_ln_modulate_split_segs: ldx/ldy/mod_ld/y_lo_off must be multiples of 4 and D <= y_lo_off, y_lo_off + D <= ldy");
  LX_CHECK_ARG(((uintptr_t)X & 15) == 0 && ((uintptr_t)Y & 7) == 0, "lx_ln_modulate_split_segs: misaligned operand");
  LnSegsP segs;
  segs.n = n_seg;
  int M = 0;
  for (int i = 0; i < n_seg; ++i) {
    LX_CHECK_ARG(seg[i].shift && seg[i].scale && seg[i].n_rows > 0 && seg[i].rows_per_batch > 0, "lx_ln_modulate_split_segs: bad segment %d", i);
    LX_CHECK_ARG((((uintptr_t)seg[i].shift | (uintptr_t)seg[i].scale) & 15) == 0, "lx_ln_modulate_split_segs: misaligned modulation table");
    segs.row0[i] = seg[i].row0; segs.n_rows[i] = seg[i].n_rows; segs.rows_per_batch[i] = seg[i].rows_per_batch;
    segs.shift[i] = seg[i].shift; segs.scale[i] = seg[i].scale;
    M += seg[i].n_rows;
  }
  hipLaunchKernelGGL(ln_modulate_split_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, X, ldx, segs, mod_ld, (uint16_t*)Y, ldy,
                     y_lo_off, M, D, eps);
  LX_LAUNCH_CHECK("lx_ln_modulate_split_segs");
  return LX_OK;
}

extern "C" int lx_qkv_prep_f32_segs(float* QKV, int ld, int q_col, int k_col, const lx_qkv_seg* seg, int n_seg, int n_batches, int H,
                                    float eps, void* stream) {
  LX_CHECK_ARG(seg && n_seg >= 1 && n_seg <= 3, "lx_qkv_prep_f32_segs: 1..3 segments");
  LX_CHECK_ARG(QKV && n_batches > 0 && H > 0, "lx_qkv_prep_f32_segs: bad arguments");
  LX_CHECK_ARG(ld % 4 == 0 && q_col % 4 == 0 && k_col % 4 == 0 && ((uintptr_t)QKV & 15) == 0, "lx_qkv_prep_f32_segs: ld / column offsets must be multiples of 4, QKV 16-byte aligned");
  QkvSegsP segs;
  segs.n = n_seg;
  int t = 0;
  for (int i = 0; i < n_seg; ++i) {
    LX_CHECK_ARG(seg[i].rows_per_batch > 0, "lx_qkv_prep_f32_segs: empty segment %d", i);
    LX_CHECK_ARG((seg[i].cos_tab == nullptr) == (seg[i].sin_tab == nullptr), "lx_qkv_prep_f32_segs: cos/sin tables must come together");
    segs.row0[i] = seg[i].row0; segs.rows_per_batch[i] = seg[i].rows_per_batch;
    segs.wq[i] = seg[i].wq; segs.wk[i] = seg[i].wk; segs.cos_tab[i] = seg[i].cos_tab; segs.sin_tab[i] = seg[i].sin_tab;
    segs.tile0[i] = t;
    t += (seg[i].rows_per_batch + 63) / 64;
  }
  segs.tile0[n_seg] = t;
  hipLaunchKernelGGL(qkv_prep_f32_kernel, dim3(t, H, n_batches), dim3(256), 0, (hipStream_t)stream, QKV, ld, q_col, k_col, segs, eps);
  LX_LAUNCH_CHECK("lx_qkv_prep_f32_segs");
  return LX_OK;
}

extern "C" int lx_attn_fwd_f32(const lx_attn_f32_desc* d, void* stream) {
  LX_CHECK_ARG(d && d->QKV && d->O, "lx_attn_fwd_f32: NULL operand");
  LX_CHECK_ARG(d->B > 0 && d->H > 0 && d->n_seg >= 1 && d->n_seg <= 3, "lx_attn_fwd_f32: bad B/H/n_seg");
  LX_CHECK_ARG(d->ld % 4 == 0 && d->q_col % 4 == 0 && d->k_col % 4 == 0 && d->v_col % 4 == 0 && ((uintptr_t)d->QKV & 15) == 0,
               "lx_attn_fwd_f32: ld / column offsets must be multiples of 4, QKV 16-byte aligned");
  LX_CHECK_ARG(d->ldo % 4 == 0 && d->o_col % 4 == 0 && d->o_lo_off % 4 == 0 && d->o_lo_off >= 0 && ((uintptr_t)d->O & 7) == 0, "lx_attn_fwd_f32: ldo / o_col / o_lo_off must be multiples of 4");
  AttnF32Args a;
  a.QKV = d->QKV; a.ld = d->ld; a.q_col = d->q_col; a.k_col = d->k_col; a.v_col = d->v_col;
  a.O = (uint16_t*)d->O; a.ldo = d->ldo; a.o_col = d->o_col; a.o_lo_off = d->o_lo_off;
  a.B = d->B; a.H = d->H; a.n_seg = d->n_seg;
  int t = 0;
  for (int i = 0; i < d->n_seg; ++i) {
    LX_CHECK_ARG(d->seg_len[i] > 0, "lx_attn_fwd_f32: empty segment %d", i);
    a.seg_row0[i] = d->seg_row0[i]; a.seg_len[i] = d->seg_len[i];
    a.qtile0[i] = t;
    t += (d->seg_len[i] + 127) / 128;
  }
  a.qtile0[d->n_seg] = t;
  const float log2e = 1.4426950408889634f;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) a.bias[i][j] = d->bias[i][j] * log2e;        // -inf stays -inf
  a.scale_log2e = d->scale * log2e;
  hipLaunchKernelGGL(attn_f32_kernel, dim3(t, d->H, d->B), dim3(256), 0, (hipStream_t)stream, a);
  LX_LAUNCH_CHECK("lx_attn_fwd_f32");
  return LX_OK;
}


extern "C" int lx_qkv_prep_split_segs(const float* QKV, int ld, int q_col, int k_col, int v_col, const lx_qkv_seg* seg, int n_seg, int n_batches,
                                      int H, float eps, void* QK2, int ld2, int q2_col, int k2_col, int lo_off, void* VT2, int vt_ld,
                                      long long vt_lo_off, void* stream) {
  LX_CHECK_ARG(seg && n_seg >= 1 && n_seg <= 3, "lx_qkv_prep_split_segs: 1..3 segments");
  LX_CHECK_ARG(QKV && QK2 && VT2 && n_batches > 0 && H > 0, "lx_qkv_prep_split_segs: bad arguments");
  LX_CHECK_ARG(ld % 4 == 0 && q_col % 4 == 0 && k_col % 4 == 0 && v_col % 4 == 0 && ((uintptr_t)QKV & 15) == 0,
               "lx_qkv_prep_split_segs: ld / column offsets must be multiples of 4, QKV 16-byte aligned");
  LX_CHECK_ARG(ld2 % 8 == 0 && q2_col % 8 == 0 && k2_col % 8 == 0 && lo_off % 8 == 0 && lo_off >= H * 128 && ((uintptr_t)QK2 & 15) == 0,
               "lx_qkv_prep_split_segs: ld2 / q2_col / k2_col / lo_off must be multiples of 8 (lo_off >= H*128), QK2 16-byte aligned");
  LX_CHECK_ARG(vt_ld % 64 == 0 && vt_lo_off % 8 == 0 && vt_lo_off >= (long long)n_batches * H * 128 * vt_ld && ((uintptr_t)VT2 & 15) == 0,
               "lx_qkv_prep_split_segs: vt_ld %% 64, vt_lo_off %% 8 and the lo V^T image behind the hi image required");
  QkvSegsP segs;
  QkvSplitArgs a;
  segs.n = n_seg;
  int t = 0;
  for (int i = 0; i < n_seg; ++i) {
    LX_CHECK_ARG(seg[i].rows_per_batch > 0, "lx_qkv_prep_split_segs: empty segment %d", i);
    LX_CHECK_ARG((seg[i].cos_tab == nullptr) == (seg[i].sin_tab == nullptr), "lx_qkv_prep_split_segs: cos/sin tables must come together");
    LX_CHECK_ARG(seg[i].vt_pos0 % 64 == 0 && seg[i].vt_pos0 >= 0, "lx_qkv_prep_split_segs: vt_pos0 must be a multiple of 64");
    segs.row0[i] = seg[i].row0; segs.rows_per_batch[i] = seg[i].rows_per_batch;
    segs.wq[i] = seg[i].wq; segs.wk[i] = seg[i].wk; segs.cos_tab[i] = seg[i].cos_tab; segs.sin_tab[i] = seg[i].sin_tab;
    a.vt_pos0[i] = seg[i].vt_pos0;
    segs.tile0[i] = t;
    t += (seg[i].rows_per_batch + 63) / 64;
  }
  segs.tile0[n_seg] = t;
  a.QKV = QKV; a.ld = ld; a.q_col = q_col; a.k_col = k_col; a.v_col = v_col;
  a.QK2 = (uint16_t*)QK2; a.ld2 = ld2; a.q2_col = q2_col; a.k2_col = k2_col; a.lo_off = lo_off;
  a.VT2 = (uint16_t*)VT2; a.vt_ld = vt_ld; a.vt_lo_off = vt_lo_off; a.H = H; a.eps = eps;
  hipLaunchKernelGGL(qkv_prep_split_kernel, dim3(t, H, n_batches), dim3(256), 0, (hipStream_t)stream, a, segs);
  LX_LAUNCH_CHECK("lx_qkv_prep_split_segs");
  return LX_OK;
}

extern "C" int lx_attn_fwd_split(const lx_attn_desc* d, int qk_lo_off, long long vt_lo_off, int o_lo_off, void* stream) {
  LX_CHECK_ARG(d && d->Q && d->K && d->VT && d->O, "lx_attn_fwd_split: NULL operand");
  LX_CHECK_ARG(d->n_seg >= 1 && d->n_seg <= 3, "lx_attn_fwd_split: n_seg=%d must be 1..3", d->n_seg);
  LX_CHECK_ARG(d->B >= 1 && d->H >= 1 && d->n_qseg == 0 && d->qseg_mask == 0, "lx_attn_fwd_split: bad B/H (n_qseg / qseg_mask are not supported here)");
  LX_CHECK_ARG(d->ldq % 8 == 0 && d->ldk % 8 == 0 && d->ldo % 4 == 0 && d->vt_ld % 64 == 0, "lx_attn_fwd_split: ldq/ldk %% 8, ldo %% 4, vt_ld %% 64 required");
  LX_CHECK_ARG(d->q_col % 8 == 0 && d->k_col % 8 == 0 && d->o_col % 4 == 0, "lx_attn_fwd_split: column offsets must be 16-byte aligned");
  LX_CHECK_ARG(qk_lo_off % 8 == 0 && qk_lo_off >= d->H * 128 && vt_lo_off % 8 == 0 && vt_lo_off > 0 && o_lo_off % 4 == 0 && o_lo_off >= 0,
               "lx_attn_fwd_split: qk_lo_off %% 8 (>= H*128), vt_lo_off %% 8 (> 0), o_lo_off %% 4 (>= 0) required");
  AttnSplitArgs a;
  a.d = *d;
  a.qk_lo_off = qk_lo_off; a.vt_lo_off = vt_lo_off; a.o_lo_off = o_lo_off;
  int t = 0;
  for (int s = 0; s < 3; ++s) {
    a.qt_start[s] = t;
    if (s < d->n_seg) {
      LX_CHECK_ARG(d->seg_len[s] >= 1, "lx_attn_fwd_split: empty segment %d", s);
      LX_CHECK_ARG(d->seg_vt0[s] % 64 == 0, "lx_attn_fwd_split: seg_vt0 must be 64-aligned");
      bool any = false;
      for (int k = 0; k < d->n_seg; ++k) any |= d->bias[s][k] > -1e37f;
      LX_CHECK_ARG(any, "lx_attn_fwd_split: query segment %d is masked from every key segment", s);
      t += (d->seg_len[s] + 255) / 256;
    }
  }
  a.qt_start[3] = t;
  LX_CHECK_ARG((d->flags & ~(LX_ATTN_Q_LOG2 | LX_ATTN_BOUNDED)) == 0 && (!(d->flags & LX_ATTN_BOUNDED) || (d->flags & LX_ATTN_Q_LOG2)),
               "lx_attn_fwd_split: flags=%d: unknown bit, or LX_ATTN_BOUNDED without LX_ATTN_Q_LOG2", d->flags);
  if (d->flags & LX_ATTN_BOUNDED) hipLaunchKernelGGL(attn_split_kernel<true>, dim3(t * d->B * d->H), dim3(512), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(attn_split_kernel<false>, dim3(t * d->B * d->H), dim3(512), 0, (hipStream_t)stream, a);
  LX_LAUNCH_CHECK("lx_attn_fwd_split");
  return LX_OK;
}
