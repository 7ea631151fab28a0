// rowops.hip -- HBM-bound row kernels of the DiT step (gfx950): AdaLN LayerNorm+modulate, per-head
// RMSNorm+RoPE(+V^T image), LoRA down-projection, skinny (M<=16) linears, timestep embedding, Euler step.
// All loads/stores are 8-16 B per lane and coalesced; reductions are wave64 shuffles (no LDS, no atomics).
#include "common.h"
#include <type_traits>

namespace {

// ------------------------------------------------------------------------------------------------------
// LayerNorm(no affine) + (1+scale)*x + shift, fp32 in -> bf16 out. One wave per row, row kept in VGPRs.
// Reference: AdaLayerNormZero/ZeroSingle/Continuous + norm2 modulation (block.py:192-207,238-253,301,305).
// ------------------------------------------------------------------------------------------------------
struct LnSegs {   // up to 3 row segments (token streams), each with its own modulation table
  int n;
  int row0[3], n_rows[3], rows_per_batch[3];
  const float* shift[3];
  const float* scale[3];
};

// LM (round 5): the rows [lora_row0, lora_row0 + lora_rows) (the streams that run with the adapter on) also get their LoRA down-projection
// T[row - lora_row0, 0..R) = Y_row . Adown[R, D]^T (R <= 16) on the matrix pipe while the workgroup's four normalised rows are at hand:
// the rows go to LDS as the 16-bit operand images they are stored as, Adown is the MFMA "A" operand (rows = r), the four rows the "B"
// operand (columns m = 0..3 of 16, the others zero), the four waves split K and add their partial tiles through LDS -- the same operands
// and instruction as lora_down_mfma_kernel, another split of K (results agree to fp32 summation order). The separate lx_lora_down
// launch over the same rows (6.7 us, 76 of the 132 per denoise step sit behind a LayerNorm) disappears; workgroups without adapter rows
// take the plain path (the decision is workgroup-uniform). Rounds 2-4 had a vector-ALU form of this (576-768 FMAs and 12-16 ds_bpermute
// wave reductions per row): 1.1 % slower per image than the separate launches, never enabled.
struct LnLora { const uint16_t* A; float* T; int R, ldt, row0, rows; };

// F16: Y is the fp16 operand image of an LX_OPERANDS_F16 GEMM (nearest even, saturated to +-65504; *f16_ovf counts the waves that clipped).
template <int NCH, bool LM = false, bool F16 = false>  // D = NCH*256: lane owns float4 chunks lane, lane+64, ...
__global__ __launch_bounds__(256, 3) void ln_modulate_kernel(const float* __restrict__ X, int ldx, const LnSegs segs, int mod_ld,
                                                          uint16_t* __restrict__ Y, int ldy, int M, int D, float eps, const LnLora lo = LnLora{},
                                                          int* __restrict__ f16_ovf = nullptr) {
  constexpr int RS = NCH * 256 + 32;                    // LDS row stride (elements): + 64 B, so the four rows' fragments fall on distinct banks
  __shared__ __attribute__((aligned(16))) uint16_t rows_s[LM ? 4 * RS : 8];
  __shared__ f32x4 red_s[LM ? 4 * 64 : 1];
  const int wave = threadIdx.x >> 6;
  const int row_l = blockIdx.x * 4 + wave;              // row in launch order
  if constexpr (!LM) {
    if (row_l >= M) return;
  }
  const bool valid = row_l < M;
  auto phys = [&](int rl, int& sg_o, int& rin_o) {      // launch-order row -> (segment, row in segment) -> physical row
    int sg = 0, acc_rows = 0;
    while (sg < segs.n - 1 && rl >= acc_rows + segs.n_rows[sg]) { acc_rows += segs.n_rows[sg]; ++sg; }
    sg_o = sg; rin_o = rl - acc_rows;
    return segs.row0[sg] + rin_o;
  };
  int sg, rin;
  const int row = phys(min(row_l, M - 1), sg, rin);
  bool wg_lora = false;                                 // workgroup-uniform: one of its four rows runs the adapter
  if constexpr (LM) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int rl = blockIdx.x * 4 + w;
      int s2, r2;
      const int pr = phys(min(rl, M - 1), s2, r2);
      wg_lora |= rl < M && pr >= lo.row0 && pr < lo.row0 + lo.rows;
    }
  }
  const float* shift = segs.shift[sg];
  const float* scale = segs.scale[sg];
  const int rows_per_batch = segs.rows_per_batch[sg];
  const int lane = threadIdx.x & 63;
  const float* xr = X + (size_t)row * ldx;
  f32x4 v[NCH];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) v[i] = *(const f32x4*)(xr + (i * 64 + lane) * 4);
  // the modulation rows do not depend on the statistics: their loads go out with the row's (one memory round trip instead of two: the
  // kernel has 2.5 workgroups per CU at S = 2560 and is bound by its own latency chain, 11 us per launch in the step, 76 launches)
  const int b = rin / rows_per_batch;
  const float* sh = shift + (size_t)b * mod_ld;
  const float* sc = scale + (size_t)b * mod_ld;
  f32x4 av[NCH], bv[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    av[i] = *(const f32x4*)(sc + (i * 64 + lane) * 4);
    bv[i] = *(const f32x4*)(sh + (i * 64 + lane) * 4);
  }
  // (hipcc sinks most of these loads below the first reduction again -- 144 registers of loads in flight do not fit the 168 that three
  //  waves per SIMD allow without spilling; pinning them with an empty asm was tried: 2-19 spilled registers per variant. What reaches
  //  the memory system early is what fits.)
#pragma unroll
  for (int i = 0; i < NCH; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  const float mean = wave_total(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float d = v[i][c] - mean;
      q += d * d;
    }
  const float rstd = rsqrtf(wave_total(q) / (float)D + eps);
  uint16_t* yr = Y + (size_t)row * ldy;
  float f16_mx = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int col = (i * 64 + lane) * 4;
    float o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = (v[i][c] - mean) * rstd * (1.0f + av[i][c]) + bv[i][c];
    u32x2 w;
    w[0] = pack_op16x2<F16>(o[0], o[1], f16_mx);
    w[1] = pack_op16x2<F16>(o[2], o[3], f16_mx);
    if (valid) *(u32x2*)(yr + col) = w;
    if constexpr (LM) {
      if (wg_lora) *(u32x2*)(rows_s + wave * RS + col) = w;
    }
  }
  if constexpr (LM) {
    if (wg_lora) {
      // K = NCH * 256 in 32-deep MFMA steps; wave w takes steps w, w + 4, ... (2 NCH of them). Adown fragments straight from global
      // memory (L2-resident: every workgroup of the launch reads the same R rows), all in flight before the barrier.
      constexpr int NS = 2 * NCH, NB = NS >= 8 ? NS / 2 : NS;     // two batches of fragments: 176 registers (one batch) would leave two waves
      const int l15 = lane & 15, kq = lane >> 4;                  // per SIMD, and the launch is 2.5 workgroups per CU: a second round
      const uint16_t* ap = lo.A + (size_t)min(l15, lo.R - 1) * D + kq * 8;
      __builtin_amdgcn_sched_barrier(0);          // (hipcc would hoist these loads above the row's arithmetic: 176 registers again)
      bf16x8 af[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) af[u] = *(const bf16x8*)(ap + (4 * u + wave) * 32);
      __syncthreads();
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const uint16_t* xp = rows_s + (l15 & 3) * RS + kq * 8;
      const bf16x8 zero = {};
#pragma unroll
      for (int u0 = 0; u0 < NS; u0 += NB) {
        if (u0 > 0) {
#pragma unroll
          for (int u = 0; u < NB; ++u) af[u] = *(const bf16x8*)(ap + (4 * (u0 + u) + wave) * 32);
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          bf16x8 xf = *(const bf16x8*)(xp + (4 * (u0 + u) + wave) * 32);
          xf = l15 < 4 ? xf : zero;
          if constexpr (F16) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af[u]), __builtin_bit_cast(f16x8, xf), acc, 0, 0, 0);
          else acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u], xf, acc, 0, 0, 0);
        }
      }
      red_s[wave * 64 + lane] = acc;                  // acc[j]: r = 4 kq + j, row m = l15 (< 4 are real)
      __syncthreads();
      if (threadIdx.x < 64) {
        const int m = threadIdx.x >> 4, r = threadIdx.x & 15;
        const float* rp = (const float*)red_s + (m + 16 * (r >> 2)) * 4 + (r & 3);
        const float t = ((rp[0] + rp[256]) + rp[512]) + rp[768];
        const int rl = blockIdx.x * 4 + m;
        int s2, r2;
        const int pr = phys(min(rl, M - 1), s2, r2);
        if (rl < M && r < lo.R && pr >= lo.row0 && pr < lo.row0 + lo.rows) lo.T[(size_t)(pr - lo.row0) * lo.ldt + r] = t;
      }
    }
  }
  if constexpr (F16) report_f16_overflow(f16_mx, f16_ovf);
}

// generic D (multiple of 4): three passes over the (L2-resident) row
template <bool F16 = false>
__global__ __launch_bounds__(256) void ln_modulate_generic(const float* __restrict__ X, int ldx, const LnSegs segs, int mod_ld,
                                                           uint16_t* __restrict__ Y, int ldy, int M, int D, float eps, int* __restrict__ f16_ovf = nullptr) {
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  int sg = 0, acc_rows = 0;
  while (sg < segs.n - 1 && row >= acc_rows + segs.n_rows[sg]) { acc_rows += segs.n_rows[sg]; ++sg; }
  const int rin = row - acc_rows;
  row = segs.row0[sg] + rin;
  const float* shift = segs.shift[sg];
  const float* scale = segs.scale[sg];
  const int rows_per_batch = segs.rows_per_batch[sg];
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
  const int b = rin / rows_per_batch;
  const float* sh = shift + (size_t)b * mod_ld;
  const float* sc = scale + (size_t)b * mod_ld;
  uint16_t* yr = Y + (size_t)row * ldy;
  float f16_mx = 0.f;
  for (int c = lane * 4; c < D; c += 256) {
    const f32x4 v = *(const f32x4*)(xr + c);
    const f32x4 a = *(const f32x4*)(sc + c);
    const f32x4 bsh = *(const f32x4*)(sh + c);
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = (v[k] - mean) * rstd * (1.0f + a[k]) + bsh[k];
    u32x2 w;
    w[0] = pack_op16x2<F16>(o[0], o[1], f16_mx);
    w[1] = pack_op16x2<F16>(o[2], o[3], f16_mx);
    *(u32x2*)(yr + c) = w;
  }
  if constexpr (F16) report_f16_overflow(f16_mx, f16_ovf);
}

// ------------------------------------------------------------------------------------------------------
// Q/K: RMSNorm(128, weight) + RoPE in place; V -> V^T image. Block = 64 positions x 1 head x 1 batch.
// 16 lanes own one (row, head) vector of 128 bf16 (8 elements = 4 rotary pairs each).
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int vt_interleave(int key) {  // within every 16 keys: [0-3, 8-11, 4-7, 12-15]
  return (key & ~15) | (((key >> 2) & 1) << 3) | (((key >> 3) & 1) << 2) | (key & 3);
}

struct QkvSegs {
  int n;
  int row0[3], rows_per_batch[3], vt_pos0[3], tile0[4];
  const float* wq[3]; const float* wk[3]; const float* cos_tab[3]; const float* sin_tab[3];
};

// FAST: every segment has norm_q / norm_k weights and RoPE tables (the DiT case). The launch is one resident generation of waves
// (3840 waves on 1024 SIMDs at S = 2560), so its duration is the length of one thread's dependent chain of memory round trips,
// not bandwidth. The FAST body therefore issues all loads of two of the thread's four rows (q, k, v chunks, RoPE table rows) before
// anything is computed or stored -- the q / k stores go to the buffer the next rows' loads read, so hipcc cannot hoist them
// itself -- without branches in between (at a control-flow join its wait-count pass falls back to vmcnt(0)).
// IN_F16: the projection launch stored IEEE fp16 (an LX_OPERANDS_F16 16-bit store without LX_EPI_QKV: stream lengths the fused epilogue does
// not take, or callers that asked for the separate pass). q / k are read as fp16 and written back in place as bf16 -- what the attention
// kernels read --, V goes to the V^T image rounded fp16 -> bf16. Only the general (non-FAST) body has this form: the path is the rare one.
__device__ __forceinline__ void unpack16x8(const u32x4 raw, float* x, bool in_f16) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (in_f16) {
      x[2 * i] = (float)__builtin_bit_cast(_Float16, (uint16_t)(raw[i] & 0xffffu));
      x[2 * i + 1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(raw[i] >> 16));
    } else {
      x[2 * i] = __uint_as_float(raw[i] << 16);
      x[2 * i + 1] = __uint_as_float(raw[i] & 0xffff0000u);
    }
  }
}
__device__ __forceinline__ u32x4 f16x8_to_bf16x8(const u32x4 raw) {
  float x[8];
  unpack16x8(raw, x, true);
  u32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = pack_bf16x2(x[2 * i], x[2 * i + 1]);
  return o;
}

template <bool FAST, bool IN_F16 = false>
__global__ __launch_bounds__(256) void qkv_prep_kernel(uint16_t* __restrict__ QKV, int ld, int q_col, int k_col, int v_col,
                                                       const QkvSegs segs, float eps, uint16_t* __restrict__ VT, int vt_ld, int H) {
  __shared__ uint16_t vt_s[64][128 + 8];
  int sg = 0;
  while (sg < segs.n - 1 && (int)blockIdx.x >= segs.tile0[sg + 1]) ++sg;
  const int p0 = ((int)blockIdx.x - segs.tile0[sg]) * 64;
  const int row0 = segs.row0[sg], rows_per_batch = segs.rows_per_batch[sg], vt_pos0 = segs.vt_pos0[sg];
  const float* __restrict__ wq = segs.wq[sg];
  const float* __restrict__ wk = segs.wk[sg];
  const float* __restrict__ cos_tab = segs.cos_tab[sg];
  const float* __restrict__ sin_tab = segs.sin_tab[sg];
  const int h = blockIdx.y;
  const int b = blockIdx.z;
  const int tid = threadIdx.x;
  const int sub = tid & 15;        // 16-B chunk within the 128-wide head vector
  const int rloc = tid >> 4;       // 0..15 : row within a 16-row pass
  const size_t rbase = (size_t)row0 + (size_t)b * rows_per_batch;
  if constexpr (FAST) {
    const f32x4 wq0 = *(const f32x4*)(wq + sub * 8), wq1 = *(const f32x4*)(wq + sub * 8 + 4);
    const f32x4 wk0 = *(const f32x4*)(wk + sub * 8), wk1 = *(const f32x4*)(wk + sub * 8 + 4);
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
      uint16_t* rowp[2];
      bool valid[2];
      u32x4 raw[2][3];
      f32x4 cs[2][4];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int p = p0 + (half * 2 + u) * 16 + rloc;
        valid[u] = p < rows_per_batch;
        const int pc = valid[u] ? p : 0;          // row 0 of the batch stands in for rows past the end (loaded, never stored)
        rowp[u] = QKV + (rbase + pc) * ld + h * 128 + sub * 8;
        raw[u][0] = *(const u32x4*)(rowp[u] + q_col);
        raw[u][1] = *(const u32x4*)(rowp[u] + k_col);
        raw[u][2] = *(const u32x4*)(rowp[u] + v_col);
        const float* ct = cos_tab + (size_t)pc * 128 + sub * 8;
        const float* st = sin_tab + (size_t)pc * 128 + sub * 8;
        cs[u][0] = *(const f32x4*)ct; cs[u][1] = *(const f32x4*)(ct + 4);
        cs[u][2] = *(const f32x4*)st; cs[u][3] = *(const f32x4*)(st + 4);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        u32x4 out[2];
#pragma unroll
        for (int which = 0; which < 2; ++which) {   // 0: q, 1: k
          float x[8];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            x[2 * i] = __uint_as_float(raw[u][which][i] << 16);
            x[2 * i + 1] = __uint_as_float(raw[u][which][i] & 0xffff0000u);
          }
          float ss = 0.f;
#pragma unroll
          for (int i = 0; i < 8; ++i) ss += x[i] * x[i];
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
          const float r = rsqrtf(ss * (1.0f / 128.0f) + eps);
          const f32x4 w0 = which ? wk0 : wq0, w1 = which ? wk1 : wq1;
#pragma unroll
          for (int i = 0; i < 4; ++i) { x[i] = x[i] * r * w0[i]; x[4 + i] = x[4 + i] * r * w1[i]; }
          float y[8];
#pragma unroll
          for (int i = 0; i < 4; ++i) {   // pairs (2i, 2i+1): out = x*cos + rot*sin, rot = (-x_odd, x_even)
            const float ce = i < 2 ? cs[u][0][2 * i] : cs[u][1][2 * i - 4], co = i < 2 ? cs[u][0][2 * i + 1] : cs[u][1][2 * i - 3];
            const float se = i < 2 ? cs[u][2][2 * i] : cs[u][3][2 * i - 4], so = i < 2 ? cs[u][2][2 * i + 1] : cs[u][3][2 * i - 3];
            y[2 * i] = x[2 * i] * ce - x[2 * i + 1] * se;
            y[2 * i + 1] = x[2 * i + 1] * co + x[2 * i] * so;
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) out[which][i] = pack_bf16x2(y[2 * i], y[2 * i + 1]);
        }
        if (valid[u]) {
          *(u32x4*)(rowp[u] + q_col) = out[0];
          *(u32x4*)(rowp[u] + k_col) = out[1];
        }
        if (VT) *(u32x4*)&vt_s[(half * 2 + u) * 16 + rloc][sub * 8] = valid[u] ? raw[u][2] : u32x4{0u, 0u, 0u, 0u};
      }
    }
  } else {
#pragma unroll 1
  for (int pass = 0; pass < 4; ++pass) {
    const int p = p0 + pass * 16 + rloc;
    const bool valid = p < rows_per_batch;
    uint16_t* rowp = QKV + (rbase + (valid ? p : 0)) * ld + h * 128 + sub * 8;
    f32x4 c0 = {1.f, 1.f, 1.f, 1.f}, c1 = c0, s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    if (cos_tab && valid) {
      const float* ct = cos_tab + (size_t)p * 128 + sub * 8;
      const float* st = sin_tab + (size_t)p * 128 + sub * 8;
      c0 = *(const f32x4*)ct; c1 = *(const f32x4*)(ct + 4);
      s0 = *(const f32x4*)st; s1 = *(const f32x4*)(st + 4);
    }
#pragma unroll
    for (int which = 0; which < 2; ++which) {   // 0: q, 1: k
      const float* wn = which ? wk : wq;
      uint16_t* ptr = rowp + (which ? k_col : q_col);
      u32x4 raw = {0u, 0u, 0u, 0u};
      if (valid) raw = *(const u32x4*)ptr;
      float x[8];
      unpack16x8(raw, x, IN_F16);
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
        u32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = pack_bf16x2(y[2 * i], y[2 * i + 1]);
        *(u32x4*)ptr = o;
      }
    }
    if (VT) {  // stash V[key][d] for the transpose
      u32x4 raw = {0u, 0u, 0u, 0u};
      if (valid) raw = *(const u32x4*)(rowp + v_col);
      *(u32x4*)&vt_s[pass * 16 + rloc][sub * 8] = IN_F16 ? f16x8_to_bf16x8(raw) : raw;
    }
  }
  }
  if (!VT) return;
  __syncthreads();
  // write V^T rows: thread -> (d, 8 consecutive slots); slot -> source key via the (involutive) interleave
  uint16_t* vtb = VT + ((size_t)(b * H + h) * 128) * vt_ld + vt_pos0 + p0;
  for (int item = tid; item < 128 * 8; item += 256) {
    const int d = item >> 3, g = item & 7;
    uint16_t e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = vt_s[vt_interleave(g * 8 + i)][d];
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (uint32_t)e[2 * i] | ((uint32_t)e[2 * i + 1] << 16);
    *(u32x4*)(vtb + (size_t)d * vt_ld + g * 8) = o;
  }
}

// ------------------------------------------------------------------------------------------------------
// fp8 (OCP e4m3) variant for the fp8 attention path (BASELINE configs[4]): the same RMSNorm + RoPE arithmetic in fp32,
// but q and k go to separate byte images Q8 / K8 [rows, H*128] scaled by q_scale / k_scale, and V to a byte V^T image whose
// 64 keys per tile are ordered for the 32x32x64 f8f6f4 MFMA's B operand: byte j = g*32 + p of a tile row holds key
// (p>>4)*32 + 8*((p&15)>>2) + 4*g + (p&3), i.e. exactly the order in which lane group g of the attention kernel holds its 32
// probabilities (lx_attn_fp8_kernel). The bf16 QKV buffer is left untouched.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float clamp_e4m3(float x) { return fminf(fmaxf(x, -448.f), 448.f); }
__device__ __forceinline__ uint32_t pack_fp8x4(float a, float b, float c, float d) {
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(clamp_e4m3(a), clamp_e4m3(b), 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(clamp_e4m3(c), clamp_e4m3(d), w, true);
  return (uint32_t)w;
}
__device__ __forceinline__ int vt8_key(int j) {   // byte position within a 64-key tile row -> key
  const int g = j >> 5, p = j & 31;
  return (p >> 4) * 32 + 8 * ((p & 15) >> 2) + 4 * g + (p & 3);
}

template <bool IN_F16>
__global__ __launch_bounds__(256) void qkv_prep_fp8_kernel(const uint16_t* __restrict__ QKV, int ld, int q_col, int k_col, int v_col,
                                                           const QkvSegs segs, float eps, uint8_t* __restrict__ Q8,
                                                           uint8_t* __restrict__ K8, int ld8, uint8_t* __restrict__ VT8, int vt_ld,
                                                           int H, float q_scale, float k_scale, float v_scale) {
  __shared__ uint16_t vt_s[64][128 + 8];
  int sg = 0;
  while (sg < segs.n - 1 && (int)blockIdx.x >= segs.tile0[sg + 1]) ++sg;
  const int p0 = ((int)blockIdx.x - segs.tile0[sg]) * 64;
  const int row0 = segs.row0[sg], rows_per_batch = segs.rows_per_batch[sg], vt_pos0 = segs.vt_pos0[sg];
  const float* __restrict__ wq = segs.wq[sg];
  const float* __restrict__ wk = segs.wk[sg];
  const float* __restrict__ cos_tab = segs.cos_tab[sg];
  const float* __restrict__ sin_tab = segs.sin_tab[sg];
  const int h = blockIdx.y;
  const int b = blockIdx.z;
  const int tid = threadIdx.x;
  const int sub = tid & 15;
  const int rloc = tid >> 4;
  const size_t rbase = (size_t)row0 + (size_t)b * rows_per_batch;
#pragma unroll 1
  for (int pass = 0; pass < 4; ++pass) {
    const int p = p0 + pass * 16 + rloc;
    const bool valid = p < rows_per_batch;
    const size_t grow = rbase + (valid ? p : 0);
    const uint16_t* rowp = QKV + grow * ld + h * 128 + sub * 8;
    f32x4 c0 = {1.f, 1.f, 1.f, 1.f}, c1 = c0, s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    if (cos_tab && valid) {
      const float* ct = cos_tab + (size_t)p * 128 + sub * 8;
      const float* st = sin_tab + (size_t)p * 128 + sub * 8;
      c0 = *(const f32x4*)ct; c1 = *(const f32x4*)(ct + 4);
      s0 = *(const f32x4*)st; s1 = *(const f32x4*)(st + 4);
    }
#pragma unroll
    for (int which = 0; which < 2; ++which) {   // 0: q, 1: k
      const float* wn = which ? wk : wq;
      u32x4 raw = {0u, 0u, 0u, 0u};
      if (valid) raw = *(const u32x4*)(rowp + (which ? k_col : q_col));
      float x[8];
      unpack16x8(raw, x, IN_F16);
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
      for (int i = 0; i < 4; ++i) {
        const float ce = i < 2 ? c0[2 * i] : c1[2 * i - 4], co = i < 2 ? c0[2 * i + 1] : c1[2 * i - 3];
        const float se = i < 2 ? s0[2 * i] : s1[2 * i - 4], so = i < 2 ? s0[2 * i + 1] : s1[2 * i - 3];
        y[2 * i] = x[2 * i] * ce - x[2 * i + 1] * se;
        y[2 * i + 1] = x[2 * i + 1] * co + x[2 * i] * so;
      }
      if (valid) {
        const float sc = which ? k_scale : q_scale;
        u32x2 o;
        o[0] = pack_fp8x4(y[0] * sc, y[1] * sc, y[2] * sc, y[3] * sc);
        o[1] = pack_fp8x4(y[4] * sc, y[5] * sc, y[6] * sc, y[7] * sc);
        *(u32x2*)((which ? K8 : Q8) + grow * ld8 + h * 128 + sub * 8) = o;
      }
    }
    u32x4 raw = {0u, 0u, 0u, 0u};
    if (valid) raw = *(const u32x4*)(rowp + v_col);
    *(u32x4*)&vt_s[pass * 16 + rloc][sub * 8] = IN_F16 ? f16x8_to_bf16x8(raw) : raw;
  }
  __syncthreads();
  // V^T byte image: thread -> (d, 16 consecutive byte positions of the 64-key tile row)
  uint8_t* vtb = VT8 + ((size_t)(b * H + h) * 128) * vt_ld + vt_pos0 + p0;
  for (int item = tid; item < 128 * 4; item += 256) {
    const int d = item >> 2, c = item & 3;
    float e[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) e[i] = __uint_as_float((uint32_t)vt_s[vt8_key(c * 16 + i)][d] << 16) * v_scale;
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = pack_fp8x4(e[4 * i], e[4 * i + 1], e[4 * i + 2], e[4 * i + 3]);
    *(u32x4*)(vtb + (size_t)d * vt_ld + c * 16) = o;
  }
}

// ------------------------------------------------------------------------------------------------------
// LoRA down-projection: T[M,R<=16] = X[M,K] . A[R,K]^T (bf16 in, fp32 out) on v_mfma_f32_16x16x32_bf16.
// Workgroup = 8 waves = 16 rows; the waves split K eight ways and reduce through LDS. A is the MFMA "A"
// operand (rows = r), X the "B" operand (cols = m): each lane ends with 4 consecutive r of one row m.
// ------------------------------------------------------------------------------------------------------
// blockIdx.y = slab: in the K-split form (lx_lora_down) slab s covers K range s; in the multi-term form (lx_lora_down_terms,
// precise mode) slab s is a different (X, A) pair over the whole K -- the cross terms x_hi.A, x_lo.A, x_hi.A_lo in ONE launch.
struct LoraTerms {
  const uint16_t* X[4];
  const uint16_t* A[4];
  int ldx[4];
  int n;                  // 0: K-split form (X[0], A[0]); > 0: one slab per term
};

template <bool F16 = false>      // F16: X and A are fp16 images (the operands of an LX_OPERANDS_F16 GEMM), v_mfma_f32_16x16x32_f16
__global__ __launch_bounds__(512) void lora_down_mfma_kernel(const LoraTerms terms, float* __restrict__ T, int ldt, int M, int K, int R, int Ks,
                                                             int split_stride) {
  __shared__ f32x4 red[8][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m0 = blockIdx.x * 16;
  const int term = terms.n > 0 ? (int)blockIdx.y : 0;
  const uint16_t* __restrict__ X = terms.X[term];
  const uint16_t* __restrict__ A = terms.A[term];
  const int ldx = terms.ldx[term];
  const int kbeg = terms.n > 0 ? 0 : blockIdx.y * Ks, kend = terms.n > 0 ? K : min(K, kbeg + Ks);
  T += (size_t)blockIdx.y * split_stride;
  const int l15 = lane & 15, kq = lane >> 4;
  const uint16_t* xp = X + (size_t)min(m0 + l15, M - 1) * ldx + kq * 8;
  const uint16_t* ap = A + (size_t)min(l15, R - 1) * K + kq * 8;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // A wave's K steps are 256 apart. Written as {load, load, wait, MFMA} per step the loop pays the memory latency once per step (7.5 us
  // per launch in round 2: the whole kernel was that chain); round 3 put up to eight steps' loads in flight, in chunks of 8 / 4 / 2 / 1
  // -- still two to four round trips per wave (3 steps at K = 3072 with four K-split slabs, 12 at 12288, 15 at 15360). Round 5: ONE batch
  // of 4, 8 or 16 slots covers a wave's whole range; slots past the end carry zero fragments (their MFMAs add +0: same sums bit for
  // bit as the ascending-k loop), so every launch of the denoise step waits for memory once. The launches are on the step's critical
  // path (tools/ab_engine_attr.py TL_SPLIT 4 1: 1.8 % of the step between one and four K-split slabs).
  int k = kbeg + wave * 32;
  const bf16x8 zero = {};
  auto batch = [&](auto n_) {
    constexpr int NS = decltype(n_)::value;
    bf16x8 af[NS], xf[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const bool in = k + u * 256 < kend;
      af[u] = in ? *(const bf16x8*)(ap + k + u * 256) : zero;
      xf[u] = in ? *(const bf16x8*)(xp + k + u * 256) : zero;
    }
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      if constexpr (F16) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af[u]), __builtin_bit_cast(f16x8, xf[u]), acc, 0, 0, 0);
      else acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u], xf[u], acc, 0, 0, 0);
    }
    k += NS * 256;
  };
  const int nsteps = k < kend ? (kend - k + 255) / 256 : 0;      // (wave-uniform)
  if (nsteps <= 4) batch(std::integral_constant<int, 4>{});
  else if (nsteps <= 8) batch(std::integral_constant<int, 8>{});
  else
    while (k < kend) batch(std::integral_constant<int, 16>{});
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int w = 1; w < 8; ++w) {
      const f32x4 o = red[w][lane];
      acc[0] += o[0]; acc[1] += o[1]; acc[2] += o[2]; acc[3] += o[3];
    }
    // acc[j]: r = 4*kq + j, m = m0 + l15
    const int m = m0 + l15, r0 = 4 * kq;
    if (m < M && r0 < R) {
      float* tp = T + (size_t)m * ldt + r0;
      if (r0 + 4 <= R && (ldt & 3) == 0) *(f32x4*)tp = acc;
      else
        for (int j = 0; j < 4 && r0 + j < R; ++j) tp[j] = acc[j];
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// Skinny linear: Y[M<=16,N] fp32 = act_out(act_in(X[M,K]) . W[N,K]^T + bias). Weight-streaming (HBM bound):
// a wave owns 4 consecutive output columns, lanes split K in 16-B chunks, rows processed 4 at a time.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }

__global__ __launch_bounds__(256) void linear_skinny_kernel(const float* __restrict__ X, int ldx, const uint16_t* __restrict__ W,
                                                            int ldw, const float* __restrict__ bias, float* __restrict__ Y,
                                                            int ldy, int M, int N, int K, int act_in, int act_out, int accumulate) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n0 = (blockIdx.x * 4 + wave) * 4;
  if (n0 >= N) return;
  const uint16_t* wr[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) wr[c] = W + (size_t)min(n0 + c, N - 1) * ldw;
  for (int mb = 0; mb < M; mb += 4) {
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[i][c] = 0.f;
    for (int k = lane * 8; k < K; k += 512) {
      float wv[4][8];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const u32x4 raw = *(const u32x4*)(wr[c] + k);
#pragma unroll
        for (int j = 0; j < 4; ++j) { wv[c][2 * j] = __uint_as_float(raw[j] << 16); wv[c][2 * j + 1] = __uint_as_float(raw[j] & 0xffff0000u); }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (mb + i < M) {
          const float* xp = X + (size_t)(mb + i) * ldx + k;
          const f32x4 x0 = *(const f32x4*)xp, x1 = *(const f32x4*)(xp + 4);
          float xv[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
          if (act_in == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) xv[j] = silu(xv[j]);
          }
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][c] = fmaf(xv[j], wv[c][j], acc[i][c]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float s = wave_sum(acc[i][c]);
        if (lane == 0 && mb + i < M && n0 + c < N) {
          if (bias) s += bias[n0 + c];
          if (act_out == 1) s = silu(s);
          float* yp = Y + (size_t)(mb + i) * ldy + n0 + c;
          *yp = accumulate ? (*yp + s) : s;
        }
      }
  }
}

__global__ void timestep_embed_kernel(const float* __restrict__ t, float* __restrict__ out, int B, int dim) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int half = dim / 2;
  if (i >= B * half) return;
  const int b = i / half, j = i % half;
  const float f = expf(-9.210340371976184f * (float)j / (float)half);  // ln(10000)
  const float a = t[b] * f;
  out[(size_t)b * dim + j] = cosf(a);          // flip_sin_to_cos=True: [cos | sin]
  out[(size_t)b * dim + half + j] = sinf(a);
}

__global__ void euler_kernel(float* __restrict__ x, const void* __restrict__ v, int v_bf16, float ds, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float vv = v_bf16 ? bf16_to_f32(((const uint16_t*)v)[i]) : ((const float*)v)[i];
    x[i] = fmaf(ds, vv, x[i]);
  }
}

__global__ void convert_kernel(void* __restrict__ dst, int dst_bf16, const void* __restrict__ src, int src_bf16, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float v = src_bf16 ? bf16_to_f32(((const uint16_t*)src)[i]) : ((const float*)src)[i];
    if (dst_bf16 == 2) {          // fp16 (saturated): the operand image of an LX_OPERANDS_F16 GEMM
      float mx = 0.f;
      ((uint16_t*)dst)[i] = (uint16_t)(pack_f16x2_sat(v, 0.f, mx) & 0xffffu);
    } else if (dst_bf16) ((uint16_t*)dst)[i] = f32_to_bf16(v);
    else ((float*)dst)[i] = v;
  }
}

}  // namespace

static int ln_launch(const float* X, int ldx, const LnSegs& segs, int mod_ld, void* Y, int ldy, int D, float eps, void* stream,
                     const LnLora* lora = nullptr, bool f16 = false, int* f16_ovf = nullptr) {
  int M = 0;
  for (int i = 0; i < segs.n; ++i) {
    LX_CHECK_ARG(segs.shift[i] && segs.scale[i] && segs.n_rows[i] > 0 && segs.rows_per_batch[i] > 0, "lx_ln_modulate: bad segment %d", i);
    LX_CHECK_ARG((((uintptr_t)segs.shift[i] | (uintptr_t)segs.scale[i]) & 15) == 0, "lx_ln_modulate: misaligned modulation table");
    M += segs.n_rows[i];
  }
  LX_CHECK_ARG(X && Y, "lx_ln_modulate: NULL operand");
  LX_CHECK_ARG(D > 0 && D % 4 == 0 && D <= 16384, "lx_ln_modulate: D=%d must be a multiple of 4 and <= 16384", D);
  LX_CHECK_ARG(ldx % 4 == 0 && ldy % 4 == 0 && mod_ld % 4 == 0, "lx_ln_modulate: ldx/ldy/mod_ld must be multiples of 4");
  LX_CHECK_ARG(((uintptr_t)X & 15) == 0 && ((uintptr_t)Y & 7) == 0, "lx_ln_modulate: misaligned operand");
  const dim3 grid((M + 3) / 4), block(256);
  hipStream_t s = (hipStream_t)stream;
  uint16_t* y = (uint16_t*)Y;
  if (lora) {
    LX_CHECK_ARG(D == 3072 || D == 256, "lx_ln_modulate_lora_segs: the fused down-projection exists for D = 3072 and 256 (D=%d): use lx_lora_down", D);
    LX_CHECK_ARG(lora->A && lora->T && lora->R >= 1 && lora->R <= 16 && lora->ldt >= lora->R && lora->rows > 0 && lora->row0 >= 0,
                 "lx_ln_modulate_lora_segs: bad adapter arguments (R=%d)", lora->R);
    LX_CHECK_ARG(((uintptr_t)lora->A & 15) == 0, "lx_ln_modulate_lora_segs: Adown must be 16-byte aligned");
    if (D == 3072) {
      if (f16) hipLaunchKernelGGL((ln_modulate_kernel<12, true, true>), grid, block, 0, s, X, ldx, segs, mod_ld, y, ldy, M, D, eps, *lora, f16_ovf);
      else hipLaunchKernelGGL((ln_modulate_kernel<12, true, false>), grid, block, 0, s, X, ldx, segs, mod_ld, y, ldy, M, D, eps, *lora, (int*)nullptr);
    } else {
      if (f16) hipLaunchKernelGGL((ln_modulate_kernel<1, true, true>), grid, block, 0, s, X, ldx, segs, mod_ld, y, ldy, M, D, eps, *lora, f16_ovf);
      else hipLaunchKernelGGL((ln_modulate_kernel<1, true, false>), grid, block, 0, s, X, ldx, segs, mod_ld, y, ldy, M, D, eps, *lora, (int*)nullptr);
    }
    LX_LAUNCH_CHECK("lx_ln_modulate_lora_segs");
    return LX_OK;
  }
  if (f16) {
    if (D == 3072) hipLaunchKernelGGL((ln_modulate_kernel<12, false, true>), grid, block, 0, s, X, ldx, segs, mod_ld, y, ldy, M, D, eps, LnLora{}, f16_ovf);
    else if (D == 256) hipLaunchKernelGGL((ln_modulate_kernel<1, false, true>), grid, block, 0, s, X, ldx, segs, mod_ld, y, ldy, M, D, eps, LnLora{}, f16_ovf);
    else hipLaunchKernelGGL(ln_modulate_generic<true>, grid, block, 0, s, X, ldx, segs, mod_ld, y, ldy, M, D, eps, f16_ovf);
    LX_LAUNCH_CHECK("lx_ln_modulate_f16_segs");
    return LX_OK;
  }
  if (D == 3072) hipLaunchKernelGGL((ln_modulate_kernel<12>), grid, block, 0, s, X, ldx, segs, mod_ld, y, ldy, M, D, eps, LnLora{}, (int*)nullptr);
  else if (D == 256) hipLaunchKernelGGL((ln_modulate_kernel<1>), grid, block, 0, s, X, ldx, segs, mod_ld, y, ldy, M, D, eps, LnLora{}, (int*)nullptr);
  else hipLaunchKernelGGL(ln_modulate_generic<false>, grid, block, 0, s, X, ldx, segs, mod_ld, y, ldy, M, D, eps, (int*)nullptr);
  LX_LAUNCH_CHECK("lx_ln_modulate");
  return LX_OK;
}

extern "C" int lx_ln_modulate(const float* X, int ldx, const float* shift, const float* scale, int mod_ld, void* Y, int ldy,
                              int M, int D, int rows_per_batch, float eps, void* stream) {
  LX_CHECK_ARG(M > 0, "lx_ln_modulate: M must be > 0");
  LnSegs segs;
  segs.n = 1;
  segs.row0[0] = 0; segs.n_rows[0] = M; segs.rows_per_batch[0] = rows_per_batch; segs.shift[0] = shift; segs.scale[0] = scale;
  return ln_launch(X, ldx, segs, mod_ld, Y, ldy, D, eps, stream);
}

extern "C" int lx_ln_modulate_segs(const float* X, int ldx, const lx_ln_seg* seg, int n_seg, int mod_ld, void* Y, int ldy, int D,
                                   float eps, void* stream) {
  LX_CHECK_ARG(seg && n_seg >= 1 && n_seg <= 3, "lx_ln_modulate_segs: 1..3 segments");
  LnSegs segs;
  segs.n = n_seg;
  for (int i = 0; i < n_seg; ++i) {
    segs.row0[i] = seg[i].row0; segs.n_rows[i] = seg[i].n_rows; segs.rows_per_batch[i] = seg[i].rows_per_batch;
    segs.shift[i] = seg[i].shift; segs.scale[i] = seg[i].scale;
  }
  return ln_launch(X, ldx, segs, mod_ld, Y, ldy, D, eps, stream);
}

extern "C" int lx_ln_modulate_f16_segs(const float* X, int ldx, const lx_ln_seg* seg, int n_seg, int mod_ld, void* Y, int ldy, int D,
                                       float eps, int32_t* f16_ovf, void* stream) {
  LX_CHECK_ARG(seg && n_seg >= 1 && n_seg <= 3, "lx_ln_modulate_f16_segs: 1..3 segments");
  LX_CHECK_ARG(((uintptr_t)f16_ovf & 3) == 0, "lx_ln_modulate_f16_segs: f16_ovf must be 4-byte aligned");
  LnSegs segs;
  segs.n = n_seg;
  for (int i = 0; i < n_seg; ++i) {
    segs.row0[i] = seg[i].row0; segs.n_rows[i] = seg[i].n_rows; segs.rows_per_batch[i] = seg[i].rows_per_batch;
    segs.shift[i] = seg[i].shift; segs.scale[i] = seg[i].scale;
  }
  return ln_launch(X, ldx, segs, mod_ld, Y, ldy, D, eps, stream, nullptr, true, (int*)f16_ovf);
}

extern "C" int lx_ln_modulate_lora_f16_segs(const float* X, int ldx, const lx_ln_seg* seg, int n_seg, int mod_ld, void* Y, int ldy, int D,
                                            float eps, const void* Adown, int R, float* T, int ldt, int lora_row0, int lora_rows,
                                            int32_t* f16_ovf, void* stream) {
  LX_CHECK_ARG(seg && n_seg >= 1 && n_seg <= 3, "lx_ln_modulate_lora_f16_segs: 1..3 segments");
  LX_CHECK_ARG(((uintptr_t)f16_ovf & 3) == 0, "lx_ln_modulate_lora_f16_segs: f16_ovf must be 4-byte aligned");
  LnSegs segs;
  segs.n = n_seg;
  for (int i = 0; i < n_seg; ++i) {
    segs.row0[i] = seg[i].row0; segs.n_rows[i] = seg[i].n_rows; segs.rows_per_batch[i] = seg[i].rows_per_batch;
    segs.shift[i] = seg[i].shift; segs.scale[i] = seg[i].scale;
  }
  const LnLora lo{(const uint16_t*)Adown, T, R, ldt, lora_row0, lora_rows};
  return ln_launch(X, ldx, segs, mod_ld, Y, ldy, D, eps, stream, &lo, true, (int*)f16_ovf);
}

extern "C" int lx_ln_modulate_lora_segs(const float* X, int ldx, const lx_ln_seg* seg, int n_seg, int mod_ld, void* Y, int ldy, int D,
                                        float eps, const void* Adown, int R, float* T, int ldt, int lora_row0, int lora_rows, void* stream) {
  LX_CHECK_ARG(seg && n_seg >= 1 && n_seg <= 3, "lx_ln_modulate_lora_segs: 1..3 segments");
  LnSegs segs;
  segs.n = n_seg;
  for (int i = 0; i < n_seg; ++i) {
    segs.row0[i] = seg[i].row0; segs.n_rows[i] = seg[i].n_rows; segs.rows_per_batch[i] = seg[i].rows_per_batch;
    segs.shift[i] = seg[i].shift; segs.scale[i] = seg[i].scale;
  }
  const LnLora lo{(const uint16_t*)Adown, T, R, ldt, lora_row0, lora_rows};
  return ln_launch(X, ldx, segs, mod_ld, Y, ldy, D, eps, stream, &lo);
}

static int qkv_launch(void* QKV, int ld, int q_col, int k_col, int v_col, QkvSegs& segs, int n_batches, int H, float eps, void* VT,
                      int vt_ld, void* stream, bool in_f16 = false) {
  LX_CHECK_ARG(QKV && n_batches > 0 && H > 0, "lx_qkv_prep: bad arguments");
  LX_CHECK_ARG(ld % 8 == 0 && q_col % 8 == 0 && k_col % 8 == 0 && v_col % 8 == 0, "lx_qkv_prep: ld and column offsets must be multiples of 8");
  if (VT) LX_CHECK_ARG(vt_ld % 64 == 0, "lx_qkv_prep: vt_ld must be a multiple of 64");
  int t = 0;
  for (int i = 0; i < segs.n; ++i) {
    LX_CHECK_ARG(segs.rows_per_batch[i] > 0, "lx_qkv_prep: empty segment %d", i);
    LX_CHECK_ARG((segs.cos_tab[i] == nullptr) == (segs.sin_tab[i] == nullptr), "lx_qkv_prep: cos/sin tables must come together");
    if (VT) LX_CHECK_ARG(segs.vt_pos0[i] % 64 == 0, "lx_qkv_prep: vt_pos0 must be a multiple of 64");
    segs.tile0[i] = t;
    t += (segs.rows_per_batch[i] + 63) / 64;
  }
  segs.tile0[segs.n] = t;
  bool fast = true;
  for (int i = 0; i < segs.n; ++i) fast = fast && segs.wq[i] && segs.wk[i] && segs.cos_tab[i];
  if (in_f16) hipLaunchKernelGGL((qkv_prep_kernel<false, true>), dim3(t, H, n_batches), dim3(256), 0, (hipStream_t)stream, (uint16_t*)QKV, ld, q_col,
                                 k_col, v_col, segs, eps, (uint16_t*)VT, vt_ld, H);
  else if (fast) hipLaunchKernelGGL(qkv_prep_kernel<true>, dim3(t, H, n_batches), dim3(256), 0, (hipStream_t)stream, (uint16_t*)QKV, ld, q_col, k_col,
                               v_col, segs, eps, (uint16_t*)VT, vt_ld, H);
  else hipLaunchKernelGGL(qkv_prep_kernel<false>, dim3(t, H, n_batches), dim3(256), 0, (hipStream_t)stream, (uint16_t*)QKV, ld, q_col, k_col,
                          v_col, segs, eps, (uint16_t*)VT, vt_ld, H);
  LX_LAUNCH_CHECK("lx_qkv_prep");
  return LX_OK;
}

extern "C" int lx_qkv_prep(void* QKV, int ld, int q_col, int k_col, int v_col, int row0, int n_rows, int rows_per_batch, int H,
                           const float* wq, const float* wk, float eps, const float* cos_tab, const float* sin_tab, void* VT,
                           int vt_ld, int vt_pos0, void* stream) {
  LX_CHECK_ARG(n_rows > 0 && rows_per_batch > 0 && n_rows % rows_per_batch == 0, "lx_qkv_prep: n_rows=%d must be a multiple of rows_per_batch=%d", n_rows, rows_per_batch);
  QkvSegs segs;
  segs.n = 1;
  segs.row0[0] = row0; segs.rows_per_batch[0] = rows_per_batch; segs.vt_pos0[0] = vt_pos0;
  segs.wq[0] = wq; segs.wk[0] = wk; segs.cos_tab[0] = cos_tab; segs.sin_tab[0] = sin_tab;
  return qkv_launch(QKV, ld, q_col, k_col, v_col, segs, n_rows / rows_per_batch, H, eps, VT, vt_ld, stream);
}

static int qkv_segs_launch(bool in_f16, void* QKV, int ld, int q_col, int k_col, int v_col, const lx_qkv_seg* seg, int n_seg, int n_batches,
                           int H, float eps, void* VT, int vt_ld, void* stream) {
  LX_CHECK_ARG(seg && n_seg >= 1 && n_seg <= 3, "lx_qkv_prep_segs: 1..3 segments");
  QkvSegs segs;
  segs.n = n_seg;
  for (int i = 0; i < n_seg; ++i) {
    segs.row0[i] = seg[i].row0; segs.rows_per_batch[i] = seg[i].rows_per_batch; segs.vt_pos0[i] = seg[i].vt_pos0;
    segs.wq[i] = seg[i].wq; segs.wk[i] = seg[i].wk; segs.cos_tab[i] = seg[i].cos_tab; segs.sin_tab[i] = seg[i].sin_tab;
  }
  return qkv_launch(QKV, ld, q_col, k_col, v_col, segs, n_batches, H, eps, VT, vt_ld, stream, in_f16);
}

extern "C" int lx_qkv_prep_segs(void* QKV, int ld, int q_col, int k_col, int v_col, const lx_qkv_seg* seg, int n_seg, int n_batches,
                                int H, float eps, void* VT, int vt_ld, void* stream) {
  return qkv_segs_launch(false, QKV, ld, q_col, k_col, v_col, seg, n_seg, n_batches, H, eps, VT, vt_ld, stream);
}

extern "C" int lx_qkv_prep_f16in_segs(void* QKV, int ld, int q_col, int k_col, int v_col, const lx_qkv_seg* seg, int n_seg, int n_batches,
                                      int H, float eps, void* VT, int vt_ld, void* stream) {
  return qkv_segs_launch(true, QKV, ld, q_col, k_col, v_col, seg, n_seg, n_batches, H, eps, VT, vt_ld, stream);
}

static int qkv_fp8_segs_launch(bool in_f16, const void* QKV, int ld, int q_col, int k_col, int v_col, const lx_qkv_seg* seg, int n_seg,
                               int n_batches, int H, float eps, void* Q8, void* K8, int ld8, void* VT8, int vt8_ld,
                               float q_scale, float k_scale, float v_scale, void* stream) {
  LX_CHECK_ARG(seg && n_seg >= 1 && n_seg <= 3, "lx_qkv_prep_fp8_segs: 1..3 segments");
  LX_CHECK_ARG(QKV && Q8 && K8 && VT8 && n_batches > 0 && H > 0, "lx_qkv_prep_fp8_segs: bad arguments");
  LX_CHECK_ARG(ld % 8 == 0 && q_col % 8 == 0 && k_col % 8 == 0 && v_col % 8 == 0, "lx_qkv_prep_fp8_segs: ld and column offsets must be multiples of 8");
  LX_CHECK_ARG(ld8 % 16 == 0 && vt8_ld % 64 == 0, "lx_qkv_prep_fp8_segs: ld8 %% 16 and vt8_ld %% 64 required");
  LX_CHECK_ARG(q_scale > 0.f && k_scale > 0.f && v_scale > 0.f, "lx_qkv_prep_fp8_segs: scales must be positive");
  QkvSegs segs;
  segs.n = n_seg;
  int t = 0;
  for (int i = 0; i < n_seg; ++i) {
    segs.row0[i] = seg[i].row0; segs.rows_per_batch[i] = seg[i].rows_per_batch; segs.vt_pos0[i] = seg[i].vt_pos0;
    segs.wq[i] = seg[i].wq; segs.wk[i] = seg[i].wk; segs.cos_tab[i] = seg[i].cos_tab; segs.sin_tab[i] = seg[i].sin_tab;
    LX_CHECK_ARG(segs.rows_per_batch[i] > 0, "lx_qkv_prep_fp8_segs: empty segment %d", i);
    LX_CHECK_ARG((segs.cos_tab[i] == nullptr) == (segs.sin_tab[i] == nullptr), "lx_qkv_prep_fp8_segs: cos/sin tables must come together");
    LX_CHECK_ARG(segs.vt_pos0[i] % 64 == 0, "lx_qkv_prep_fp8_segs: vt_pos0 must be a multiple of 64");
    segs.tile0[i] = t;
    t += (segs.rows_per_batch[i] + 63) / 64;
  }
  segs.tile0[n_seg] = t;
  if (in_f16) hipLaunchKernelGGL(qkv_prep_fp8_kernel<true>, dim3(t, H, n_batches), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)QKV, ld, q_col,
                                 k_col, v_col, segs, eps, (uint8_t*)Q8, (uint8_t*)K8, ld8, (uint8_t*)VT8, vt8_ld, H, q_scale, k_scale, v_scale);
  else hipLaunchKernelGGL(qkv_prep_fp8_kernel<false>, dim3(t, H, n_batches), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)QKV, ld, q_col,
                          k_col, v_col, segs, eps, (uint8_t*)Q8, (uint8_t*)K8, ld8, (uint8_t*)VT8, vt8_ld, H, q_scale, k_scale, v_scale);
  LX_LAUNCH_CHECK("lx_qkv_prep_fp8_segs");
  return LX_OK;
}

extern "C" int lx_qkv_prep_fp8_segs(const void* QKV, int ld, int q_col, int k_col, int v_col, const lx_qkv_seg* seg, int n_seg,
                                    int n_batches, int H, float eps, void* Q8, void* K8, int ld8, void* VT8, int vt8_ld,
                                    float q_scale, float k_scale, float v_scale, void* stream) {
  return qkv_fp8_segs_launch(false, QKV, ld, q_col, k_col, v_col, seg, n_seg, n_batches, H, eps, Q8, K8, ld8, VT8, vt8_ld, q_scale, k_scale, v_scale,
                             stream);
}

extern "C" int lx_qkv_prep_fp8_f16in_segs(const void* QKV, int ld, int q_col, int k_col, int v_col, const lx_qkv_seg* seg, int n_seg,
                                          int n_batches, int H, float eps, void* Q8, void* K8, int ld8, void* VT8, int vt8_ld,
                                          float q_scale, float k_scale, float v_scale, void* stream) {
  return qkv_fp8_segs_launch(true, QKV, ld, q_col, k_col, v_col, seg, n_seg, n_batches, H, eps, Q8, K8, ld8, VT8, vt8_ld, q_scale, k_scale, v_scale,
                             stream);
}

static int lora_down_launch(const char* name, bool f16, const void* X, int ldx, const void* Adown, float* T, int ldt, int M, int K, int R, int n_split,
                            int split_stride, void* stream) {
  LX_CHECK_ARG(X && Adown && T && M > 0, "%s: NULL operand", name);
  LX_CHECK_ARG(R >= 1 && R <= 16, "%s: R=%d must be in [1,16]", name, R);
  LX_CHECK_ARG(K % 32 == 0 && ldx % 8 == 0 && ldt >= R, "%s: K %% 32, ldx %% 8 and ldt >= R required (K=%d)", name, K);
  LX_CHECK_ARG(((uintptr_t)X & 15) == 0 && ((uintptr_t)Adown & 15) == 0 && ((uintptr_t)T & 15) == 0, "%s: operands must be 16-byte aligned", name);
  LX_CHECK_ARG(n_split >= 1 && n_split <= 16 && (n_split == 1 || split_stride >= (M - 1) * ldt + R) && split_stride % 4 == 0,
               "%s: bad n_split=%d / split_stride=%d", name, n_split, split_stride);
  const int Ks = ((K / 32 + n_split - 1) / n_split) * 32;
  LoraTerms lt = {};
  lt.X[0] = (const uint16_t*)X; lt.A[0] = (const uint16_t*)Adown; lt.ldx[0] = ldx; lt.n = 0;
  if (f16) hipLaunchKernelGGL(lora_down_mfma_kernel<true>, dim3((M + 15) / 16, n_split), dim3(512), 0, (hipStream_t)stream, lt, T, ldt, M, K, R, Ks, split_stride);
  else hipLaunchKernelGGL(lora_down_mfma_kernel<false>, dim3((M + 15) / 16, n_split), dim3(512), 0, (hipStream_t)stream, lt, T, ldt, M, K, R, Ks, split_stride);
  LX_LAUNCH_CHECK(name);
  return LX_OK;
}

extern "C" int lx_lora_down(const void* X, int ldx, const void* Adown, float* T, int ldt, int M, int K, int R, int n_split,
                            int split_stride, void* stream) {
  return lora_down_launch("lx_lora_down", false, X, ldx, Adown, T, ldt, M, K, R, n_split, split_stride, stream);
}

extern "C" int lx_lora_down_f16(const void* X, int ldx, const void* Adown, float* T, int ldt, int M, int K, int R, int n_split,
                                int split_stride, void* stream) {
  return lora_down_launch("lx_lora_down_f16", true, X, ldx, Adown, T, ldt, M, K, R, n_split, split_stride, stream);
}

extern "C" int lx_lora_down_terms(const void* const* X, const int* ldx, const void* const* Adown, int n_terms, float* T, int ldt, int M, int K,
                                  int R, int slab_stride, void* stream) {
  LX_CHECK_ARG(X && ldx && Adown && T && M > 0 && n_terms >= 1 && n_terms <= 4, "lx_lora_down_terms: bad arguments (1..4 terms)");
  LX_CHECK_ARG(R >= 1 && R <= 16 && K % 32 == 0 && ldt >= R && ((uintptr_t)T & 15) == 0, "lx_lora_down_terms: R in [1,16], K %% 32, ldt >= R, T 16-byte aligned required");
  LX_CHECK_ARG((n_terms == 1 || slab_stride >= (M - 1) * ldt + R) && slab_stride % 4 == 0, "lx_lora_down_terms: bad slab_stride=%d", slab_stride);
  LoraTerms lt = {};
  lt.n = n_terms;
  for (int i = 0; i < n_terms; ++i) {
    LX_CHECK_ARG(X[i] && Adown[i] && ldx[i] % 8 == 0 && ((uintptr_t)X[i] & 15) == 0 && ((uintptr_t)Adown[i] & 15) == 0, "lx_lora_down_terms: term %d: NULL / misaligned operand or ldx %% 8", i);
    lt.X[i] = (const uint16_t*)X[i]; lt.A[i] = (const uint16_t*)Adown[i]; lt.ldx[i] = ldx[i];
  }
  hipLaunchKernelGGL(lora_down_mfma_kernel<false>, dim3((M + 15) / 16, n_terms), dim3(512), 0, (hipStream_t)stream, lt, T, ldt, M, K, R, K, slab_stride);
  LX_LAUNCH_CHECK("lx_lora_down_terms");
  return LX_OK;
}

extern "C" int lx_linear_skinny(const float* X, int ldx, const void* W, int ldw, const float* bias, float* Y, int ldy, int M, int N,
                                int K, int act_in, int act_out, int accumulate, void* stream) {
  LX_CHECK_ARG(X && W && Y, "lx_linear_skinny: NULL operand");
  LX_CHECK_ARG(M >= 1 && M <= 65536, "lx_linear_skinny: M=%d must be in [1,65536]", M);
  LX_CHECK_ARG(K % 8 == 0 && ldw % 8 == 0 && ldx % 4 == 0, "lx_linear_skinny: K %% 8, ldw %% 8, ldx %% 4 required (K=%d)", K);
  const dim3 grid((N + 15) / 16), block(256);
  hipLaunchKernelGGL(linear_skinny_kernel, grid, block, 0, (hipStream_t)stream, X, ldx, (const uint16_t*)W, ldw, bias, Y, ldy, M, N,
                     K, act_in, act_out, accumulate);
  LX_LAUNCH_CHECK("lx_linear_skinny");
  return LX_OK;
}

extern "C" int lx_timestep_embed(const float* t, float* out, int B, int dim, void* stream) {
  LX_CHECK_ARG(t && out && B > 0 && dim > 0 && dim % 2 == 0, "lx_timestep_embed: bad arguments");
  const int n = B * dim / 2;
  hipLaunchKernelGGL(timestep_embed_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, out, B, dim);
  LX_LAUNCH_CHECK("lx_timestep_embed");
  return LX_OK;
}

extern "C" int lx_euler_step(float* x, const void* v, int v_is_bf16, float dsigma, size_t n, void* stream) {
  LX_CHECK_ARG(x && v, "lx_euler_step: NULL operand");
  if (n == 0) return LX_OK;
  const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(euler_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, v, v_is_bf16, dsigma, n);
  LX_LAUNCH_CHECK("lx_euler_step");
  return LX_OK;
}

extern "C" int lx_convert(void* dst, int dst_bf16, const void* src, int src_bf16, size_t n, void* stream) {
  LX_CHECK_ARG(dst && src, "lx_convert: NULL operand");
  if (n == 0) return LX_OK;
  const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(convert_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dst, dst_bf16, src, src_bf16, n);
  LX_LAUNCH_CHECK("lx_convert");
  return LX_OK;
}

// ------------------------------------------------------------------------------------------------------
// RoPE tables (diffusers FluxPosEmbed / get_1d_rotary_pos_embed, transformer.py:130-134): fp64 frequencies,
// repeat-interleaved real layout, stored fp32. One thread per (position, rotary pair).
// ------------------------------------------------------------------------------------------------------
namespace {
__global__ void rope_table_kernel(const float* __restrict__ ids, int n_axes, int a0, int a1, int a2, double theta,
                                  float* __restrict__ cos_t, float* __restrict__ sin_t, int L) {
  const int dims[3] = {a0, a1, a2};
  const int total = a0 + a1 + a2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L * (total / 2)) return;
  const int p = i / (total / 2);
  int j = i % (total / 2), axis = 0, off = 0;
  while (axis < n_axes - 1 && j >= dims[axis] / 2) { j -= dims[axis] / 2; off += dims[axis]; ++axis; }
  const double freq = 1.0 / pow(theta, (double)(2 * j) / (double)dims[axis]);
  const double ang = (double)ids[(size_t)p * n_axes + axis] * freq;
  const float c = (float)cos(ang), s = (float)sin(ang);
  const size_t o = (size_t)p * total + off + 2 * j;
  cos_t[o] = c; cos_t[o + 1] = c;
  sin_t[o] = s; sin_t[o + 1] = s;
}
}  // namespace

extern "C" int lx_rope_table(const float* ids, int L, int a0, int a1, int a2, double theta, float* cos_t, float* sin_t, void* stream) {
  LX_CHECK_ARG(ids && cos_t && sin_t && L > 0, "lx_rope_table: bad arguments");
  LX_CHECK_ARG(a0 > 0 && a1 > 0 && a2 > 0 && a0 % 2 == 0 && a1 % 2 == 0 && a2 % 2 == 0, "lx_rope_table: axes dims must be positive and even");
  const int n = L * (a0 + a1 + a2) / 2;
  hipLaunchKernelGGL(rope_table_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, ids, 3, a0, a1, a2, theta, cos_t, sin_t, L);
  LX_LAUNCH_CHECK("lx_rope_table");
  return LX_OK;
}
