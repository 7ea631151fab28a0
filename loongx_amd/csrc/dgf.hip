// dgf.hip -- Dynamic Gated Fusion ("DUAN" in the reference, src/train/model.py:947-1035) for gfx950.
// fp32, channel-major [B,C,L].  Five small launches, no atomics (bit-reproducible):
//   1 stats   per (b,c) row: mean/var of x, mean of c                      (reads x, c once)
//   2 gate    per (b, 64-position tile): sigmoid(W2 relu(W1 c + b1) + b2) summed over the tile
//   3 coef    per b: gate mean, layer statistics, gamma/beta MLP -> affine (A, Bc) per channel
//   4 apply   per (b,c) row: y = A x + Bc, importance = mean |y|
//   5 mask    per b: rank-count top-k over channels, zero the dropped rows
#include "common.h"

int lx_chan_gemm_split(const float* X, long x_bstride, int ldx, const float* W, int ldw, const float* bias, float* Y, long y_bstride, int ldy,
                       int B, int N, int K, int L, int epilogue, float* part, void* stream);       // cs3.hip
int lx_chan_gemm_wide(const float* X, long x_bstride, int ldx, const uint16_t* Wh, const uint16_t* Wl, const float* bias, float* Y, long y_bstride,
                      int ldy, int B, int N, int K, int L, int epilogue, float* part, float* xsum, int tile, void* stream);        // cs3.hip
int lx_chan_gemm_wide_tile(int B, int L);
int lx_split_bf16_pair(const float* a, uint16_t* ah, uint16_t* al, int na, const float* b, uint16_t* bh, uint16_t* bl, int nb, void* stream);

namespace {

__device__ __forceinline__ float block_sum(float v, float* red) {  // 256 threads
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

template <bool WITH_C>
__global__ __launch_bounds__(256) void duan_stats_kernel(const float* __restrict__ x, const float* __restrict__ c,
                                                         float* __restrict__ stats, int C, int L) {
  // WITH_C = false: the condition's per-channel mean comes out of the gate's first GEMM as per-tile sums (cpart); this kernel reads x alone.
  // Rows of up to 4096 elements (L % 4 == 0, 16-byte aligned) are held in registers: one pass over memory, all loads in flight at once,
  // the variance from the registers (same two-pass arithmetic: mean first, then squared deviations).
  __shared__ float red[4];
  const int row = blockIdx.y * C + blockIdx.x;
  const float* xr = x + (size_t)row * L;
  const float* cr = c + (size_t)row * L;
  float sx = 0.f, sc = 0.f, mx, var;
  if (L <= 4096 && (L & 3) == 0 && (((uintptr_t)xr | (uintptr_t)cr) & 15) == 0) {
    f32x4 xv[4], cv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = (u * 256 + threadIdx.x) * 4;
      xv[u] = i < L ? *(const f32x4*)(xr + i) : f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (WITH_C) cv[u] = i < L ? *(const f32x4*)(cr + i) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      sx += (xv[u][0] + xv[u][1]) + (xv[u][2] + xv[u][3]);
      if constexpr (WITH_C) sc += (cv[u][0] + cv[u][1]) + (cv[u][2] + cv[u][3]);
    }
    mx = block_sum(sx, red) / (float)L;
    float q = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = (u * 256 + threadIdx.x) * 4;
      if (i < L) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = xv[u][e] - mx; q += d * d; }
      }
    }
    var = block_sum(q, red) / (float)L;
  } else {
    for (int i = threadIdx.x; i < L; i += 256) {
      sx += xr[i];
      if constexpr (WITH_C) sc += cr[i];
    }
    mx = block_sum(sx, red) / (float)L;
    float q = 0.f;
    for (int i = threadIdx.x; i < L; i += 256) { const float d = xr[i] - mx; q += d * d; }
    var = block_sum(q, red) / (float)L;
  }
  float mc = 0.f;
  if constexpr (WITH_C) mc = block_sum(sc, red) / (float)L;
  if (threadIdx.x == 0) {
    stats[(size_t)row * 4 + 0] = mx;
    stats[(size_t)row * 4 + 1] = var;
    if constexpr (WITH_C) stats[(size_t)row * 4 + 2] = mc;
  }
}

// gate network on a 64-position tile. hidden[Hd][64] lives in LDS. Hd <= 128.
__global__ __launch_bounds__(256) void duan_gate_kernel(const float* __restrict__ c, const float* __restrict__ w1,
                                                        const float* __restrict__ b1, const float* __restrict__ w2,
                                                        const float* __restrict__ b2, float* __restrict__ gpart, int C, int L,
                                                        int Hd, int ntile) {
  __shared__ float hid[128][64];
  const int b = blockIdx.y, tile = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l = tile * 64 + lane;
  const bool lv = l < L;
  const float* cb = c + (size_t)b * C * L;
  // phase 1: wave g computes hidden units g*32 .. g*32+31 for its 64 positions
  float acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = 0.f;
  for (int ch = 0; ch < C; ++ch) {
    const float cv = lv ? cb[(size_t)ch * L + l] : 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int hd = wave * 32 + i;
      if (hd < Hd) acc[i] = fmaf(w1[(size_t)hd * C + ch], cv, acc[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const int hd = wave * 32 + i;
    if (hd < Hd) { const float v = acc[i] + b1[hd]; hid[hd][lane] = v > 0.f ? v : 0.f; }
  }
  __syncthreads();
  // phase 2: wave g handles channels g, g+4, ...; sum sigmoid over the tile's valid positions
  for (int ch = wave; ch < C; ch += 4) {
    float a = b2[ch];
    for (int hd = 0; hd < Hd; ++hd) a = fmaf(w2[(size_t)ch * Hd + hd], hid[hd][lane], a);
    float s = lv ? 1.0f / (1.0f + __expf(-a)) : 0.f;
    s = wave_sum(s);
    if (lane == 0) gpart[((size_t)b * ntile + tile) * C + ch] = s;
  }
}

// One workgroup per (64 channels, b). Every workgroup recomputes what all channels share (layer statistics in fp64, the hidden layer of
// the gamma / beta MLP). Round 5: every dot product is a WAVE's -- lanes along the contraction, coalesced weight rows straight from
// global memory, all loads of a row in flight at once, one shuffle reduction -- and the gate's mean over the L / 64 position tiles is
// summed by four groups of lanes in parallel. The earlier form walked each contraction on one lane (64-128 dependent steps, and 64
// dependent global loads for the tile sums): 152 us per call at C = 512, L = 4096 on 128 workgroups, the largest kernel of the DUAN.
// Sums are in a fixed order (tree per wave, then ascending): deterministic, batch-independent.
__global__ __launch_bounds__(256) void duan_coef_kernel(const float* __restrict__ stats, const float* __restrict__ gpart,
                                                        const float* __restrict__ mw1, const float* __restrict__ mb1,
                                                        const float* __restrict__ mw2, const float* __restrict__ mb2,
                                                        float* __restrict__ coef, int C, int L, int Hd, int ntile, float eps,
                                                        const float* __restrict__ cpart, int nctile) {
  __shared__ float hid2[128];
  __shared__ __attribute__((aligned(16))) float mc[1024];
  __shared__ float gb[2][64];              // gamma / beta of this workgroup's 64 channels
  __shared__ float gsum[4][64];
  __shared__ double dred[8];
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* st = stats + (size_t)b * C * 4;
  // layer statistics over (C, L): combine per-row (mean, var) exactly in fp64
  double sm = 0.0;
  for (int ch = tid; ch < C; ch += 256) {
    sm += (double)st[ch * 4];
    if (cpart) {                                       // the condition's mean from the gate GEMM's per-tile sums, ascending tile order
      float a = 0.f;
      for (int tb = 0; tb < nctile; tb += 64) {         // 64 tiles' rows in flight: the kernel is a chain of memory round trips
        float v[64];
#pragma unroll
        for (int u = 0; u < 64; ++u) v[u] = tb + u < nctile ? cpart[((size_t)b * nctile + tb + u) * C + ch] : 0.f;
#pragma unroll
        for (int u = 0; u < 64; ++u) a += v[u];
      }
      mc[ch] = a / (float)L;
    } else {
      mc[ch] = st[ch * 4 + 2];
    }
  }
  for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
  if (lane == 0) dred[wave] = sm;
  __syncthreads();
  const double mu_l = (dred[0] + dred[1] + dred[2] + dred[3]) / (double)C;
  double sv = 0.0;
  for (int ch = tid; ch < C; ch += 256) {
    const double d = (double)st[ch * 4] - mu_l;
    sv += (double)st[ch * 4 + 1] + d * d;
  }
  for (int o = 32; o > 0; o >>= 1) sv += __shfl_xor(sv, o, 64);
  if (lane == 0) dred[4 + wave] = sv;
  __syncthreads();
  const double var_l = (dred[4] + dred[5] + dred[6] + dred[7]) / (double)C;
  const float mul = (float)mu_l, sig_l = sqrtf((float)var_l + eps);
  // gamma / beta MLP, layer 1: hidden unit hd = relu(b1 + W1[hd, :] . pooled condition); wave w takes hd = w, w + 4, ...
  // (four rows x up to 16 column chunks = 64 loads of a lane in flight together: the kernel is a chain of memory round trips otherwise)
  if ((C & 255) == 0 && (((uintptr_t)mw1) & 15) == 0) {
    // rows of C = 256 q floats: q 16-byte loads per lane and row; eight rows (<= 32 loads) in flight per batch
    const int nq = C >> 8;                              // <= 4 (C <= 1024)
    for (int i0 = 0; wave + 4 * i0 < Hd; i0 += 8) {
      f32x4 v[8][4];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int hd = wave + 4 * (i0 + u);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          v[u][q] = (hd < Hd && q < nq) ? *(const f32x4*)(mw1 + (size_t)hd * C + (q * 64 + lane) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int hd = wave + 4 * (i0 + u);
        float a = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (q < nq) {
            const f32x4 m = *(const f32x4*)(mc + (q * 64 + lane) * 4);
            a = fmaf(v[u][q][0], m[0], a); a = fmaf(v[u][q][1], m[1], a); a = fmaf(v[u][q][2], m[2], a); a = fmaf(v[u][q][3], m[3], a);
          }
        }
        a = wave_total(a);
        if (lane == 0 && hd < Hd) { a += mb1[hd]; hid2[hd] = a > 0.f ? a : 0.f; }
      }
    }
  } else {
  for (int i0 = 0; wave + 4 * i0 < Hd; i0 += 4) {
    float v[4][16];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int hd = wave + 4 * (i0 + u);
#pragma unroll
      for (int q = 0; q < 16; ++q) v[u][q] = (hd < Hd && lane + 64 * q < C) ? mw1[(size_t)hd * C + lane + 64 * q] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int hd = wave + 4 * (i0 + u);
      float a = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) a = fmaf(v[u][q], lane + 64 * q < C ? mc[lane + 64 * q] : 0.f, a);
      a = wave_total(a);
      if (lane == 0 && hd < Hd) { a += mb1[hd]; hid2[hd] = a > 0.f ? a : 0.f; }
    }
  }
  }
  __syncthreads();
  // layer 2: rows [0, C) = gamma, [C, 2C) = beta; the 128 rows of this workgroup's 64 channels, 32 per wave, lanes along the hidden units
  for (int i0 = 0; i0 < 32; i0 += 32) {               // rows r = wave + 4 i: thirty-two rows x two hidden-unit chunks in flight
    float v[32][2];
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const int r = wave + 4 * (i0 + u), half = r >> 6, cr = blockIdx.x * 64 + (r & 63);
#pragma unroll
      for (int q = 0; q < 2; ++q) v[u][q] = (cr < C && lane + 64 * q < Hd) ? mw2[(size_t)(half * C + cr) * Hd + lane + 64 * q] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const int r = wave + 4 * (i0 + u), half = r >> 6, cr = blockIdx.x * 64 + (r & 63);
      float a = fmaf(v[u][1], lane + 64 < Hd ? hid2[lane + 64] : 0.f, v[u][0] * (lane < Hd ? hid2[lane] : 0.f));
      a = wave_total(a);
      if (lane == 0) gb[half][r & 63] = cr < C ? a + mb2[half * C + cr] : 0.f;
    }
  }
  // the gate's mean over L: group q = wave sums its quarter of the tiles for channel `lane` (coalesced rows of gpart), then q = 0..3 in order
  const int ch = blockIdx.x * 64 + lane;
  {
    const int per = (ntile + 3) / 4, t0 = wave * per, t1 = min(ntile, t0 + per);
    float g = 0.f;
    for (int tb = t0; tb < t1; tb += 16) {             // sixteen tiles' rows in flight, added in ascending tile order
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = (ch < C && tb + u < t1) ? gpart[((size_t)b * ntile + tb + u) * C + ch] : 0.f;
#pragma unroll
      for (int u = 0; u < 16; ++u) g += v[u];
    }
    gsum[wave][lane] = g;
  }
  __syncthreads();
  if (tid < 64 && ch < C) {
    const float g = (((gsum[0][tid] + gsum[1][tid]) + gsum[2][tid]) + gsum[3][tid]) / (float)L;
    const float gam = gb[0][tid], bet = gb[1][tid];
    const float mu = g * st[ch * 4] + (1.f - g) * mul;
    const float sig = g * sqrtf(st[ch * 4 + 1] + eps) + (1.f - g) * sig_l;
    const float A = (1.f + gam) / sig;
    coef[((size_t)b * C + ch) * 2] = A;
    coef[((size_t)b * C + ch) * 2 + 1] = bet - A * mu;
  }
}

__global__ __launch_bounds__(256) void duan_apply_kernel(const float* __restrict__ x, const float* __restrict__ coef,
                                                         float* __restrict__ y, float* __restrict__ imp, int C, int L) {
  __shared__ float red[4];
  const int row = blockIdx.y * C + blockIdx.x;
  const float A = coef[(size_t)row * 2], Bc = coef[(size_t)row * 2 + 1];
  const float* xr = x + (size_t)row * L;
  float* yr = y + (size_t)row * L;
  float s = 0.f;
  for (int i = threadIdx.x; i < L; i += 256) {
    const float v = fmaf(A, xr[i], Bc);
    yr[i] = v;
    s += fabsf(v);
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) imp[row] = s / (float)L;
}

// One workgroup per (channel, b): its rank among the batch element's importances (ties: the lower channel index first, as a stable
// descending sort keeps them), and the row zeroed when the rank is past keep_k -- C x B workgroups instead of B walking the dropped
// rows one after the other (64 us at batch 16).
__global__ __launch_bounds__(256) void duan_mask_kernel(const float* __restrict__ imp, float* __restrict__ y, int C, int L, int keep_k) {
  __shared__ float red[4];
  const int b = blockIdx.y, ch = blockIdx.x;
  const float* ib = imp + (size_t)b * C;
  const float v = ib[ch];
  float cnt = 0.f;
  for (int o = threadIdx.x; o < C; o += 256) {
    const float w = ib[o];
    cnt += ((w > v) || (w == v && o < ch)) ? 1.f : 0.f;
  }
  const int rank = (int)(block_sum(cnt, red) + 0.5f);      // (counts <= 1024: exact in fp32)
  if (rank < keep_k) return;
  float* yr = y + ((size_t)b * C + ch) * L;
  if ((L & 3) == 0 && (((uintptr_t)yr) & 15) == 0) {
    for (int i = threadIdx.x * 4; i < L; i += 1024) *(f32x4*)(yr + i) = f32x4{0.f, 0.f, 0.f, 0.f};
  } else {
    for (int i = threadIdx.x; i < L; i += 256) yr[i] = 0.f;
  }
}

}  // namespace

extern "C" size_t lx_duan_workspace_bytes(int B, int C, int L, int Hd) {
  const size_t ntile = (size_t)(L + 63) / 64;
  // + the gate network's hidden activations [B, Hd, L] for the MFMA form of the gate (C % 4 == 0)
  // + bf16 hi / lo images of the gate's two weight matrices and the condition's per-tile channel sums (the wide GEMM form)
  return sizeof(float) * ((size_t)B * C * 4 + (size_t)B * ntile * C + (size_t)B * C * 2 + (size_t)B * C + (size_t)B * Hd * L + (size_t)B * ntile * C) + 8 * (size_t)Hd * C + 2048;
}

extern "C" int lx_duan_fwd(const float* x, const float* c, const float* gw1, const float* gb1, const float* gw2, const float* gb2,
                           const float* mw1, const float* mb1, const float* mw2, const float* mb2, float* y, int B, int C, int L,
                           int Hd, float eps, int keep_k, void* ws, size_t ws_bytes, void* stream) {
  LX_CHECK_ARG(x && c && y && gw1 && gb1 && gw2 && gb2 && mw1 && mb1 && mw2 && mb2 && ws, "lx_duan_fwd: NULL operand");
  LX_CHECK_ARG(B > 0 && C > 0 && C <= 1024 && L > 0 && Hd > 0 && Hd <= 128, "lx_duan_fwd: need C <= 1024 and hidden_dim <= 128 (C=%d Hd=%d)", C, Hd);
  LX_CHECK_ARG(keep_k >= 1 && keep_k <= C, "lx_duan_fwd: keep_k=%d out of [1,%d]", keep_k, C);
  LX_CHECK_ARG(ws_bytes >= lx_duan_workspace_bytes(B, C, L, Hd), "lx_duan_fwd: workspace too small");
  const bool mfma_gate = C % 4 == 0 && Hd % 4 == 0 && L % 4 == 0;
  const bool wide = mfma_gate && Hd % 128 == 0 && C % 128 == 0 && C > 128 && (((uintptr_t)gb1 | (uintptr_t)gb2) & 15) == 0;
  const int tile = wide ? lx_chan_gemm_wide_tile(B, L) : 64;      // positions per workgroup of the gate's second GEMM (a launch-shape choice:
  const int ntile = (L + 63) / 64, nctile = ntile;                //  gpart / cpart hold one row per 64 positions either way)
  float* stats = (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  float* gpart = stats + (size_t)B * C * 4;
  float* coef = gpart + (size_t)B * ntile * C;
  float* imp = coef + (size_t)B * C * 2;
  hipStream_t s = (hipStream_t)stream;
  float* cpart = nullptr;
  if (wide) hipLaunchKernelGGL(duan_stats_kernel<false>, dim3(C, B), dim3(256), 0, s, x, c, stats, C, L);
  else hipLaunchKernelGGL(duan_stats_kernel<true>, dim3(C, B), dim3(256), 0, s, x, c, stats, C, L);
  if (mfma_gate) {
    // the two 1x1 convolutions of the gate as channel-major GEMMs on the bf16 matrix pipe, every operand a split-bf16 pair (cs3.hip):
    // hid = relu(W1 c + b1);  gpart[b][tile][ch] = sum over the tile's positions of sigmoid(W2 hid + b2)
    float* hid = (float*)(((uintptr_t)(imp + (size_t)B * C) + 255) & ~(uintptr_t)255);
    // (2^-16 per product at 16/3 of the exact-fp32 MFMA's rate; the gate feeds a mean over L of sigmoids: the kept-channel sets of the
    //  goldens and of the full-size case are unchanged. Rounds 2-4 ran two exact-fp32 lx_chan_gemm_f32 launches here: 2 x 150 us; the
    //  split-bf16 tile kernel 2 x 84 us; the wide form below 52 + 57 us. A FUSED form -- one workgroup per 128 positions, the hidden layer
    //  kept in LDS as bf16 pairs, W1 / W2 staged per workgroup -- was built and measured in round 5: 195-242 us per call; its 384 MFMAs per
    //  wave are 6 us of a ~100-us workgroup, the rest is staging 512 KB of weights and 32 KB of c per workgroup with one workgroup per CU.)
    int rc;
    if (wide) {
      // wide form: a workgroup per 64 positions and ALL output rows, X split once per workgroup, W from images split once per call;
      // the first GEMM also leaves the condition's per-tile channel sums (its mean over L is what the gamma / beta MLP takes)
      cpart = (float*)(((uintptr_t)(hid + (size_t)B * Hd * L) + 255) & ~(uintptr_t)255);
      uint16_t* w1h = (uint16_t*)(((uintptr_t)(cpart + (size_t)B * nctile * C) + 255) & ~(uintptr_t)255);
      uint16_t *w1l = w1h + (size_t)Hd * C, *w2h = w1l + (size_t)Hd * C, *w2l = w2h + (size_t)Hd * C;
      rc = lx_split_bf16_pair(gw1, w1h, w1l, Hd * C, gw2, w2h, w2l, Hd * C, stream);
      if (rc != LX_OK) return rc;
      rc = lx_chan_gemm_wide(c, (long)C * L, L, w1h, w1l, gb1, hid, (long)Hd * L, L, B, Hd, C, L, 2, nullptr, cpart, 64, stream);
      if (rc != LX_OK) return rc;
      rc = lx_chan_gemm_wide(hid, (long)Hd * L, L, w2h, w2l, gb2, nullptr, 0, 0, B, C, Hd, L, 3, gpart, nullptr, tile, stream);
      if (rc != LX_OK) return rc;
    } else {
      rc = lx_chan_gemm_split(c, (long)C * L, L, gw1, C, gb1, hid, (long)Hd * L, L, B, Hd, C, L, 2, nullptr, stream);
      if (rc != LX_OK) return rc;
      rc = lx_chan_gemm_split(hid, (long)Hd * L, L, gw2, Hd, gb2, nullptr, 0, 0, B, C, Hd, L, 3, gpart, stream);
      if (rc != LX_OK) return rc;
    }
  } else {
    hipLaunchKernelGGL(duan_gate_kernel, dim3(ntile, B), dim3(256), 0, s, c, gw1, gb1, gw2, gb2, gpart, C, L, Hd, ntile);
  }
  hipLaunchKernelGGL(duan_coef_kernel, dim3((C + 63) / 64, B), dim3(256), 0, s, stats, gpart, mw1, mb1, mw2, mb2, coef, C, L, Hd, ntile, eps, cpart, nctile);
  hipLaunchKernelGGL(duan_apply_kernel, dim3(C, B), dim3(256), 0, s, x, coef, y, imp, C, L);
  hipLaunchKernelGGL(duan_mask_kernel, dim3(C, B), dim3(256), 0, s, imp, y, C, L, keep_k);
  LX_LAUNCH_CHECK("lx_duan_fwd");
  return LX_OK;
}
