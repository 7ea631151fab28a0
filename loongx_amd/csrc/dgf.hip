// dgf.hip -- Dynamic Gated Fusion ("DUAN" in the reference, src/train/model.py:947-1035) for gfx950.
// fp32, channel-major [B,C,L].  Five small launches, no atomics (bit-reproducible):
//   1 stats   per (b,c) row: mean/var of x, mean of c                      (reads x, c once)
//   2 gate    per (b, 64-position tile): sigmoid(W2 relu(W1 c + b1) + b2) summed over the tile
//   3 coef    per b: gate mean, layer statistics, gamma/beta MLP -> affine (A, Bc) per channel
//   4 apply   per (b,c) row: y = A x + Bc, importance = mean |y|
//   5 mask    per b: rank-count top-k over channels, zero the dropped rows
#include "common.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* red) {  // 256 threads
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void duan_stats_kernel(const float* __restrict__ x, const float* __restrict__ c,
                                                         float* __restrict__ stats, int C, int L) {
  __shared__ float red[4];
  const int row = blockIdx.y * C + blockIdx.x;
  const float* xr = x + (size_t)row * L;
  const float* cr = c + (size_t)row * L;
  float sx = 0.f, sc = 0.f;
  for (int i = threadIdx.x; i < L; i += 256) { sx += xr[i]; sc += cr[i]; }
  const float mx = block_sum(sx, red) / (float)L;
  const float mc = block_sum(sc, red) / (float)L;
  float q = 0.f;
  for (int i = threadIdx.x; i < L; i += 256) { const float d = xr[i] - mx; q += d * d; }
  const float var = block_sum(q, red) / (float)L;
  if (threadIdx.x == 0) {
    stats[(size_t)row * 4 + 0] = mx;
    stats[(size_t)row * 4 + 1] = var;
    stats[(size_t)row * 4 + 2] = mc;
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

// One workgroup per (64 channels, b) -- 8 x B workgroups instead of B (16 workgroups ran this for 53 us at batch 16). Every workgroup
// recomputes what all channels share (layer statistics in fp64, the hidden layer of the gamma / beta MLP: waves over hidden units,
// lanes over channels, so the weight rows are read coalesced); the per-channel sums keep their order (gate mean over tiles, MLP rows
// over hidden units), so the affine is bit-identical to the one-workgroup form.
__global__ __launch_bounds__(256) void duan_coef_kernel(const float* __restrict__ stats, const float* __restrict__ gpart,
                                                        const float* __restrict__ mw1, const float* __restrict__ mb1,
                                                        const float* __restrict__ mw2, const float* __restrict__ mb2,
                                                        float* __restrict__ coef, int C, int L, int Hd, int ntile, float eps) {
  __shared__ float hid2[128];
  __shared__ float mc[1024];
  __shared__ float wt[128 * 65];          // (>= 64 * 129)
  __shared__ double dred[8];
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* st = stats + (size_t)b * C * 4;
  // layer statistics over (C, L): combine per-row (mean, var) exactly in fp64
  double sm = 0.0;
  for (int ch = tid; ch < C; ch += 256) { sm += (double)st[ch * 4]; mc[ch] = st[ch * 4 + 2]; }
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
  // gamma/beta MLP on the pooled condition: hidden unit hd = relu(b1 + sum over channels IN CHANNEL ORDER) -- one lane walks the
  // channels of its hidden unit (the order of the reference's matmul row is not defined; this keeps the order of the earlier kernel)
  // (weight rows go through LDS in 64-column chunks: read coalesced, then each lane walks ITS row -- stride 65: conflict-free)
  {
    float a = tid < Hd ? mb1[tid] : 0.f;
    for (int c0 = 0; c0 < C; c0 += 64) {
      __syncthreads();
      for (int e = tid; e < Hd * 64; e += 256) {
        const int hd = e >> 6, cc = e & 63;
        wt[hd * 65 + cc] = c0 + cc < C ? mw1[(size_t)hd * C + c0 + cc] : 0.f;
      }
      __syncthreads();
      if (tid < Hd) {
        const int n = min(64, C - c0);
        for (int cc = 0; cc < n; ++cc) a = fmaf(wt[tid * 65 + cc], mc[c0 + cc], a);
      }
    }
    if (tid < Hd) hid2[tid] = a > 0.f ? a : 0.f;
  }
  const int ch = blockIdx.x * 64 + tid;
  float gam = 0.f, bet = 0.f;
  for (int half = 0; half < 2; ++half) {              // gamma rows [0, C), beta rows [C, 2C) of the second MLP layer: 64 rows x Hd at a time
    __syncthreads();
    for (int e = tid; e < 64 * Hd; e += 256) {
      const int r = e / Hd, hd = e - r * Hd, cr = blockIdx.x * 64 + r;
      wt[r * 129 + hd] = cr < C ? mw2[(size_t)(half * C + cr) * Hd + hd] : 0.f;
    }
    __syncthreads();
    if (tid < 64 && ch < C) {
      float a = mb2[half * C + ch];
      for (int hd = 0; hd < Hd; ++hd) a = fmaf(wt[tid * 129 + hd], hid2[hd], a);
      if (half == 0) gam = a; else bet = a;
    }
  }
  if (tid < 64 && ch < C) {
    float g = 0.f;
    for (int t = 0; t < ntile; ++t) g += gpart[((size_t)b * ntile + t) * C + ch];
    g /= (float)L;
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
  return sizeof(float) * ((size_t)B * C * 4 + (size_t)B * ntile * C + (size_t)B * C * 2 + (size_t)B * C + (size_t)B * Hd * L) + 512;
}

extern "C" int lx_duan_fwd(const float* x, const float* c, const float* gw1, const float* gb1, const float* gw2, const float* gb2,
                           const float* mw1, const float* mb1, const float* mw2, const float* mb2, float* y, int B, int C, int L,
                           int Hd, float eps, int keep_k, void* ws, size_t ws_bytes, void* stream) {
  LX_CHECK_ARG(x && c && y && gw1 && gb1 && gw2 && gb2 && mw1 && mb1 && mw2 && mb2 && ws, "lx_duan_fwd: NULL operand");
  LX_CHECK_ARG(B > 0 && C > 0 && C <= 1024 && L > 0 && Hd > 0 && Hd <= 128, "lx_duan_fwd: need C <= 1024 and hidden_dim <= 128 (C=%d Hd=%d)", C, Hd);
  LX_CHECK_ARG(keep_k >= 1 && keep_k <= C, "lx_duan_fwd: keep_k=%d out of [1,%d]", keep_k, C);
  LX_CHECK_ARG(ws_bytes >= lx_duan_workspace_bytes(B, C, L, Hd), "lx_duan_fwd: workspace too small");
  const int ntile = (L + 63) / 64;
  float* stats = (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  float* gpart = stats + (size_t)B * C * 4;
  float* coef = gpart + (size_t)B * ntile * C;
  float* imp = coef + (size_t)B * C * 2;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(duan_stats_kernel, dim3(C, B), dim3(256), 0, s, x, c, stats, C, L);
  if (C % 4 == 0 && Hd % 4 == 0 && L % 4 == 0) {
    // the two 1x1 convolutions of the gate as channel-major fp32 GEMMs on the f32 MFMA (cs3.hip, lx_chan_gemm_f32):
    // hid = relu(W1 c + b1);  gpart[b][tile][ch] = sum over the tile's positions of sigmoid(W2 hid + b2)
    float* hid = (float*)(((uintptr_t)(imp + (size_t)B * C) + 255) & ~(uintptr_t)255);
    int rc = lx_chan_gemm_f32(c, (long)C * L, L, gw1, C, gb1, hid, (long)Hd * L, L, B, Hd, C, L, 2, nullptr, stream);
    if (rc != LX_OK) return rc;
    rc = lx_chan_gemm_f32(hid, (long)Hd * L, L, gw2, Hd, gb2, nullptr, 0, 0, B, C, Hd, L, 3, gpart, stream);
    if (rc != LX_OK) return rc;
  } else {
    hipLaunchKernelGGL(duan_gate_kernel, dim3(ntile, B), dim3(256), 0, s, c, gw1, gb1, gw2, gb2, gpart, C, L, Hd, ntile);
  }
  hipLaunchKernelGGL(duan_coef_kernel, dim3((C + 63) / 64, B), dim3(256), 0, s, stats, gpart, mw1, mb1, mw2, mb2, coef, C, L, Hd, ntile, eps);
  hipLaunchKernelGGL(duan_apply_kernel, dim3(C, B), dim3(256), 0, s, x, coef, y, imp, C, L);
  hipLaunchKernelGGL(duan_mask_kernel, dim3(C, B), dim3(256), 0, s, imp, y, C, L, keep_k);
  LX_LAUNCH_CHECK("lx_duan_fwd");
  return LX_OK;
}
