// vae.hip -- row kernels of the FLUX VAE (diffusers AutoencoderKL; reference call sites src/flux/generate.py:375-380 decode,
// src/flux/pipeline_tools.py:7-30 encode) for gfx950. The VAE sits either side of the denoise loop; its
// convolutions run as implicit GEMMs on the DiT's MFMA kernel (lx_gemm_bf16: im2col rows x [Cout, 9 Cin] weights, fused
// bias / fp32 residual epilogues), so what lives here is the HBM-bound glue in NHWC layout:
//   * GroupNorm(32 groups, eps, affine) [+ SiLU]: partial sums per (image, row chunk, group), then one normalising pass;
//   * im2col for 3x3 convolutions: pad 1 / stride 1, the encoder's stride-2 downsample with its (0,1,0,1) padding, and the
//     decoder's nearest-neighbour 2x upsample folded into the gather (the upsampled image is never materialised);
//   * row softmax (fp32 scores -> bf16 probabilities) for the single-head mid-block attention.
// All accesses are 8-16 B per lane along the channel axis.
#include "common.h"

namespace {

// ---- GroupNorm ---------------------------------------------------------------------------------------------------------------
// x: [B, P, C] (P = H*W pixels), fp32 or bf16. Pass 1: block (chunk, b) sums its rows per channel in registers, folds channels
// into groups through LDS, writes part[b][chunk][g] = (sum, sumsq). Pass 2 re-reduces the few partials per group on the fly.
template <typename T>
__device__ __forceinline__ f32x4 load4(const T* p);
template <>
__device__ __forceinline__ f32x4 load4<float>(const float* p) { return *(const f32x4*)p; }
template <>
__device__ __forceinline__ f32x4 load4<uint16_t>(const uint16_t* p) {
  const u32x2 r = *(const u32x2*)p;
  return f32x4{__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u), __uint_as_float(r[1] << 16), __uint_as_float(r[1] & 0xffff0000u)};
}

template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, int P, int C, int G, int rows_per_chunk, float* __restrict__ part) {
  __shared__ float red[2][256];                         // every thread's (sum, sumsq); folded per group in a FIXED order (no atomics)
  const int chunk = blockIdx.x, b = blockIdx.y, nchunk = gridDim.x;
  const int lanes_per_row = C / 4, rows_per_iter = 256 / lanes_per_row;
  const int tid = threadIdx.x, c4 = (tid % lanes_per_row) * 4, rsub = tid / lanes_per_row;
  const int r0 = chunk * rows_per_chunk, r1 = min(P, r0 + rows_per_chunk);
  float s = 0.f, q = 0.f;                               // the thread's 4 channels lie in ONE group (C/G >= 4, multiple of 4)
  const T* xb = x + (size_t)b * P * C + c4;
  for (int r = r0 + rsub; r < r1; r += rows_per_iter) {
    const f32x4 v = load4<T>(xb + (size_t)r * C);
    s += (v[0] + v[1]) + (v[2] + v[3]);
    q += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  }
  red[0][tid] = s;
  red[1][tid] = q;
  __syncthreads();
  if (tid < G) {
    const int lpg = (C / G) / 4;                        // lanes per group within a row
    float ss = 0.f, qq = 0.f;
    for (int rs = 0; rs < rows_per_iter; ++rs)
      for (int j = 0; j < lpg; ++j) {
        const int t = rs * lanes_per_row + tid * lpg + j;
        ss += red[0][t];
        qq += red[1][t];
      }
    float* o = part + (((size_t)b * nchunk + chunk) * G + tid) * 2;
    o[0] = ss;
    o[1] = qq;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, int P, int C, int G, const float* __restrict__ part, int nchunk,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int silu,
                                                       uint16_t* __restrict__ y) {
  const int b = blockIdx.y;
  const size_t n4 = (size_t)P * C / 4;
  const float inv_n = 1.0f / ((float)P * (float)(C / G));
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const int c4 = (int)((i * 4) % C);
    const int g = c4 / (C / G);
    float s = 0.f, q = 0.f;
    for (int k = 0; k < nchunk; ++k) {
      const float* p = part + (((size_t)b * nchunk + k) * G + g) * 2;
      s += p[0];
      q += p[1];
    }
    const float mean = s * inv_n;
    const float rstd = rsqrtf(fmaxf(q * inv_n - mean * mean, 0.f) + eps);
    const f32x4 v = load4<T>(x + (size_t)b * P * C + i * 4);
    const f32x4 ga = *(const f32x4*)(gamma + c4), be = *(const f32x4*)(beta + c4);
    float o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float t = (v[c] - mean) * rstd * ga[c] + be[c];
      if (silu) t = t / (1.0f + __expf(-t));
      o[c] = t;
    }
    *(u32x2*)(y + (size_t)b * P * C + i * 4) = u32x2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
  }
}

// ---- im2col for 3x3 convolutions, NHWC bf16 -> [B*Ho*Wo, Kpad] bf16, column = (dy*3 + dx)*C + c ------------------------------------
// mode 0: stride 1, pad 1 (Ho = H, Wo = W);  mode 1: stride 2, pad (0,1,0,1) (Ho = H/2, Wo = W/2: diffusers Downsample2D with padding=0);
// mode 2: nearest 2x upsample, then stride 1 pad 1 (Ho = 2H, Wo = 2W: Upsample2D + its conv). Columns >= 9*C are zero padding.
__global__ __launch_bounds__(256) void im2col3x3_kernel(const uint16_t* __restrict__ x, int B, int H, int W, int C, int mode, uint16_t* __restrict__ out,
                                                        int Kpad) {
  const int Ho = mode == 1 ? H / 2 : (mode == 2 ? 2 * H : H), Wo = mode == 1 ? W / 2 : (mode == 2 ? 2 * W : W);
  const int cv = C % 8 == 0 ? 8 : 1;                       // channels per work item
  const int per_tap = C / cv;
  const size_t items = (size_t)B * Ho * Wo * 9 * per_tap;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < items; i += (size_t)gridDim.x * 256) {
    const int cc = (int)(i % per_tap) * cv;
    size_t r = i / per_tap;
    const int tap = (int)(r % 9);
    r /= 9;                                                // output pixel index (b, yo, xo)
    const int xo = (int)(r % Wo), yo = (int)((r / Wo) % Ho), b = (int)(r / ((size_t)Wo * Ho));
    const int dy = tap / 3, dx = tap % 3;
    int yi, xi;
    bool ok;
    if (mode == 1) { yi = 2 * yo + dy; xi = 2 * xo + dx; ok = yi < H && xi < W; }
    else if (mode == 2) { const int yu = yo + dy - 1, xu = xo + dx - 1; ok = yu >= 0 && yu < Ho && xu >= 0 && xu < Wo; yi = yu >> 1; xi = xu >> 1; }
    else { yi = yo + dy - 1; xi = xo + dx - 1; ok = yi >= 0 && yi < H && xi >= 0 && xi < W; }
    uint16_t* o = out + r * Kpad + tap * C + cc;
    const uint16_t* s = x + (((size_t)b * H + (ok ? yi : 0)) * W + (ok ? xi : 0)) * C + cc;
    if (cv == 8) *(u32x4*)o = ok ? *(const u32x4*)s : u32x4{0u, 0u, 0u, 0u};
    else *o = ok ? *s : (uint16_t)0;
  }
  // zero the K padding (columns [9C, Kpad))
  const int padc = Kpad - 9 * C;
  if (padc > 0) {
    const size_t rows = (size_t)B * Ho * Wo, n = rows * padc;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[(i / padc) * Kpad + 9 * C + (i % padc)] = 0;
  }
}

// ---- row softmax: fp32 scores [M, N] (lds) * scale -> bf16 probabilities [M, N] (ldp). One wave per row. ----------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S, int lds, float scale, uint16_t* __restrict__ Pm, int ldp, int M, int N) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* s = S + (size_t)row * lds;
  float m = -INFINITY;
  for (int c = lane * 4; c < N; c += 256) {
    const f32x4 v = *(const f32x4*)(s + c);
    m = fmaxf(m, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
  }
  m = wave_max(m) * scale;
  float l = 0.f;
  for (int c = lane * 4; c < N; c += 256) {
    const f32x4 v = *(const f32x4*)(s + c);
#pragma unroll
    for (int k = 0; k < 4; ++k) l += __expf(v[k] * scale - m);
  }
  const float inv = 1.0f / wave_sum(l);
  uint16_t* p = Pm + (size_t)row * ldp;
  for (int c = lane * 4; c < N; c += 256) {
    const f32x4 v = *(const f32x4*)(s + c);
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = __expf(v[k] * scale - m) * inv;
    *(u32x2*)(p + c) = u32x2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
  }
}

}  // namespace

extern "C" size_t lx_groupnorm_workspace_bytes(int B, int P, int G) {
  const int nchunk = P >= 4096 ? 64 : (P >= 256 ? 16 : 1);
  return (size_t)B * nchunk * G * 2 * sizeof(float);
}

extern "C" int lx_groupnorm_silu(const void* x, int x_is_bf16, int B, int P, int C, int G, const float* gamma, const float* beta, float eps,
                                 int silu, void* y, void* ws, size_t ws_bytes, void* stream) {
  LX_CHECK_ARG(x && y && gamma && beta && ws, "lx_groupnorm_silu: NULL operand");
  LX_CHECK_ARG(B > 0 && P > 0 && C > 0 && G > 0 && G <= 64 && C % G == 0 && (C / G) % 4 == 0 && C <= 1024 && 256 % (C / 4) == 0, "lx_groupnorm_silu: need G <= 64, C %% G == 0, (C/G) %% 4 == 0, C/4 a divisor of 256 (C=%d G=%d)", C, G);
  LX_CHECK_ARG(ws_bytes >= lx_groupnorm_workspace_bytes(B, P, G), "lx_groupnorm_silu: workspace too small");
  const int nchunk = P >= 4096 ? 64 : (P >= 256 ? 16 : 1);
  const int rpc = (P + nchunk - 1) / nchunk;
  hipStream_t s = (hipStream_t)stream;
  float* part = (float*)ws;
  const size_t n4 = (size_t)P * C / 4;
  const int gx = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  if (x_is_bf16) {
    hipLaunchKernelGGL(gn_stats_kernel<uint16_t>, dim3(nchunk, B), dim3(256), 0, s, (const uint16_t*)x, P, C, G, rpc, part);
    hipLaunchKernelGGL(gn_apply_kernel<uint16_t>, dim3(gx, B), dim3(256), 0, s, (const uint16_t*)x, P, C, G, part, nchunk, gamma, beta, eps, silu, (uint16_t*)y);
  } else {
    hipLaunchKernelGGL(gn_stats_kernel<float>, dim3(nchunk, B), dim3(256), 0, s, (const float*)x, P, C, G, rpc, part);
    hipLaunchKernelGGL(gn_apply_kernel<float>, dim3(gx, B), dim3(256), 0, s, (const float*)x, P, C, G, part, nchunk, gamma, beta, eps, silu, (uint16_t*)y);
  }
  LX_LAUNCH_CHECK("lx_groupnorm_silu");
  return LX_OK;
}

extern "C" int lx_im2col3x3(const void* x, int B, int H, int W, int C, int mode, void* out, int Kpad, void* stream) {
  LX_CHECK_ARG(x && out && B > 0 && H > 0 && W > 0 && C > 0, "lx_im2col3x3: bad arguments");
  LX_CHECK_ARG(mode >= 0 && mode <= 2 && (mode != 1 || (H % 2 == 0 && W % 2 == 0)), "lx_im2col3x3: mode 0|1|2 (mode 1 needs even H, W)");
  LX_CHECK_ARG(Kpad >= 9 * C && Kpad % 8 == 0, "lx_im2col3x3: Kpad=%d must be >= 9*C and a multiple of 8", Kpad);
  const int Ho = mode == 1 ? H / 2 : (mode == 2 ? 2 * H : H), Wo = mode == 1 ? W / 2 : (mode == 2 ? 2 * W : W);
  const size_t items = (size_t)B * Ho * Wo * 9 * (C % 8 == 0 ? C / 8 : C);
  const int grid = (int)((items + 255) / 256 < 16384 ? (items + 255) / 256 : 16384);
  hipLaunchKernelGGL(im2col3x3_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, B, H, W, C, mode, (uint16_t*)out, Kpad);
  LX_LAUNCH_CHECK("lx_im2col3x3");
  return LX_OK;
}

extern "C" int lx_softmax_rows(const float* S, int lds, float scale, void* P, int ldp, int M, int N, void* stream) {
  LX_CHECK_ARG(S && P && M > 0 && N > 0 && N % 4 == 0 && lds % 4 == 0 && ldp % 4 == 0, "lx_softmax_rows: N, lds, ldp must be multiples of 4");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, S, lds, scale, (uint16_t*)P, ldp, M, N);
  LX_LAUNCH_CHECK("lx_softmax_rows");
  return LX_OK;
}
