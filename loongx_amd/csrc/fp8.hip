// fp8.hip -- producers of the e4m3 operand images of the fp8 GEMM path (gemm.hip, lx_gemm_fp8_kernel; BASELINE configs[4]):
//   * AdaLN LayerNorm + modulation writing BOTH the bf16 operand (LoRA down-projection, attention prep stay bf16) and its e4m3
//     image x * scale for the projections;
//   * bf16 / fp32 -> e4m3 row converter (the attention output before to_out / proj_out);
//   * LoRA down-projection reading an e4m3 activation image (the MLP hidden exists only as fp8 in this mode).
// OCP e4m3 (gfx950 v_cvt_pk_fp8_f32, saturating by an explicit clamp to +-448). HBM-bound row kernels, 8-16 B per lane.
#include "common.h"

namespace {

__device__ __forceinline__ float clamp8(float x) { return fminf(fmaxf(x, -448.f), 448.f); }
__device__ __forceinline__ uint32_t pk8(float a, float b, float c, float d) {
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(clamp8(a), clamp8(b), 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(clamp8(c), clamp8(d), w, true);
  return (uint32_t)w;
}

struct LnSegs8 {
  int n;
  int row0[3], n_rows[3], rows_per_batch[3];
  const float* shift[3];
  const float* scale[3];
};

// one wave per row, three passes over the (L2-resident) row; Y (bf16) may be NULL
__global__ __launch_bounds__(256) void ln_modulate_fp8_kernel(const float* __restrict__ X, int ldx, const LnSegs8 segs, int mod_ld, uint16_t* __restrict__ Y,
                                                              int ldy, uint8_t* __restrict__ Y8, int ldy8, float s8, int M, int D, float eps) {
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
  for (int c = lane * 4; c < D; c += 256) {
    const f32x4 v = *(const f32x4*)(xr + c);
    const f32x4 a = *(const f32x4*)(sc + c);
    const f32x4 bsh = *(const f32x4*)(sh + c);
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = (v[k] - mean) * rstd * (1.0f + a[k]) + bsh[k];
    if (Y) *(u32x2*)(Y + (size_t)row * ldy + c) = u32x2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
    *(uint32_t*)(Y8 + (size_t)row * ldy8 + c) = pk8(o[0] * s8, o[1] * s8, o[2] * s8, o[3] * s8);
  }
}

__global__ __launch_bounds__(256) void convert_fp8_kernel(const void* __restrict__ src, int src_bf16, int lds, uint8_t* __restrict__ dst, int ldd, float s8,
                                                          int M, int K) {
  const int kq = K >> 3;
  const size_t n = (size_t)M * kq, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int m = (int)(i / kq), k = (int)(i % kq) * 8;
    float x[8];
    if (src_bf16) {
      const u32x4 r = *(const u32x4*)((const uint16_t*)src + (size_t)m * lds + k);
#pragma unroll
      for (int j = 0; j < 4; ++j) { x[2 * j] = __uint_as_float(r[j] << 16); x[2 * j + 1] = __uint_as_float(r[j] & 0xffff0000u); }
    } else {
      const f32x4 a = *(const f32x4*)((const float*)src + (size_t)m * lds + k), b = *(const f32x4*)((const float*)src + (size_t)m * lds + k + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { x[j] = a[j]; x[4 + j] = b[j]; }
    }
    *(u32x2*)(dst + (size_t)m * ldd + k) = u32x2{pk8(x[0] * s8, x[1] * s8, x[2] * s8, x[3] * s8), pk8(x[4] * s8, x[5] * s8, x[6] * s8, x[7] * s8)};
  }
}

// LoRA down-projection with an e4m3 activation image: T[M, R<=16] = descale * X8[M, K] . A[R, K]^T. Same decomposition as
// lora_down_mfma_kernel (rowops.hip): 16 rows per workgroup, 8 waves split K, v_mfma_f32_16x16x32_bf16 after an in-register
// e4m3 -> bf16 conversion of the lane's 8 activations (exact: every e4m3 value is a bf16 value).
__global__ __launch_bounds__(512) void lora_down_fp8_kernel(const uint8_t* __restrict__ X8, int ldx, float descale, const uint16_t* __restrict__ A,
                                                            float* __restrict__ T, int ldt, int M, int K, int R, int Ks, int split_stride) {
  __shared__ f32x4 red[8][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m0 = blockIdx.x * 16;
  const int kbeg = blockIdx.y * Ks, kend = min(K, kbeg + Ks);
  T += (size_t)blockIdx.y * split_stride;
  const int l15 = lane & 15, kq = lane >> 4;
  const uint8_t* xp = X8 + (size_t)min(m0 + l15, M - 1) * ldx + kq * 8;
  const uint16_t* ap = A + (size_t)min(l15, R - 1) * K + kq * 8;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k = kbeg + wave * 32; k < kend; k += 8 * 32) {
    const bf16x8 af = *(const bf16x8*)(ap + k);
    const u32x2 xb = *(const u32x2*)(xp + k);
    u32x4 xw;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)xb[h], false), hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)xb[h], true);
      xw[2 * h] = pack_bf16x2(lo[0], lo[1]);
      xw[2 * h + 1] = pack_bf16x2(hi[0], hi[1]);
    }
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf16x8, xw), acc, 0, 0, 0);
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int w = 1; w < 8; ++w) {
      const f32x4 o = red[w][lane];
      acc[0] += o[0]; acc[1] += o[1]; acc[2] += o[2]; acc[3] += o[3];
    }
    const int m = m0 + l15, r0 = 4 * kq;
    if (m < M && r0 < R) {
      float* tp = T + (size_t)m * ldt + r0;
      for (int j = 0; j < 4 && r0 + j < R; ++j) tp[j] = acc[j] * descale;
    }
  }
}

}  // namespace

extern "C" int lx_ln_modulate_fp8_segs(const float* X, int ldx, const lx_ln_seg* seg, int n_seg, int mod_ld, void* Y, int ldy, void* Y8, int ldy8,
                                       float y8_scale, int D, float eps, void* stream) {
  LX_CHECK_ARG(seg && n_seg >= 1 && n_seg <= 3, "lx_ln_modulate_fp8_segs: 1..3 segments");
  LX_CHECK_ARG(X && Y8 && D > 0 && D % 4 == 0 && y8_scale > 0.f, "lx_ln_modulate_fp8_segs: bad arguments");
  LX_CHECK_ARG(ldx % 4 == 0 && ldy % 4 == 0 && ldy8 % 4 == 0 && mod_ld % 4 == 0 && ((uintptr_t)X & 15) == 0 && ((uintptr_t)Y & 7) == 0 && ((uintptr_t)Y8 & 3) == 0,
               "lx_ln_modulate_fp8_segs: leading dimensions must be multiples of 4, operands aligned");
  LnSegs8 segs;
  segs.n = n_seg;
  int M = 0;
  for (int i = 0; i < n_seg; ++i) {
    LX_CHECK_ARG(seg[i].shift && seg[i].scale && seg[i].n_rows > 0 && seg[i].rows_per_batch > 0, "lx_ln_modulate_fp8_segs: bad segment %d", i);
    LX_CHECK_ARG((((uintptr_t)seg[i].shift | (uintptr_t)seg[i].scale) & 15) == 0, "lx_ln_modulate_fp8_segs: misaligned modulation table");
    segs.row0[i] = seg[i].row0; segs.n_rows[i] = seg[i].n_rows; segs.rows_per_batch[i] = seg[i].rows_per_batch;
    segs.shift[i] = seg[i].shift; segs.scale[i] = seg[i].scale;
    M += seg[i].n_rows;
  }
  hipLaunchKernelGGL(ln_modulate_fp8_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, X, ldx, segs, mod_ld, (uint16_t*)Y, ldy, (uint8_t*)Y8, ldy8,
                     y8_scale, M, D, eps);
  LX_LAUNCH_CHECK("lx_ln_modulate_fp8_segs");
  return LX_OK;
}

extern "C" int lx_convert_fp8(const void* src, int src_is_bf16, int lds, void* dst, int ldd, float scale, int M, int K, void* stream) {
  LX_CHECK_ARG(src && dst && M > 0 && K > 0 && scale > 0.f, "lx_convert_fp8: bad arguments");
  LX_CHECK_ARG(K % 8 == 0 && lds % 8 == 0 && ldd % 8 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 7) == 0, "lx_convert_fp8: K, lds, ldd must be multiples of 8, operands aligned");
  const size_t n = (size_t)M * (K / 8);
  const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(convert_fp8_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, src_is_bf16, lds, (uint8_t*)dst, ldd, scale, M, K);
  LX_LAUNCH_CHECK("lx_convert_fp8");
  return LX_OK;
}

extern "C" int lx_lora_down_fp8(const void* X8, int ldx, float x_descale, const void* Adown, float* T, int ldt, int M, int K, int R, int n_split,
                                int split_stride, void* stream) {
  LX_CHECK_ARG(X8 && Adown && T && M > 0, "lx_lora_down_fp8: NULL operand");
  LX_CHECK_ARG(R >= 1 && R <= 16, "lx_lora_down_fp8: R=%d must be in [1,16]", R);
  LX_CHECK_ARG(K % 32 == 0 && ldx % 8 == 0 && ldt >= R, "lx_lora_down_fp8: K %% 32, ldx %% 8 and ldt >= R required (K=%d)", K);
  LX_CHECK_ARG(((uintptr_t)X8 & 7) == 0 && ((uintptr_t)Adown & 15) == 0, "lx_lora_down_fp8: operands must be aligned");
  LX_CHECK_ARG(n_split >= 1 && n_split <= 16 && (n_split == 1 || split_stride >= (M - 1) * ldt + R), "lx_lora_down_fp8: bad n_split=%d / split_stride=%d", n_split, split_stride);
  const int Ks = ((K / 32 + n_split - 1) / n_split) * 32;
  hipLaunchKernelGGL(lora_down_fp8_kernel, dim3((M + 15) / 16, n_split), dim3(512), 0, (hipStream_t)stream, (const uint8_t*)X8, ldx, x_descale,
                     (const uint16_t*)Adown, T, ldt, M, K, R, Ks, split_stride);
  LX_LAUNCH_CHECK("lx_lora_down_fp8");
  return LX_OK;
}
