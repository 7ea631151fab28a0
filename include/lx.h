/* lx.h -- C ABI of the MI355X-native LoongX denoise hot path (liblx_amd.so).
 *
 * Drop-in boundary.  The reference (LanceZPF/loongx) is pure Python and has no FFI of its own:
 * its hot path calls torch / diffusers / s4torch operators from
 *   src/flux/block.py        (attn_forward :7-176, block_forward :179-278, single_block_forward :281-339)
 *   src/flux/transformer.py  (tranformer_forward :47-252)
 *   src/flux/generate.py     (denoise loop :313-369, scheduler.step :349)
 *   src/train/model.py       (CS3 encoders :16-373, DUAN/DGF :947-1035, fuse_* :731-779)
 * Each entry point below replaces the operator(s) named in its comment.  The Python binding a maintainer
 * would add is a ctypes stub; loongx_amd/_lib.py is that binding.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller; kernels never allocate, never synchronise,
 *    never retain a pointer after the call returns (model handles excepted, see lx_dit_*);
 *  - every call enqueues on `stream` (a hipStream_t passed as void*) and returns LX_OK or a negative
 *    lx_status; lx_last_error() returns the message of the calling thread's last failure;
 *  - bf16 = bfloat16 bits (uint16), row-major, leading dimensions in ELEMENTS.
 */
#ifndef LX_H_
#define LX_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ABI history:
 *  0.4.3  + LX_ATTN_P_EXP2; lx_attn_fwd_fp8's default probability bytes are the log-linear code of the score (POW2 scales)
 *  0.4.2  + lx_qkv_prep_f16in_segs, lx_qkv_prep_fp8_f16in_segs (the separate RMSNorm + RoPE + V^T pass on a projection an LX_OPERANDS_F16
 *         launch stored as fp16: stream lengths LX_EPI_QKV does not take)
 *  0.4.1  + lx_ln_modulate_lora_f16_segs; lx_ln_modulate_lora_segs computes its down-projection on the matrix pipe (Adown 16-byte aligned);
 *         + LX_ATTN_PREFER_4WAVE
 *  0.4.0  fp16 operand format: LX_OPERANDS_F16 (lx_gemm_desc.f16_ovf in col_scale's slot), LX_ATTN_O_F16, lx_attn_desc.qseg_mask (in the
 *         padding behind `flags`) + f16_ovf (appended), + lx_ln_modulate_f16_segs, lx_lora_down_f16, lx_convert dst 2 = fp16; the split-K pair
 *         kernel and its area of the workspace are gone (lx_gemm_workspace_bytes() shrank; the error word is still the int 64 ints before
 *         the end)
 *  0.3.3  + lx_attn_last_kernel (which attention kernel lx_attn_fwd launched)
 *  0.3.2  lx_gemm_workspace_bytes() grew (split-tile slots of lx_gemm4_kernel); no layout change
 *  0.3.1  lx_attn_desc grew (flags, appended: LX_ATTN_Q_LOG2 / LX_ATTN_BOUNDED)
 *  0.3.0  lx_gemm_desc grew (LX_EPI_QKV e4m3 outputs: qkv_q8 ... qkv_v_scale, appended); + lx_qkv_prep_split_segs, lx_attn_fwd_split,
 *         lx_lora_down_terms
 *  0.2.0  caller-owned GEMM workspace, precise mode, VAE row kernels, channel-major fp32 GEMM */
#define LX_VERSION 403

typedef enum lx_status {
  LX_OK = 0,
  LX_ERR_INVALID = -1,   /* bad argument (shape / alignment / NULL) */
  LX_ERR_UNSUPPORTED = -2,
  LX_ERR_LAUNCH = -3,    /* HIP launch failure */
  LX_ERR_NO_DEVICE = -4
} lx_status;

int lx_version(void);
const char* lx_last_error(void);
/* fills name[0..n) with the gcnArchName of the current device; LX_ERR_NO_DEVICE when there is none */
int lx_device_arch(char* name, size_t n);

/* ------------------------------------------------------------------------------------------------
 * GEMM with fused epilogues  --  replaces every nn.Linear on the DiT path:
 *   to_q/k/v, add_{q,k,v}_proj (block.py:27-29,46-48,81-83), to_out[0]/to_add_out (:154-160),
 *   ff / ff_context (:258-265), proj_mlp / proj_out of the single block (:302-333),
 *   x_embedder / context_embedder / proj_out (transformer.py:92-93,115,244).
 * C[m,n] = sum_k A[m,k] * W[n,k]  (bf16 x bf16 -> fp32 MFMA accumulate), then per `epilogue`.
 * LoRA (peft, lora_controller.py:5-42): rows of a problem with lora_t != NULL get
 *   acc += sum_r lora_t[m, toff + r] * lora_up[n, r],  toff = lora_r * min(n / lora_mod_cols, lora_toff_max)
 * i.e. y = x W^T + s * (x A^T) B^T evaluated in fp32, with lora_t = x A^T from lx_lora_down and
 * lora_up = s*B.  The image stream passes lora_t = NULL (enable_lora scales the adapter to 0).
 * ------------------------------------------------------------------------------------------------ */
enum {
  LX_EPI_STORE_BF16 = 0, /* C(bf16)  = act(acc + bias)                      */
  LX_EPI_STORE_F32 = 1,  /* C(fp32)  = acc + bias                           */
  LX_EPI_RESID_F32 = 2,  /* C(fp32) += gate[m / rows_per_batch, n] * (acc + bias)   (gate NULL => 1) */
  LX_EPI_GELU = 0x100,   /* OR-able flag: GELU(tanh) on columns n >= gelu_col_start */
  LX_W_TILED = 0x200,    /* OR-able flag: W is pre-tiled (see lx_tile_weight_layout): [N/256][K/64] blocks of 256x64,
                            each stored as the swizzled LDS image the kernel consumes; needs N % 256 == 0, ldw == K */
  LX_EPI_SPLIT_BF16 = 0x400, /* OR-able flag (with LX_EPI_STORE_BF16): also store bf16(x - bf16(x)) at column n + c_lo_off, so
                            that a consumer GEMM with k_segs >= 2 sees x to 16 mantissa bits (precise mode) */
  LX_EPI_STORE_FP8 = 3,  /* (fp8 GEMMs only) C(e4m3 bytes) = act(acc * col_scale + bias) * out_scale, saturated to +-448 */
  LX_EPI_QKV = 0x1000,   /* OR-able flag (with LX_EPI_STORE_BF16, bf16 operands): the first 3*qkv_d output columns are the attention
                          * projections [k | v | q] of heads of 128 (block.py:43-99): the epilogue applies RMSNorm(128, norm_k /
                          * norm_q) + RoPE to the k and q columns in fp32 BEFORE the single bf16 rounding and stores them in place,
                          * and writes the v columns straight into the V^T image the attention kernel reads (and nowhere else):
                          * lx_qkv_prep of the same buffer is then not needed. See the qkv_* fields. */
  LX_OPERANDS_FP8 = 0x800, /* OR-able flag: A and W are OCP e4m3 bytes (lda / ldw in bytes, K % 128 == 0), products on the
                            64-deep f8f6f4 MFMA at twice the bf16 rate; acc[m,n] *= col_scale[n] before everything else.
                            BASELINE configs[4]; opt-in (model_config["gemm_fp8"]). Every problem of a launch must carry it. */
  LX_OPERANDS_F16 = 0x2000 /* OR-able flag: A and W are IEEE fp16 (same 16-bit layouts, tiling and leading dimensions as bf16), products on
                            v_mfma_f32_*_f16 at the bf16 rate, fp32 accumulate: 11 significand bits on the operands instead of 8 -- the
                            reference computes every Linear in fp32 (train/config/seed_512.yaml:2) and the bf16 mode's whole per-forward
                            error is the 8-bit rounding of the A operands (tools/bf16_ablation.py). A 16-bit store (LX_EPI_STORE_BF16 kind)
                            then writes fp16 as well -- the next GEMM's operand -- rounded to nearest even and SATURATED to +-65504; the
                            launch adds the number of producer waves that clipped a value to *f16_ovf (when non-NULL). LX_EPI_QKV outputs
                            (k, q, V^T: the attention kernel's operands) stay bf16. Every problem of a launch must carry it; not with
                            LX_OPERANDS_FP8 / k_segs >= 2 / LX_EPI_SPLIT_BF16. model_config["operands"] = "fp16" / dtype=torch.float16. */
};

typedef struct lx_gemm_desc {
  const void* A;        /* [M,K] bf16, lda */
  const void* W;        /* [N,K] bf16, ldw (nn.Linear.weight layout) */
  const float* bias;    /* [N] or NULL */
  void* C;              /* [M,N], ldc; bf16 or fp32 per epilogue */
  const float* gate;    /* [ceil(M/rows_per_batch), gate_ld] or NULL */
  const float* lora_t;  /* [M, lora_ldt] or NULL */
  const float* lora_up; /* [N, lora_r] */
  int32_t M, N, K;
  int32_t lda, ldw, ldc;
  int32_t rows_per_batch, gate_ld;
  int32_t lora_r, lora_ldt, lora_mod_cols, lora_toff_max;
  int32_t epilogue;
  int32_t gelu_col_start;
  int32_t lora_nsplit;       /* lora_t is the sum of this many K-split partial slabs (0/1 = a single slab) ... */
  int32_t lora_split_stride; /* ... spaced this many floats apart (as written by lx_lora_down) */
  /* Precise mode ("split bf16": fp32-class products on the bf16 MFMA, for the reference's fp32 configuration,
   * train/config/seed_512.yaml:2). k_segs = 0/1: plain. k_segs = 2: A = A_hi + A_lo with A_hi = bf16(a), A_lo = bf16(a - A_hi)
   * stored a_lo_off columns after A_hi in the same rows; C accumulates A_hi W^T + A_lo W^T (exact for bf16-representable
   * weights, which is what FLUX.1 checkpoints hold). k_segs = 3: W is [N, 2K] = [W_hi | W_lo] as well (ldw >= 2K) and
   * A_hi W_lo^T is added: 3 of the 4 cross terms, relative error ~2^-16. One accumulation, one epilogue. */
  int32_t k_segs, a_lo_off;
  int32_t c_lo_off;          /* LX_EPI_SPLIT_BF16: column distance of the lo image of the output */
  float out_scale;           /* LX_EPI_STORE_FP8: multiplier applied before the e4m3 rounding */
  union {                    /* one pointer slot for the two operand modes that exclude each other (the descriptor travels by value in the
                              * 4 KiB kernarg segment, eight at a time, twice for the mixed-height plan: it cannot grow) */
    const float* col_scale;  /* LX_OPERANDS_FP8: [N] fp32, 1 / (activation scale * weight scale of row n); NULL => 1 */
    int32_t* f16_ovf;        /* LX_OPERANDS_F16 with a 16-bit store: device counter, += the number of producer waves of this launch that
                              * saturated a value to +-65504 (NULL: not reported). Zeroed and read by the caller. */
  };
  /* LX_EPI_QKV (replaces lx_qkv_prep for this problem's rows; block.py:60-99 attn.norm_q / norm_k + apply_rotary_emb):
   * columns [0, qkv_d) = k, [qkv_d, 2 qkv_d) = v, [2 qkv_d, 3 qkv_d) = q; qkv_d % 256 == 0; rows_per_batch % 32 == 0 (one
   * token stream: row m is position (m % rows_per_batch) of batch m / rows_per_batch); N may stop short of 3 qkv_d at a
   * multiple of 256 (a block whose q nobody reads). */
  const float* qkv_norm_q;   /* [128] fp32 RMSNorm weights (eps 1e-6) */
  const float* qkv_norm_k;   /* [128] */
  const float* qkv_rope;     /* [rows_per_batch, 128] fp32: (cos, sin) of rotary pair i at [2i, 2i+1] (identity: 1, 0) */
  void* qkv_vt;              /* bf16 [B * qkv_d/128, 128, qkv_vt_ld]: V^T image, key p of the stream at slot qkv_vt_pos0 + il(p), il =
                              * the 16-key interleave of lx_qkv_prep */
  void* qkv_k;               /* NULL: the k columns are stored in place in C. Otherwise bf16 [M, qkv_k_ld]: row m's k head h at column h*128 of
                              * a SEPARATE image (one per layer: a token stream whose queries see only its own keys -- model_config
                              * independent_condition / union_cond_attn = False -- then keeps its keys and values across denoise steps) */
  int32_t qkv_d, qkv_vt_ld, qkv_vt_pos0;   /* qkv_vt_ld, qkv_vt_pos0 % 64 == 0 */
  int32_t qkv_k_ld;          /* % 8 == 0, >= qkv_d */
  /* LX_EPI_QKV with e4m3 outputs, for the fp8 attention path (lx_attn_fwd_fp8; BASELINE configs[4]): when qkv_q8 is non-NULL the
   * epilogue writes what lx_qkv_prep_fp8_segs would have made of this projection -- q and k (after RMSNorm + RoPE, in fp32) as
   * bytes e4m3(x * qkv_q_scale) / e4m3(x * qkv_k_scale) into qkv_q8 / qkv_k8 [M, qkv_ld8] (head h at column h*128), v as
   * e4m3(v * qkv_v_scale) into the byte V^T image qkv_vt8 [B * qkv_d/128, 128, qkv_vt_ld] whose 64-key tiles are in the f8f6f4
   * MFMA's operand order (byte j = g*32 + p of a tile row: key (p>>4)*32 + 8*((p&15)>>2) + 4*g + (p&3)), at the same slot
   * qkv_vt_pos0 -- and NOTHING in bf16: C's k / v / q columns, qkv_vt and qkv_k are then not written (qkv_vt may be NULL). */
  void* qkv_q8;
  void* qkv_k8;
  void* qkv_vt8;
  int32_t qkv_ld8;           /* % 16 == 0, >= qkv_d (the byte V^T image has qkv_vt_ld bytes per row) */
  float qkv_q_scale, qkv_k_scale, qkv_v_scale;   /* > 0 */
} lx_gemm_desc;

#define LX_GEMM_MAX_GROUP 4
/* One launch over `n` independent problems (the three token streams of a block share one launch so
 * that small-M streams still fill the chip).  K % 64 == 0, N % 8 == 0, lda/ldw/ldc % 8 == 0.
 * The launch plan is chosen per call (256-row tiles / 128-row tiles / full rounds of 256 + a 128-row tail); results are
 * deterministic for a given plan. Environment, read when the library is loaded (lx_gemm_reload_env() re-reads it):
 * LX_GEMM_BM=256|128 forces a tile height. */
int lx_gemm_bf16(const lx_gemm_desc* problems, int n, void* stream);
void lx_gemm_reload_env(void);

/* The same with a caller-owned workspace, which unlocks launch plans that exchange partial accumulators between workgroups:
 * the split-K PAIR plan (two workgroups per 256-row tile, each half of K, accumulator blocks swapped through L2) for long-K
 * launches with <= 128 tiles of 256x256 that would otherwise leave half the chip idle (the N = 3072 projections of the DiT
 * at batch 1). Plans that split K differ from the others by one fp32 rounding per element.
 *   workspace: lx_gemm_workspace_bytes() bytes of device memory, 256-byte aligned, ZERO-FILLED ONCE by the caller before its
 *   first use and never touched by the caller afterwards; ONE PER STREAM that may have an lx_gemm_bf16_ws call in flight
 *   (the library keeps no scratch of its own, so launches on different streams never share slots or flags).
 *   NULL / 0 => exactly lx_gemm_bf16.
 * LX_GEMM_PAIR = 0 never | 1 (default) where it measures faster (K >= 6144) | 2 whenever possible (tests).
 * Passing a workspace also selects the plans that depend on the launch's tile count: lx_gemm4_kernel (the 256x256x64 tile with one
 * wave per SIMD, AGPR accumulators) for launches whose tiles fill whole rounds (LX_GEMM4 = 0 never | 1 default | 2 whenever its
 * epilogues allow; LX_GEMM4_SK = 1 default | 0 | 2 two-way only: its split form -- two workgroups per tile, half of K each, each parking
 * the half of the tile's rows the other one finishes; three, a third of K each, where 3 x the split tiles fit one round and K is long
 * (>= 96 K tiles) or the tail at most a sixth of a round -- for a partial last round (up to a third of a round; up to half of one for
 * launches of >= 96 K tiles without LX_EPI_QKV; launches of at most 16 rounds) and for the pair plan's shapes -- which exchanges through the
 * workspace under the same bounded-wait / error-word contract; split-bf16 problems (k_segs >= 2) take the same plans by cost).
 * Without one the plans do not depend
 * on the batch size: a data-parallel shard reproduces the single-GPU batch bit for bit.
 * Layout (for callers that poll asynchronously): the error word is the int 64 ints before the end of the workspace.
 * The pair plan needs both workgroups of a tile resident at once (<= 256 workgroups on a 256-CU device, checked); if other work
 * holds CUs for longer than the bounded wait (~1 s), the waiting workgroup raises the workspace's error word and finishes with
 * an invalid tile instead of hanging or trapping: lx_gemm_workspace_status() (synchronises `stream`) then returns
 * LX_ERR_LAUNCH once, resets the workspace, and the caller should recompute without the workspace. */
size_t lx_gemm_workspace_bytes(void);
int lx_gemm_bf16_ws(const lx_gemm_desc* problems, int n, void* workspace, size_t ws_bytes, void* stream);
int lx_gemm_workspace_status(void* workspace, void* stream);

/* LoRA down-projection (peft lora_A): T_s[M, R] (fp32, ldt) = X[M, K_s] (bf16, ldx) . Adown[R, K_s]^T (bf16), R <= 16,
 * for n_split contiguous K slices s (n_split = 1: the whole K); slab s is written at T + s*split_stride. Splitting K
 * spreads a tall-skinny product over the whole chip without atomics; the consumer (lx_gemm_bf16) adds the slabs. */
int lx_lora_down(const void* X, int ldx, const void* Adown, float* T, int ldt, int M, int K, int R, int n_split,
                 int split_stride, void* stream);
/* The same on fp16 images of X and Adown (the operands of an LX_OPERANDS_F16 launch; v_mfma_f32_16x16x32_f16) */
int lx_lora_down_f16(const void* X, int ldx, const void* Adown, float* T, int ldt, int M, int K, int R, int n_split,
                     int split_stride, void* stream);
/* The multi-term form (precise mode: x = x_hi + x_lo, A = A_hi [+ A_lo]): slab s of T (slab_stride floats apart, same row stride) =
 * X[s][M, K] . Adown[s][R, K]^T over the whole K, for n_terms <= 4 (X, Adown) pairs in ONE launch; the consumer GEMM adds the slabs
 * (lora_nsplit = n_terms). Same arithmetic per slab as lx_lora_down(X[s], ldx[s], Adown[s], T + s * slab_stride, ..., n_split = 1). */
int lx_lora_down_terms(const void* const* X, const int* ldx, const void* const* Adown, int n_terms, float* T, int ldt, int M, int K, int R,
                       int slab_stride, void* stream);

/* Skinny linear for a few rows (AdaLayerNorm modulation linears, time/text embedders:
 * block.py:192-207,301,305; transformer.py:102-114,243).  Weight-streaming, HBM bound; rows are processed four at a
 * time against the same weight rows, independently (M rows give bit-identical results to M one-row calls). Meant for
 * M <= 16: with more rows (e.g. the modulations of a whole sigma schedule, M = steps x batch) every wave re-loads all
 * x rows and lx_gemm_bf16 on bf16 hi/lo halves of x is the better tool (DiTEngine.prepare_schedule).
 * Y[M,N] fp32 (=|+=) act_out(act_in(X[M,K] fp32) . W[N,K]^T (bf16) + bias).  act: 0 none, 1 SiLU. */
int lx_linear_skinny(const float* X, int ldx, const void* W, int ldw, const float* bias, float* Y, int ldy,
                     int M, int N, int K, int act_in, int act_out, int accumulate, void* stream);

/* Sinusoidal timestep projection (diffusers get_timestep_embedding, flip_sin_to_cos=True, shift 0):
 * out[b, 0:half] = cos(t[b]*f_i), out[b, half:2*half] = sin(t[b]*f_i), f_i = exp(-ln(1e4) i/half). */
int lx_timestep_embed(const float* t, float* out, int B, int dim, void* stream);

/* RoPE tables (diffusers FluxPosEmbed, transformer.py:130-134): ids fp32 [L,3]; cos/sin fp32 [L, a0+a1+a2],
 * frequencies 1/theta^(2j/d_axis) and angles in fp64, each value repeated for the pair (2j, 2j+1). */
int lx_rope_table(const float* ids, int L, int a0, int a1, int a2, double theta, float* cos_t, float* sin_t, void* stream);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm (no affine, eps) + AdaLN modulation: Y(bf16)[m,:] = LN(X(fp32)[m,:]) * (1 + scale[b,:]) + shift[b,:]
 * b = m / rows_per_batch.  Replaces norm1/norm1_context/norm/norm2(+mod)/norm_out (block.py:192-207,238-253,301,305).
 * shift/scale: fp32 [n_batches, mod_ld].  D % 8 == 0, D <= 8192.
 * ------------------------------------------------------------------------------------------------ */
int lx_ln_modulate(const float* X, int ldx, const float* shift, const float* scale, int mod_ld, void* Y, int ldy,
                   int M, int D, int rows_per_batch, float eps, void* stream);
/* Same, over up to 3 row segments (the text / image / condition token streams) in ONE launch: segment i covers rows
 * [row0, row0+n_rows) of X and Y with its own shift/scale tables; batch = (row - row0) / rows_per_batch. */
typedef struct lx_ln_seg { int32_t row0, n_rows, rows_per_batch, _pad; const float* shift; const float* scale; } lx_ln_seg;
int lx_ln_modulate_segs(const float* X, int ldx, const lx_ln_seg* seg, int n_seg, int mod_ld, void* Y, int ldy, int D,
                        float eps, void* stream);
/* The same with Y as IEEE fp16 -- the A operand of an LX_OPERANDS_F16 GEMM: rounded to nearest even, saturated to +-65504;
 * *f16_ovf (device, may be NULL) += the number of waves (= rows) that clipped a value. */
int lx_ln_modulate_f16_segs(const float* X, int ldx, const lx_ln_seg* seg, int n_seg, int mod_ld, void* Y, int ldy, int D,
                            float eps, int32_t* f16_ovf, void* stream);
/* The same, and for the rows [lora_row0, lora_row0 + lora_rows) of Y (the token streams that run with their adapters on,
 * lora_controller.py:5-42) also the LoRA down-projection of the row just written: T[row - lora_row0, 0..R) (fp32, ldt) =
 * Y_row(bf16) . Adown[R, D]^T (bf16), R <= 16 -- what lx_lora_down(Y rows, Adown, T, n_split = 1) computes, without the
 * second pass over the rows (block.py:24, 299: the q/k/v(/proj_mlp) adapters read the AdaLN-normalised stream). D = 3072 | 256. */
int lx_ln_modulate_lora_segs(const float* X, int ldx, const lx_ln_seg* seg, int n_seg, int mod_ld, void* Y, int ldy, int D,
                             float eps, const void* Adown, int R, float* T, int ldt, int lora_row0, int lora_rows, void* stream);
/* lx_ln_modulate_lora_segs with fp16 operand images: Y as in lx_ln_modulate_f16_segs, Adown the fp16 image lx_lora_down_f16 takes */
int lx_ln_modulate_lora_f16_segs(const float* X, int ldx, const lx_ln_seg* seg, int n_seg, int mod_ld, void* Y, int ldy, int D,
                                 float eps, const void* Adown, int R, float* T, int ldt, int lora_row0, int lora_rows,
                                 int32_t* f16_ovf, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Per-head RMSNorm (weight, eps) + interleaved-pair RoPE on Q and K, in place, and V transposed into the
 * attention kernel's [B*H, dh, Spad] key-major image.  Replaces attn.norm_q/k/added_q/added_k +
 * apply_rotary_emb (block.py:38-41,60-67,74-78,92-99).
 *   QKV: bf16 [M, ld] with K at column k_col, V at v_col, Q at q_col (each H*128 wide, head-major).
 *   rows [row0, row0+n_rows): batch b = r / rows_per_batch, position p = r % rows_per_batch;
 *   cos/sin fp32 [rows_per_batch, 128] (NULL => no RoPE); wq/wk fp32 [128] (NULL => no RMSNorm);
 *   VT: bf16 [B, H, 128, vt_ld]; the row's key slot is vt_pos0 + p, stored with the 16-key interleave
 *   the attention kernel expects (see loongx_amd/csrc/attn.hip).  VT NULL => V untouched.
 * ------------------------------------------------------------------------------------------------ */
int lx_qkv_prep(void* QKV, int ld, int q_col, int k_col, int v_col, int row0, int n_rows, int rows_per_batch,
                int H, const float* wq, const float* wk, float eps, const float* cos_tab, const float* sin_tab,
                void* VT, int vt_ld, int vt_pos0, void* stream);

/* Same, over up to 3 token segments in ONE launch (each with its own norm weights, RoPE table and V^T slot). */
typedef struct lx_qkv_seg {
  int32_t row0, rows_per_batch, vt_pos0, _pad;
  const float* wq; const float* wk; const float* cos_tab; const float* sin_tab;
} lx_qkv_seg;
int lx_qkv_prep_segs(void* QKV, int ld, int q_col, int k_col, int v_col, const lx_qkv_seg* seg, int n_seg, int n_batches,
                     int H, float eps, void* VT, int vt_ld, void* stream);
/* The same on a QKV buffer whose q / k / v columns hold IEEE fp16 (the 16-bit store of an LX_OPERANDS_F16 projection launch WITHOUT LX_EPI_QKV):
 * q and k are read as fp16 and written back in place as bf16 (what lx_attn_fwd reads), V^T gets bf16(v). block.py:38-41,60-67,74-78,92-99. */
int lx_qkv_prep_f16in_segs(void* QKV, int ld, int q_col, int k_col, int v_col, const lx_qkv_seg* seg, int n_seg, int n_batches,
                           int H, float eps, void* VT, int vt_ld, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Joint attention over up to 3 token segments [text | image | condition] -- replaces
 * F.scaled_dot_product_attention + the mask/bias construction of block.py:106-135.
 *   Q,K: bf16 rows of `ld` elements; head h at column q_col/k_col + h*128; O is written over Q's slot
 *   layout with row stride ldo at column o_col (may alias Q: each Q tile is read before its O is written).
 *   Segment s of batch b occupies rows seg_row0[s] + b*seg_len[s] ... (+seg_len[s]); keys of segment s live
 *   at VT positions seg_vt0[s].. (64-aligned, zero padded).
 *   bias[3][3]: additive score bias (natural-log units) for (query segment, key segment); -INFINITY masks
 *   the pair entirely; NULL => all zero.  scale = 1/sqrt(dh).
 * ------------------------------------------------------------------------------------------------ */
typedef struct lx_attn_desc {
  const void* Q; const void* K; const void* VT; void* O;
  int32_t ldq, ldk, ldo, vt_ld;
  int32_t q_col, k_col, o_col;
  int32_t B, H;
  int32_t n_seg;
  int32_t seg_row0[3], seg_len[3], seg_vt0[3];
  float bias[3][3];
  float scale;
  int32_t n_qseg;        /* 0 / n_seg: every segment has queries. k < n_seg: only segments 0..k-1 do (keys and values of all n_seg segments
                          * are still attended to): the rows of the other segments of O are not written */
  int32_t flags;         /* LX_ATTN_* below (0 = the plain contract above) */
  int32_t qseg_mask;     /* 0: the query segments are the first n_qseg (above). Otherwise bit s set = segment s has queries, any subset (n_qseg is then
                          * ignored): the last single block of a forward needs the image rows' output only (transformer.py:243-252) -- its text and
                          * condition rows serve keys and values, and their q columns are not even computed */
  int32_t* f16_ovf;      /* LX_ATTN_O_F16: device counter, += the number of waves that saturated an output value (NULL: not reported) */
} lx_attn_desc;
/* LX_ATTN_Q_LOG2: q already carries scale * log2(e) (e.g. folded into the norm_q weight handed to LX_EPI_QKV / lx_qkv_prep): the kernel
 * takes q.k as the exp2 argument as it is (`scale` is ignored).
 * LX_ATTN_BOUNDED (needs LX_ATTN_Q_LOG2): the CALLER guarantees |q.k (log2 units) + bias * log2(e)| <= 100 for every (query, key) --
 * true whenever q and k come out of the per-head RMSNorm of block.py:38-41,60-67 with 16.4 * max|norm_q| * max|norm_k| + |bias| * 1.45 <=
 * 100: a normalised head vector has length <= sqrt(128) * max|w|, RoPE is a rotation. Softmax is shift-invariant and exp2 of such an
 * argument neither overflows nor leaves fp32's normal range over 2^16 keys, so the kernel keeps NO running maximum: no row max, no
 * rescale, no per-score multiply-add -- p = exp2(q.k [+ bias]). Results differ from the max-tracking form by rounding only. */
/* LX_ATTN_INVARIANT: the choice of kernel must not depend on the batch size of the launch (the planner otherwise picks between two kernels
 * whose row sums are accumulated in different orders -- equal to rounding, not bit for bit): what a data-parallel shard needs to reproduce
 * the single-GPU batch exactly (the engine sets it together with its batch-size-invariant GEMM plans, LX_PAIR_PLAN=0). */
/* LX_ATTN_O_F16: O is written as IEEE fp16 (nearest even, saturated to +-65504) instead of bf16: the A operand of an LX_OPERANDS_F16
 * output projection (to_out / proj_out). Q, K and V^T stay bf16 (4e-4 of the per-forward budget, tools/bf16_ablation.py). Also accepted
 * by lx_attn_fwd_fp8 (the only flag it takes). */
/* LX_ATTN_PREFER_4WAVE: take lx_attn4_kernel whenever the launch fits it (LX_ATTN_BOUNDED, addresses within its 32-bit offsets), whatever
 * the planner's shape rule says -- for benchmarks and tests of that kernel. LX_ATTN_INVARIANT is the opposite pin (always the 8-wave
 * kernels); the two together are rejected. */
/* LX_ATTN_P_EXP2 (lx_attn_fwd_fp8 only): probabilities by v_exp_f32 + nearest-even e4m3 rounding. Without it (the default where the scales
 * fold to a power of two, see lx_attn_fwd_fp8) the kernel takes each probability byte as the log-linear code of its score: one conversion
 * instead of an exponential -- the chord of 2^f over a mantissa step instead of its rounding; results differ within the e4m3 step. */
enum { LX_ATTN_Q_LOG2 = 1, LX_ATTN_BOUNDED = 2, LX_ATTN_INVARIANT = 4, LX_ATTN_O_F16 = 8, LX_ATTN_PREFER_4WAVE = 16, LX_ATTN_P_EXP2 = 32 };
int lx_attn_fwd(const lx_attn_desc* d, void* stream);
/* Which kernel the calling thread's last successful lx_attn_fwd launched (a planner decision, exposed for benchmarks and tests):
 * LX_ATTN_KERNEL_8WAVE: the 8-wave kernels of attn.hip (two waves per SIMD, 32 query rows per wave);
 * LX_ATTN_KERNEL_4WAVE: lx_attn4_kernel (attn4.hip: one wave per SIMD, 64 query rows per wave, persistent over query tiles) -- bounded-score
 *   launches of at least two rounds of workgroups with at most 64 key tiles per query tile (LX_ATTN_PREFER_4WAVE / LX_ATTN_INVARIANT pin the choice per launch). */
enum { LX_ATTN_KERNEL_NONE = 0, LX_ATTN_KERNEL_8WAVE = 1, LX_ATTN_KERNEL_4WAVE = 2 };
int lx_attn_last_kernel(void);

/* ------------------------------------------------------------------------------------------------
 * fp8 (OCP e4m3) attention path -- BASELINE configs[4] ("fp8 MFMA attention path"); opt-in, the bf16 path above is the
 * default. lx_qkv_prep_fp8_segs does the arithmetic of lx_qkv_prep_segs (RMSNorm + RoPE in fp32, block.py:75-99) but
 * leaves the bf16 buffer untouched and writes byte images instead:
 *   Q8, K8 : [rows, ld8] e4m3, head h at byte column h*128, values multiplied by q_scale / k_scale;
 *   VT8    : [B, H, 128, vt8_ld] e4m3, V * v_scale transposed per head, the 64 keys of every tile stored in the order
 *            the 32x32x64 f8f6f4 MFMA's B operand wants (attn.hip, lx_attn_fp8_kernel).
 * lx_attn_fwd_fp8 takes the same descriptor as lx_attn_fwd with Q / K / VT pointing at those images (ldq, ldk, q_col, k_col,
 * vt_ld in bytes; O is bf16 as before) plus qk_descale = 1 / (q_scale * k_scale) and v_descale = 1 / v_scale.
 * When scale * qk_descale * log2(e) is an exact power of two 2^-k (q_scale = 2^k * scale * log2(e) / k_scale: 16.32 for k = 11,
 * k_scale = 16, the default scale 1/sqrt(128)) the kernel applies the whole factor as the MX block scale of its score MFMAs and
 * takes exp2 of the scores as they come out of the matrix pipe; any other combination runs the generic path (one fma per score).
 * Softmax statistics and the output accumulators are fp32; P is rounded to e4m3.
 * ------------------------------------------------------------------------------------------------ */
int lx_qkv_prep_fp8_segs(const void* QKV, int ld, int q_col, int k_col, int v_col, const lx_qkv_seg* seg, int n_seg,
                         int n_batches, int H, float eps, void* Q8, void* K8, int ld8, void* VT8, int vt8_ld,
                         float q_scale, float k_scale, float v_scale, void* stream);
/* lx_qkv_prep_fp8_segs reading fp16 q / k / v columns (see lx_qkv_prep_f16in_segs); the QKV buffer is not modified */
int lx_qkv_prep_fp8_f16in_segs(const void* QKV, int ld, int q_col, int k_col, int v_col, const lx_qkv_seg* seg, int n_seg,
                               int n_batches, int H, float eps, void* Q8, void* K8, int ld8, void* VT8, int vt8_ld,
                               float q_scale, float k_scale, float v_scale, void* stream);
int lx_attn_fwd_fp8(const lx_attn_desc* d, float qk_descale, float v_descale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Precise mode -- fp32-class arithmetic for the reference's shipped fp32 configuration (train/config/seed_512.yaml:2):
 * GEMMs as split-bf16 products on the bf16 MFMA (lx_gemm_desc.k_segs), everything between them in fp32. The producers of the
 * hi/lo operand pairs and the fp32 attention live here; opt-in (model_config["precise"] / LxFluxTransformer(precise=True)).
 * ------------------------------------------------------------------------------------------------ */
/* dst(bf16)[m, k] = bf16(src[m, k]);  dst[m, lo_off + k] = bf16(src[m, k] - dst[m, k])  -- the operand pair of a k_segs >= 2 GEMM */
int lx_split_bf16(const float* src, int lds, void* dst, int ldd, int lo_off, int M, int K, void* stream);
/* lx_ln_modulate_segs with a split output: hi at column c, lo at column y_lo_off + c of Y (bf16, ldy >= y_lo_off + D) */
int lx_ln_modulate_split_segs(const float* X, int ldx, const lx_ln_seg* seg, int n_seg, int mod_ld, void* Y, int ldy, int y_lo_off,
                              int D, float eps, void* stream);
/* lx_qkv_prep_segs on an fp32 [M, ld] buffer: per-head RMSNorm + RoPE of q (q_col) and k (k_col) in place, in fp32
 * (block.py:38-41,60-67,74-78,92-99); v is left alone (the fp32 attention reads it row-major); vt_pos0 of the segments is ignored */
int lx_qkv_prep_f32_segs(float* QKV, int ld, int q_col, int k_col, const lx_qkv_seg* seg, int n_seg, int n_batches, int H, float eps,
                         void* stream);
/* Joint attention (the lx_attn_fwd contract: up to 3 token segments, additive (query segment, key segment) bias, -INFINITY masks
 * a pair) with fp32 q / k / v read from one [M, ld] buffer at q_col / k_col / v_col (head h at + h*128), exact fp32 products and
 * accumulation on v_mfma_f32_32x32x2_f32, fp32 online softmax. O (bf16, ldo) gets the output as a hi/lo pair: hi at
 * o_col + h*128 + d, lo o_lo_off columns further (o_lo_off = 0: hi only). */
typedef struct lx_attn_f32_desc {
  const float* QKV; int32_t ld, q_col, k_col, v_col;
  void* O; int32_t ldo, o_col, o_lo_off;
  int32_t B, H, n_seg;
  int32_t seg_row0[3], seg_len[3];
  float bias[3][3];
  float scale;
} lx_attn_f32_desc;
int lx_attn_fwd_f32(const lx_attn_f32_desc* d, void* stream);

/* Precise-mode attention on the bf16 matrix pipe (the default of precise mode since round 3; lx_attn_fwd_f32 stays as the exact
 * fp32-MFMA reference). Every operand is a bf16 PAIR x = x_hi + x_lo (x_hi = bf16(x), x_lo = bf16(x - x_hi): 16 mantissa bits) and
 * every product is evaluated as hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_bf16 into one fp32 accumulation: 3/16 of the
 * fp32-MFMA cost, relative error ~2^-16 per operand. Replaces F.scaled_dot_product_attention of block.py:129-131 in the reference's
 * shipped fp32 configuration (train/config/seed_512.yaml:2).
 *
 * lx_qkv_prep_split_segs: fp32 projections QKV [M, ld] (q at q_col, k at k_col, v at v_col; head h at + h*128) -> per-head RMSNorm +
 * RoPE of q and k in fp32 (block.py:38-41,60-67,74-78,92-99; segments as in lx_qkv_prep_segs), written as pairs into QK2 (bf16
 * [M, ld2]: q_hi at q2_col + h*128, q_lo lo_off columns further; k likewise at k2_col), and v as a pair of V^T images in the layout
 * lx_attn_fwd reads (VT2: bf16 [B*H, 128, vt_ld] with the 16-key interleave; the lo image vt_lo_off ELEMENTS after the hi image). */
int lx_qkv_prep_split_segs(const float* QKV, int ld, int q_col, int k_col, int v_col, const lx_qkv_seg* seg, int n_seg, int n_batches, int H,
                           float eps, void* QK2, int ld2, int q2_col, int k2_col, int lo_off, void* VT2, int vt_ld, long long vt_lo_off,
                           void* stream);
/* Joint attention (the lx_attn_fwd contract; n_qseg must be 0) on those pairs: d->Q / d->K / d->VT are the hi images (q_col, k_col,
 * ldq, ldk, vt_ld as in lx_attn_fwd), the lo images sit qk_lo_off columns / vt_lo_off elements further. fp32 online softmax with an
 * exact running maximum. d->O (bf16, ldo) gets the output as a pair: hi at o_col + h*128 + d, lo o_lo_off columns further (0: hi only). */
int lx_attn_fwd_split(const lx_attn_desc* d, int qk_lo_off, long long vt_lo_off, int o_lo_off, void* stream);

/* ------------------------------------------------------------------------------------------------
 * fp8 GEMM path (LX_OPERANDS_FP8; BASELINE configs[4], opt-in model_config["gemm_fp8"]): producers of the e4m3 operand images.
 * The reference has no fp8 path; the contract is "the bf16 result within the measured fp8 tolerance" (tests/test_fp8_gpu.py).
 * ------------------------------------------------------------------------------------------------ */
/* lx_ln_modulate_segs writing the bf16 operand Y (may be NULL) AND its e4m3 image Y8[m, c] = e4m3(y * y8_scale) (ldy8 in bytes) */
int lx_ln_modulate_fp8_segs(const float* X, int ldx, const lx_ln_seg* seg, int n_seg, int mod_ld, void* Y, int ldy, void* Y8, int ldy8,
                            float y8_scale, int D, float eps, void* stream);
/* dst(e4m3)[m, k] = src(bf16 | fp32)[m, k] * scale, saturated to +-448 */
int lx_convert_fp8(const void* src, int src_is_bf16, int lds, void* dst, int ldd, float scale, int M, int K, void* stream);
/* lx_lora_down with an e4m3 activation image: T_s = x_descale * X8[M, K_s] . Adown[R, K_s]^T (Adown bf16) */
int lx_lora_down_fp8(const void* X8, int ldx, float x_descale, const void* Adown, float* T, int ldt, int M, int K, int R, int n_split,
                     int split_stride, void* stream);

/* x(fp32) += dsigma * v   (FlowMatchEulerDiscreteScheduler.step, generate.py:349); v is bf16 or fp32 */
int lx_euler_step(float* x, const void* v, int v_is_bf16, float dsigma, size_t n, void* stream);
/* dst(fp32)[i] = src(fp32|bf16)[i] ; dst(bf16) = src(fp32) : layout plumbing between streams. dst_bf16: 0 fp32 | 1 bf16 | 2 IEEE fp16
 * (saturated to +-65504: the operand image of an LX_OPERANDS_F16 GEMM) */
int lx_convert(void* dst, int dst_bf16, const void* src, int src_bf16, size_t n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * CS3 (Cross-Scale State Space) encoder pieces -- src/train/model.py:16-373. fp32, [B,C,L] channel-major.
 * ------------------------------------------------------------------------------------------------ */
/* S4 SSM as a modal linear recurrence evaluated by a wavefront prefix scan (fp64 state):
 *   y[b,h,l] = Re sum_n w[h,n] * s_n[l] + D[h]*u[b,h,l],  s_n[l] = lam[h,n]*s_n[l-1] + u[b,h,l]
 * lam,w: complex fp64 as (re,im) pairs [H,N,2].  Equals s4torch's FFT convolution with kernel K[h,l]=Re sum_n w lam^l. */
int lx_s4_scan(const float* u, const double* lam, const double* w, const float* Dskip, float* y,
               int B, int H, int L, int N, void* stream);
/* Same operator as a direct causal convolution with the materialised kernel Kker[H,L] (cross-check / fallback). */
int lx_s4_conv(const float* u, const float* Kker, const float* Dskip, float* y, int B, int H, int L, void* stream);
/* Pointwise channel mix used by every S4 block and the S4 encoder/decoder Linear:
 *   z = W[Hout,Hin] . act(x[b,:,l]) + bias (+ resid[b,:,l]);  optional LayerNorm over Hout (gamma,beta,eps 1e-5)
 *   act: 0 none, 1 GELU(erf).  Hin,Hout <= 64. */
int lx_chanmix(const float* x, const float* W, const float* bias, const float* resid, const float* ln_g,
               const float* ln_b, float* y, int B, int Hin, int Hout, int L, int act, void* stream);
/* Multi-scale adaptive average pooling (FeaturePyramidPooling / nn.AdaptiveAvgPool1d bin rule):
 *   y[b,c, off_i + j] = mean x[b,c, floor(j L/s_i) : ceil((j+1) L/s_i)], rows of y have ldy elements, y_col0 offset. */
int lx_pyramid_pool(const float* x, float* y, int B, int C, int L, const int* sizes, int n_sizes, int ldy, int y_col0,
                    void* stream);
/* Row LayerNorm(gamma,beta,eps)+ReLU in place over fp32 rows (encoder projection heads, model.py:60-72). */
int lx_layernorm_relu(float* x, const float* g, const float* b, int M, int D, float eps, void* stream);
/* fp32 linear Y[M,N] (=|+=) X[M,K] . W[N,K]^T + bias for the encoder heads / fusion linears / LoRA glue.
 * x_trans != 0: X is stored [K, M] (ldx = its leading dim), i.e. channel-major activations [C,L] consumed per position;
 * y_trans != 0: Y is stored [N, M] (channel-major output). */
int lx_linear_f32(const float* X, int ldx, int x_trans, const float* W, int ldw, const float* bias, float* Y, int ldy,
                  int y_trans, int M, int N, int K, int accumulate, void* stream);

/* Channel-major fp32 GEMM on the f32-input MFMA (exact fp32 products):
 *   Y[b][n][l] = epi(sum_k W[n][k] * X[b][k][l] + bias[n]),  X: [B, K, L] (row stride ldx, batch stride x_bstride), Y: [B, N, L].
 * A Linear / 1x1 convolution over the channel axis at every position of a [B, C, L] activation without a transpose: fuse_eeg's
 * Linear(1024 -> 512) (model.py:731-755) and the DUAN gate's two convolutions (model.py:947-1035).
 * epilogue 0: store; 1: Y += ...; 2: ReLU; 3: sigmoid, then summed over each 64-position tile into
 * part[b][ceil(L/64)][N] (Y unused) -- deterministic, no atomics. */
int lx_chan_gemm_f32(const float* X, long x_bstride, int ldx, const float* W, int ldw, const float* bias, float* Y, long y_bstride,
                     int ldy, int B, int N, int K, int L, int epilogue, float* part, void* stream);

/* ------------------------------------------------------------------------------------------------
 * DGF (Dynamic Gated Fusion; `DUAN` in the reference, src/train/model.py:947-1035). fp32 [B,C,L].
 * gate: conv1x1 C->Hd (gw1 [Hd,C], gb1) ReLU conv1x1 Hd->C (gw2 [C,Hd], gb2) sigmoid, averaged over L;
 * mlp: conv1x1 C->Hd (mw1, mb1) ReLU conv1x1 Hd->2C (mw2 [2C,Hd], mb2) on the L-pooled condition.
 * ws: >= lx_duan_workspace_bytes(B,C,L,Hd) bytes of scratch.
 * ------------------------------------------------------------------------------------------------ */
size_t lx_duan_workspace_bytes(int B, int C, int L, int Hd);
int lx_duan_fwd(const float* x, const float* c, const float* gw1, const float* gb1, const float* gw2, const float* gb2,
                const float* mw1, const float* mb1, const float* mw2, const float* mb2, float* y,
                int B, int C, int L, int Hd, float eps, int keep_k, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * FLUX VAE (diffusers AutoencoderKL; reference call sites src/flux/generate.py:375-380 decode, src/flux/pipeline_tools.py:7-30
 * encode) -- SURVEY 8f.3, either side of the denoise loop. Convolutions run as implicit GEMMs on lx_gemm_bf16 (im2col rows x
 * [Cout, 9 Cin] weights); these are the NHWC row kernels around them. bf16 activations, fp32 statistics.
 * ------------------------------------------------------------------------------------------------ */
/* y(bf16)[b,p,c] = act(GroupNorm_G(x)[b,p,c] * gamma[c] + beta[c]); x: [B, P, C] fp32 or bf16, statistics per (b, group) over
 * P pixels x C/G channels in fp32, deterministic (no atomics); silu != 0: act = SiLU. ws: lx_groupnorm_workspace_bytes(). */
size_t lx_groupnorm_workspace_bytes(int B, int P, int G);
int lx_groupnorm_silu(const void* x, int x_is_bf16, int B, int P, int C, int G, const float* gamma, const float* beta, float eps,
                      int silu, void* y, void* ws, size_t ws_bytes, void* stream);
/* im2col for a 3x3 convolution over NHWC bf16 x[B,H,W,C]: out[(b,yo,xo), (dy*3+dx)*C + c], rows of Kpad >= 9C elements (zero
 * padded). mode 0: stride 1 / pad 1; mode 1: stride 2 / pad (0,1,0,1) (Downsample2D, padding=0); mode 2: nearest 2x upsample
 * folded into the gather, then stride 1 / pad 1 (Upsample2D + conv). */
int lx_im2col3x3(const void* x, int B, int H, int W, int C, int mode, void* out, int Kpad, void* stream);
/* P(bf16)[m, :] = softmax(scale * S(fp32)[m, :]) -- the single-head mid-block attention of the VAE */
int lx_softmax_rows(const float* S, int lds, float scale, void* P, int ldp, int M, int N, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LX_H_ */
