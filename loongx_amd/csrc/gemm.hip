// gemm.hip -- the GEMM planner (lx_gemm_bf16 / lx_gemm_bf16_ws, include/lx.h) and the 8-wave bf16-operand kernels.
// Kernel templates: gemm8.h (8 waves, two per SIMD) and gemm4.h (one wave per SIMD); the other operand formats are instantiated in
// gemm_f16.hip, gemm_modes.hip, gemm4.hip and gemm4_modes.hip and reached through the launchers of gemm_common.h.
#include "gemm8.h"


// ---- launch planning ---------------------------------------------------------------------------------------------
// One workgroup per CU, so a launch runs in "rounds" of 256 tiles. Per-round cost model calibrated on MI355X with
// rocprofv3 kernel durations over a K sweep (tools/gemm_ksweep.py): fixed (launch + prologue + epilogue burst) plus a
// per-64-deep-K-tile slope at full occupancy.
static double round_us(int bm, int K) { return bm == 256 ? 15.0 + 1.81 * (K / 64) : 10.5 + 1.06 * (K / 64); }

// runtime switches, read once per process (lx_gemm_reload_env() re-reads them): LX_GEMM_BM = 256 | 128 forces an 8-wave tile height,
// LX_GEMM4 = 0 | 1 | 2 and LX_GEMM4_SK = 0 | 1 | 2 (two-way only) select lx_gemm4_kernel / its split form (include/lx.h), LX_GEMM4_FAULT = 1 injects a
// partner time-out into the split form (tests). The A/B knobs of rounds 3-4 (pair kernel, one-grid switch, e4m3 q/k/v epilogue on
// the 4-wave kernel, tail divisor, round limit) are gone with the plans that lost: one plan per launch shape.
struct GemmEnv { int bm, g4, sk, g4_fault; };
static int env_int(const char* name, int dflt) {
  const char* s = getenv(name);
  return s ? atoi(s) : dflt;
}
static GemmEnv read_gemm_env() { return GemmEnv{env_int("LX_GEMM_BM", 0), env_int("LX_GEMM4", 1), env_int("LX_GEMM4_SK", 1), env_int("LX_GEMM4_FAULT", 0)}; }
static GemmEnv g_gemm_env = read_gemm_env();
static const GemmEnv& gemm_env() { return g_gemm_env; }
extern "C" void lx_gemm_reload_env(void) { g_gemm_env = read_gemm_env(); }

// ---- split-tile scratch of lx_gemm4_kernel, in a CALLER-PROVIDED workspace (lx_gemm_bf16_ws) ----------------------------------------
// [256 slots of 256 KiB (one parked 256 x 256 fp32 tile each) | 256 flags | error word + pad to 64 ints]. The library keeps no scratch
// of its own: whoever owns a stream owns its workspace, so launches on different streams never share slots or flags. The error word is
// the int 64 ints before the end (callers that poll it asynchronously read that position).
namespace {
constexpr size_t SK_FLAGS_OFF = (size_t)256 * SK_SLOT_FLOATS * sizeof(float);
constexpr size_t SK_WS_BYTES = SK_FLAGS_OFF + (256 + 64) * sizeof(int);
static_assert(SK_WS_BYTES % 256 == 0, "workspace size keeps the 256-byte alignment contract");
constexpr int SK_MIN_KT_ALL = 96;                       // K tiles from which a launch of <= 128 tiles runs entirely in the split form (K >= 6144)
constexpr int SK_MAX_ROUNDS = 16;                       // launches of up to this many rounds may split their tail

int device_cus() {
  static int n = -1;
  if (n < 0) {
    int dev = 0;
    hipDeviceProp_t pr;
    n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 0;
  }
  return n;
}
}  // namespace

static long tiles_of(const lx_gemm_desc& p, int bm) { return (long)((p.M + bm - 1) / bm) * ((p.N + BN - 1) / BN); }

static lx_gemm_desc sub_rows(const lx_gemm_desc& p, int r0, int rows) {   // rows [r0, r0+rows) of a problem
  lx_gemm_desc q = p;
  q.A = (const uint16_t*)p.A + (size_t)r0 * p.lda;
  const int epi = p.epilogue & 0xff;
  q.C = epi == LX_EPI_STORE_BF16 ? (void*)((uint16_t*)p.C + (size_t)r0 * p.ldc) : (void*)((float*)p.C + (size_t)r0 * p.ldc);
  if (p.lora_t) q.lora_t = p.lora_t + (size_t)r0 * p.lora_ldt;
  if (p.qkv_k) q.qkv_k = (uint16_t*)p.qkv_k + (size_t)r0 * p.qkv_k_ld;
  if (p.qkv_q8) {
    q.qkv_q8 = (uint8_t*)p.qkv_q8 + (size_t)r0 * p.qkv_ld8;
    q.qkv_k8 = (uint8_t*)p.qkv_k8 + (size_t)r0 * p.qkv_ld8;
  }
  q.M = rows;
  return q;
}

static int launch_plan(const GemmArgs& a, int bm, hipStream_t s, int variant = LX_GV_BF16) {
  if (a.n == 0) return LX_OK;
  if (variant == LX_GV_SPLIT) lx_gemm8_launch_split(bm, a, s);
  else if (variant == LX_GV_F16) lx_gemm8_launch_f16(bm, a, s);
  else if (variant == LX_GV_FP8) lx_gemm8_launch_fp8(bm, a, s);
  else lx_gemm8_launch_bf16(bm, a, s);
  LX_LAUNCH_CHECK("lx_gemm_bf16");
  return LX_OK;
}

// the 8-wave bf16 kernels live in this translation unit
void lx_gemm8_launch_bf16(int bm, const GemmArgs& a, hipStream_t s) {
  const int t = a.tile_start[a.n];
  if (bm == 256) hipLaunchKernelGGL((lx_gemm_kernel<256, false>), dim3(t), dim3(NTHREADS), 0, s, a);
  else hipLaunchKernelGGL((lx_gemm_kernel<128, false>), dim3(t), dim3(NTHREADS), 0, s, a);
}
void lx_gemm8_mixed_launch_bf16(const GemmArgs& big, const GemmArgs& tail, int n_big_pad, hipStream_t s) {
  hipLaunchKernelGGL(lx_gemm_mixed_kernel<false>, dim3(n_big_pad + tail.tile_start[tail.n]), dim3(NTHREADS), 0, s, big, tail, n_big_pad);
}

static void plan_add(GemmArgs& a, const lx_gemm_desc& p, int m_base, int bm) {
  a.p[a.n] = p;
  a.m_base[a.n] = m_base;
  a.tile_start[a.n + 1] = a.tile_start[a.n] + (int)tiles_of(p, bm);
  ++a.n;
  for (int i = a.n + 1; i <= MAX_SUB; ++i) a.tile_start[i] = a.tile_start[a.n];
}

extern "C" size_t lx_gemm_workspace_bytes(void) { return SK_WS_BYTES; }

extern "C" int lx_gemm_workspace_status(void* workspace, void* stream) {
  LX_CHECK_ARG(workspace, "lx_gemm_workspace_status: NULL workspace");
  int* err = (int*)((char*)workspace + SK_FLAGS_OFF) + 256;
  int v = 0;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemcpyAsync(&v, err, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
    lx_set_error("lx_gemm_workspace_status: reading the error word failed: %s", hipGetErrorString(hipGetLastError()));
    return LX_ERR_LAUNCH;
  }
  if (v == 0) return LX_OK;
  // a timed-out pair leaves flags behind: reset all of them with the error word so the workspace is usable again
  (void)hipMemsetAsync((char*)workspace + SK_FLAGS_OFF, 0, (256 + 64) * sizeof(int), s);
  (void)hipStreamSynchronize(s);
  lx_set_error("lx_gemm_bf16_ws: a split-K pair workgroup timed out waiting for its partner (CUs held by other work?); the results "
               "of that launch are invalid. Re-run with the workspace omitted (lx_gemm_bf16) or LX_GEMM4_SK=0");
  return LX_ERR_LAUNCH;
}

extern "C" int lx_gemm_bf16(const lx_gemm_desc* problems, int n, void* stream) { return lx_gemm_bf16_ws(problems, n, nullptr, 0, stream); }

extern "C" int lx_gemm_bf16_ws(const lx_gemm_desc* problems, int n, void* workspace, size_t ws_bytes, void* stream) {
  LX_CHECK_ARG(problems && n >= 1 && n <= LX_GEMM_MAX_GROUP, "lx_gemm_bf16: n=%d out of range [1,%d]", n, LX_GEMM_MAX_GROUP);
  long t256 = 0, t128 = 0;
  int kmax = 0;
  bool split = false, fp8 = false, f16 = false, qkv = false;
  for (int i = 0; i < n; ++i) fp8 = fp8 || (problems[i].epilogue & LX_OPERANDS_FP8) != 0;
  for (int i = 0; i < n; ++i) f16 = f16 || (problems[i].epilogue & LX_OPERANDS_F16) != 0;
  for (int i = 0; i < n; ++i) {
    const lx_gemm_desc& p = problems[i];
    if (f16) {
      LX_CHECK_ARG((p.epilogue & LX_OPERANDS_F16) != 0 && !(p.epilogue & LX_OPERANDS_FP8) && p.k_segs <= 1 && !(p.epilogue & LX_EPI_SPLIT_BF16),
                   "lx_gemm_bf16[%d]: LX_OPERANDS_F16 must be set on every problem of a launch and excludes the e4m3 and split-bf16 modes", i);
      if (p.f16_ovf) LX_CHECK_ARG(((uintptr_t)p.f16_ovf & 3) == 0, "lx_gemm_bf16[%d]: f16_ovf must be 4-byte aligned", i);
    }
    if (fp8) {
      LX_CHECK_ARG((p.epilogue & LX_OPERANDS_FP8) != 0 && p.k_segs <= 1 && !(p.epilogue & LX_EPI_SPLIT_BF16), "lx_gemm_bf16[%d]: LX_OPERANDS_FP8 must be set on every problem of a launch and excludes the split-bf16 mode", i);
      LX_CHECK_ARG(p.K % 128 == 0 && p.lda % 16 == 0 && p.ldw % 16 == 0, "lx_gemm_bf16[%d]: fp8 operands need K %% 128 == 0 and lda / ldw %% 16 == 0 (K=%d)", i, p.K);
      if (p.col_scale) LX_CHECK_ARG(((uintptr_t)p.col_scale & 15) == 0, "lx_gemm_bf16[%d]: col_scale must be 16-byte aligned", i);
      if (p.lora_t) LX_CHECK_ARG(p.lora_r <= 4 && p.lora_nsplit <= 4, "lx_gemm_bf16[%d]: the fp8 path applies LoRA in the tile prologue only (rank <= 4, <= 4 slabs)", i);
      if ((p.epilogue & 0xff) == LX_EPI_STORE_FP8) LX_CHECK_ARG(p.out_scale > 0.f && p.ldc % 8 == 0, "lx_gemm_bf16[%d]: LX_EPI_STORE_FP8 needs out_scale > 0 and ldc %% 8 == 0", i);
    }
    LX_CHECK_ARG(p.k_segs >= 0 && p.k_segs <= 3, "lx_gemm_bf16[%d]: k_segs=%d must be 0..3", i, p.k_segs);
    const int segs = p.k_segs > 1 ? p.k_segs : 1;
    if (segs > 1 || (p.epilogue & LX_EPI_SPLIT_BF16)) split = true;
    if (segs > 1) LX_CHECK_ARG(p.a_lo_off >= p.K && p.a_lo_off % 8 == 0 && p.lda >= p.a_lo_off + p.K, "lx_gemm_bf16[%d]: a_lo_off=%d needs K <= a_lo_off, a_lo_off + K <= lda, multiple of 8", i, p.a_lo_off);
    if (segs == 3) LX_CHECK_ARG(p.ldw >= 2 * p.K, "lx_gemm_bf16[%d]: k_segs = 3 reads W as [N, 2K] = [W_hi | W_lo]: ldw=%d < 2K", i, p.ldw);
    if (p.epilogue & LX_EPI_SPLIT_BF16) LX_CHECK_ARG((p.epilogue & 0xff) == LX_EPI_STORE_BF16 && p.c_lo_off >= p.N && p.c_lo_off % 8 == 0 && p.ldc >= p.c_lo_off + p.N,
                                                     "lx_gemm_bf16[%d]: LX_EPI_SPLIT_BF16 needs a bf16 store and N <= c_lo_off, c_lo_off + N <= ldc, multiple of 8", i);
    LX_CHECK_ARG(p.A && p.W && p.C, "lx_gemm_bf16[%d]: NULL operand", i);
    LX_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "lx_gemm_bf16[%d]: bad shape M=%d N=%d K=%d", i, p.M, p.N, p.K);
    LX_CHECK_ARG(p.K % BK == 0, "lx_gemm_bf16[%d]: K=%d must be a multiple of %d", i, p.K, BK);
    LX_CHECK_ARG(p.N % 8 == 0, "lx_gemm_bf16[%d]: N=%d must be a multiple of 8", i, p.N);
    LX_CHECK_ARG(p.lda % 8 == 0 && p.ldw % 8 == 0 && p.lda >= p.K && p.ldw >= p.K, "lx_gemm_bf16[%d]: lda/ldw must be >= K and multiples of 8", i);
    LX_CHECK_ARG(((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.W & 15) == 0 && ((uintptr_t)p.C & 15) == 0, "lx_gemm_bf16[%d]: operands must be 16-byte aligned", i);
    LX_CHECK_ARG(p.ldc % 8 == 0 && p.ldc >= p.N, "lx_gemm_bf16[%d]: ldc=%d must be >= N and a multiple of 8", i, p.ldc);
    const int epi = p.epilogue & 0xff;
    LX_CHECK_ARG(epi >= LX_EPI_STORE_BF16 && (epi <= LX_EPI_RESID_F32 || (fp8 && epi == LX_EPI_STORE_FP8)), "lx_gemm_bf16[%d]: unknown epilogue %d", i, p.epilogue);
    LX_CHECK_ARG(p.rows_per_batch > 0, "lx_gemm_bf16[%d]: rows_per_batch must be > 0", i);
    if (p.epilogue & LX_W_TILED) LX_CHECK_ARG(p.N % BN == 0 && p.ldw == (segs == 3 ? 2 * p.K : p.K), "lx_gemm_bf16[%d]: LX_W_TILED needs N %% 256 == 0 and ldw == K (2K with k_segs = 3)", i);
    if (p.gate) LX_CHECK_ARG(p.gate_ld >= p.N && p.gate_ld % 4 == 0, "lx_gemm_bf16[%d]: gate_ld=%d", i, p.gate_ld);
    if (p.lora_t) {
      LX_CHECK_ARG(p.lora_up && p.lora_r >= 1 && p.lora_r <= 16, "lx_gemm_bf16[%d]: LoRA needs lora_up and 1 <= r <= 16", i);
      LX_CHECK_ARG(p.lora_mod_cols <= 0 || p.lora_mod_cols % BN == 0, "lx_gemm_bf16[%d]: lora_mod_cols must be a multiple of %d", i, BN);
    }
    if (p.epilogue & LX_EPI_QKV) {
      LX_CHECK_ARG(!fp8 && segs == 1 && !(p.epilogue & LX_EPI_SPLIT_BF16) && epi == LX_EPI_STORE_BF16, "lx_gemm_bf16[%d]: LX_EPI_QKV goes with plain bf16 operands and LX_EPI_STORE_BF16", i);
      LX_CHECK_ARG(p.qkv_d > 0 && p.qkv_d % BN == 0 && (p.N >= 3 * p.qkv_d || p.N % BN == 0), "lx_gemm_bf16[%d]: LX_EPI_QKV needs qkv_d %% 256 == 0 and whole projection tiles (qkv_d=%d N=%d)", i, p.qkv_d, p.N);
      LX_CHECK_ARG(p.rows_per_batch % 32 == 0 && p.M % 32 == 0, "lx_gemm_bf16[%d]: LX_EPI_QKV needs rows_per_batch %% 32 == 0 and M %% 32 == 0 (%d, %d)", i, p.rows_per_batch, p.M);
      const bool q8 = p.qkv_q8 != nullptr;
      LX_CHECK_ARG(p.qkv_norm_q && p.qkv_norm_k && (p.qkv_vt || q8) && p.qkv_rope && ((uintptr_t)p.qkv_norm_q & 15) == 0 && ((uintptr_t)p.qkv_norm_k & 15) == 0 && ((uintptr_t)p.qkv_vt & 15) == 0 &&
                   ((uintptr_t)p.qkv_rope & 15) == 0, "lx_gemm_bf16[%d]: LX_EPI_QKV needs 16-byte aligned qkv_norm_q / qkv_norm_k / qkv_rope / qkv_vt", i);
      if (q8) {
        LX_CHECK_ARG(p.qkv_k8 && p.qkv_vt8 && (((uintptr_t)p.qkv_q8 | (uintptr_t)p.qkv_k8 | (uintptr_t)p.qkv_vt8) & 15) == 0 && p.qkv_ld8 % 16 == 0 && p.qkv_ld8 >= p.qkv_d,
                     "lx_gemm_bf16[%d]: LX_EPI_QKV e4m3 outputs need 16-byte aligned qkv_q8 / qkv_k8 / qkv_vt8 and qkv_ld8 %% 16 == 0, >= qkv_d", i);
        LX_CHECK_ARG(p.qkv_q_scale > 0.f && p.qkv_k_scale > 0.f && p.qkv_v_scale > 0.f && !p.qkv_k, "lx_gemm_bf16[%d]: LX_EPI_QKV e4m3 outputs need positive scales and no separate bf16 key image", i);
      }
      LX_CHECK_ARG(p.qkv_vt_ld > 0 && p.qkv_vt_ld % 64 == 0 && p.qkv_vt_pos0 >= 0 && p.qkv_vt_pos0 % 64 == 0 && p.qkv_vt_pos0 + p.rows_per_batch <= p.qkv_vt_ld,
                   "lx_gemm_bf16[%d]: qkv_vt_ld / qkv_vt_pos0 must be multiples of 64 with the stream inside a V^T row", i);
      if (p.qkv_k) LX_CHECK_ARG(((uintptr_t)p.qkv_k & 15) == 0 && p.qkv_k_ld % 8 == 0 && p.qkv_k_ld >= p.qkv_d, "lx_gemm_bf16[%d]: qkv_k must be 16-byte aligned with qkv_k_ld %% 8 == 0, >= qkv_d", i);
      qkv = true;
    }
    if (p.bias) LX_CHECK_ARG(((uintptr_t)p.bias & 15) == 0, "lx_gemm_bf16[%d]: bias must be 16-byte aligned", i);
    t256 += tiles_of(p, 256);
    t128 += tiles_of(p, 128);
    kmax = p.K * segs > kmax ? p.K * segs : kmax;
  }
  hipStream_t s = (hipStream_t)stream;
  if (fp8) {     // e4m3 operands: all 256-row tiles or all 128-row tiles
    const int ncu = device_cus() > 0 ? device_cus() : 256;
    const double ca = (double)((t256 + ncu - 1) / ncu) * round_us(256, kmax / 4), cb = (double)((t128 + ncu - 1) / ncu) * round_us(128, kmax / 4);
    const int bm = gemm_env().bm ? gemm_env().bm : (cb < ca ? 128 : 256);
    GemmArgs all;
    all.n = 0;
    all.tile_start[0] = 0;
    for (int i = 1; i <= MAX_SUB; ++i) all.tile_start[i] = 0;
    for (int i = 0; i < n; ++i) plan_add(all, problems[i], 0, bm);
    return launch_plan(all, bm, s, LX_GV_FP8);
  }
  const int NCU = device_cus() > 0 ? device_cus() : 256;   // one workgroup per CU: a launch runs in rounds of NCU tiles
  const GemmEnv& env = gemm_env();
  const int forced = env.bm;      // 256 | 128 | 0 = plan
  // lx_gemm4_kernel (one wave per SIMD, 256-row tiles only): the launches whose epilogue it has and whose tile count fills whole rounds.
  // LX_GEMM4 = 0 never | 1 (default) where the last round is at least 3/4 full or there are >= 8 rounds | 2 whenever the epilogue allows (tests).
  if (env.g4 && forced == 0 && (workspace || env.g4 == 2)) {     // (no workspace = the batch-size-invariant plans only)
    bool ok = true;
    for (int i = 0; i < n; ++i) {
      const lx_gemm_desc& p = problems[i];
      ok = ok && (p.epilogue & 0xff) <= LX_EPI_RESID_F32 && p.K / BK >= 2 && !((p.epilogue & LX_EPI_QKV) && p.qkv_q8);   // (e4m3 q/k/v images stay on the 8-wave kernels: measured 0.2 % slower per image here at 1024 x 1024 -- 8-byte V^T stores per lane against 16 -- profiles/r04f_attnfp8_ab.txt, and removed from this kernel in round 5)
      if (p.lora_t)        // the kernel's LoRA step: two ranks per 8-byte load, one 32-deep MFMA k-step, up to four K-split slabs
        ok = ok && p.lora_r <= 8 && p.lora_r % 2 == 0 && p.lora_nsplit >= 1 && p.lora_nsplit <= 4 && p.lora_ldt % 2 == 0 && p.lora_split_stride % 2 == 0 &&
             ((((uintptr_t)p.lora_t) | ((uintptr_t)p.lora_up)) & 7) == 0;
    }
    const long rounds = (t256 + NCU - 1) / NCU;
    bool fills = env.g4 == 2 || (t256 >= NCU && (rounds * NCU - t256 <= NCU / 4 || rounds >= 8));
    // split form (LX_GEMM4_SK = 1 default | 0 off): the tiles of a partial last round, or all tiles of a launch with <= 128 of them and
    // a long K, by two workgroups each (half of K), meeting through the caller's workspace. One K for the whole launch, >= 16 K tiles.
    // (launches of up to LX_GEMM4_SK_ROUNDS = 16 rounds: 7 until round 4 -- at batch 16 the N = 3072 projections are 1920 tiles = 7.5 rounds,
    //  whole tiles paid the half-empty eighth round: configs[2] 1.1318 / 1.1288 -> 1.1419 / 1.1418 images/s, profiles/r04aa_*)
    bool uniform_k4 = true;
    for (int i = 1; i < n; ++i) uniform_k4 = uniform_k4 && problems[i].K == problems[0].K;
    const long tail = t256 % NCU, full = t256 - tail;
    for (int i = 1; i < n; ++i) uniform_k4 = uniform_k4 && problems[i].k_segs == problems[0].k_segs;
    const int kt_all = kmax / BK;                      // K tiles of a tile, all segments of a split-bf16 problem counted
    const bool can_split = env.sk && workspace && ws_bytes >= SK_WS_BYTES && ((uintptr_t)workspace & 255) == 0 && uniform_k4 && kt_all >= 16 &&
                           tail > 0 && tail * 2 <= 256 && tail * 2 <= NCU && rounds <= SK_MAX_ROUNDS;
    bool split_all = can_split && full == 0 && kt_all >= SK_MIN_KT_ALL;       // (long-K launches of <= 128 tiles: ff2 / single proj_out at batch 1)
    // a tail of up to a third of a round; up to half a round where the exchange is a small part of the tile: long K (>= 96 K tiles) and no
    // q/k/v epilogue in the launch (the 1024 x 1024 batch-4 N = 3072 projections, 1632 tiles = 6 rounds + 96: 0.2594 / 0.2584 -> 0.2651 /
    // 0.2650 images/s, profiles/r04z_*; the double blocks' q/k/v launch at batch 1, 104 tail tiles at K = 3072: 146 vs 136 us for the
    // 8-wave mixed plan's half-height tiles -- stays there).
    const int tail_div = !qkv && kt_all >= 96 ? 2 : 3;
    bool split_tail = can_split && full > 0 && tail * tail_div <= NCU;      // (a tail of more than a third of a round: the 8-wave mixed plan's half-height tiles win -- the double blocks' q/k/v launch, 104 tail tiles: 154 vs 136 us)
    // three workgroups per split tile (thirds of K, round 5) where 3 x the split tiles fit one round AND the K loop is what the tail
    // costs: long K (>= 96 K tiles: the step-invariant-condition forwards' ff2 / proj_out, 72 tiles: 1.474 -> 1.553 images/s), or a
    // small tail (up to a sixth of a round: 2 ... 32 tiles at K = 3072: -3 ... -5 us per launch; at 72 tiles the second partner's
    // sums and the 216 workgroups' operand traffic cost what the shorter loop saves: 270.3 vs 269.5 us, tools/gemm_tail_cost.py).
    // bf16 / fp16 kernels; LX_GEMM4_SK = 2 keeps the two-way form everywhere (A/B).
    int np = 2;
    if (!split && env.sk != 2 && kt_all >= 24 && tail * 3 <= 256 && tail * 3 <= NCU && (split_all || split_tail) && (kt_all >= 96 || tail * 6 <= NCU)) np = 3;
    if (split) {
      // precise mode (two or three passes over K: the K-independent cost of a round and of the exchange weigh a third as much as on the
      // bf16 path; no q/k/v epilogue, no mixed plan to compete with): by cost -- measured slopes per K tile and round, 1.31 us for this
      // kernel (8.9 fixed, ~6 for an exchange), 1.55 / 1.06 for the 8-wave kernels at 256 / 128 rows (tools/gemm_ksweep.py)
      const double r4 = 8.9 + 1.31 * kt_all;
      const bool sk_tail = can_split && tail > 0;
      const double c4 = (double)(full / NCU) * r4 + (tail == 0 ? 0.0 : sk_tail ? 8.9 + 1.31 * (kt_all - kt_all / 2) + 6.0 : r4);
      const double c8a = (double)((t256 + NCU - 1) / NCU) * (5.8 + 1.55 * kt_all), c8b = (double)((t128 + NCU - 1) / NCU) * (10.5 + 1.06 * kt_all);
      fills = env.g4 == 2 || c4 < (c8a < c8b ? c8a : c8b);
      split_all = fills && sk_tail && full == 0;
      split_tail = fills && sk_tail && full > 0;
      if (!fills) split_all = split_tail = false;
    }
    if (ok && (fills || split_all || split_tail)) {
      GemmArgs all;
      all.n = 0;
      all.tile_start[0] = 0;
      for (int i = 1; i <= MAX_SUB; ++i) all.tile_start[i] = 0;
      for (int i = 0; i < n; ++i) plan_add(all, problems[i], 0, 256);
      float* slots = nullptr;
      int *flags = nullptr, *err = nullptr;
      unsigned grid = (unsigned)t256;
      int sk_full = (int)t256, sk_parts = 1;
      if (split_all || split_tail) {
        slots = (float*)workspace;
        flags = (int*)((char*)workspace + SK_FLAGS_OFF);
        err = flags + 256;
        grid = (unsigned)(full + np * tail);
        sk_full = (int)full;
        sk_parts = env.g4_fault ? 3 : 2;
      }
      if (split) lx_gemm4_launch_split(all, grid, sk_full, sk_parts, slots, flags, err, s);
      else if (f16) lx_gemm4_launch_f16(all, grid, sk_full, sk_parts, slots, flags, err, s, np);
      else lx_gemm4_launch_bf16(all, grid, sk_full, sk_parts, slots, flags, err, s, np);
      LX_LAUNCH_CHECK("lx_gemm_bf16");
      return LX_OK;
    }
  }
  if (split) {   // precise mode on the 8-wave kernels: all 256-row tiles or all 128-row tiles (no mixed / pair plans)
    const double ca = (double)((t256 + NCU - 1) / NCU) * round_us(256, kmax), cb = (double)((t128 + NCU - 1) / NCU) * round_us(128, kmax);
    const int bm = forced ? forced : (cb < ca ? 128 : 256);
    GemmArgs all;
    all.n = 0;
    all.tile_start[0] = 0;
    for (int i = 1; i <= MAX_SUB; ++i) all.tile_start[i] = 0;
    for (int i = 0; i < n; ++i) plan_add(all, problems[i], 0, bm);
    return launch_plan(all, bm, s, LX_GV_SPLIT);
  }
  if (workspace) LX_CHECK_ARG(ws_bytes >= SK_WS_BYTES && ((uintptr_t)workspace & 255) == 0, "lx_gemm_bf16_ws: workspace needs %zu bytes (lx_gemm_workspace_bytes()), 256-byte aligned", SK_WS_BYTES);
  const int variant = f16 ? LX_GV_F16 : LX_GV_BF16;
  // candidate schedules: all 256-row tiles, all 128-row tiles, or full rounds of 256-row tiles + a 128-row-tile tail
  const double c_a = (double)((t256 + NCU - 1) / NCU) * round_us(256, kmax);
  const double c_b = (double)((t128 + NCU - 1) / NCU) * round_us(128, kmax);
  GemmArgs big, tail;
  big.n = tail.n = 0;
  big.tile_start[0] = tail.tile_start[0] = 0;
  for (int i = 1; i <= MAX_SUB; ++i) big.tile_start[i] = tail.tile_start[i] = 0;
  double c_c = 1e30;
  const long full = (t256 / NCU) * NCU;
  if (forced == 0 && full > 0 && t256 > full) {
    // peel 256-row tile-rows off the ends of the problems (last problem first) until the big launch fits in full rounds
    long remain = t256;
    int keep_rows[LX_GEMM_MAX_GROUP];
    for (int i = 0; i < n; ++i) keep_rows[i] = problems[i].M;
    for (int i = n - 1; i >= 0 && remain > full; --i) {
      const int tn = (problems[i].N + BN - 1) / BN;
      while (remain > full && keep_rows[i] > 0) {
        const int last_rows = keep_rows[i] % 256 ? keep_rows[i] % 256 : 256;
        keep_rows[i] -= last_rows;
        remain -= tn;
      }
    }
    long tail_tiles = 0;
    for (int i = 0; i < n; ++i) {
      if (keep_rows[i] > 0) plan_add(big, sub_rows(problems[i], 0, keep_rows[i]), 0, 256);
      if (keep_rows[i] < problems[i].M) {
        const lx_gemm_desc q = sub_rows(problems[i], keep_rows[i], problems[i].M - keep_rows[i]);
        plan_add(tail, q, keep_rows[i], 128);
        tail_tiles += tiles_of(q, 128);
      }
    }
    c_c = (double)((remain + NCU - 1) / NCU) * round_us(256, kmax) + (double)((tail_tiles + NCU - 1) / NCU) * round_us(128, kmax) + 2.0;
  }
  int choice = forced == 256 ? 0 : forced == 128 ? 1 : (c_c < c_a && c_c < c_b) ? 2 : (c_b < c_a ? 1 : 0);
  if (choice == 2) {
    if (big.n > 0 && tail.n > 0) {                // both tile heights in ONE grid (+0.6 % per step against two launches)
      const int n_big = big.tile_start[big.n], n_big_pad = (n_big + 7) & ~7;
      if (f16) lx_gemm8_mixed_launch_f16(big, tail, n_big_pad, s);
      else lx_gemm8_mixed_launch_bf16(big, tail, n_big_pad, s);
      LX_LAUNCH_CHECK("lx_gemm_bf16");
      return LX_OK;
    }
    const int rc = launch_plan(big, 256, s, variant);
    if (rc != LX_OK) return rc;
    return launch_plan(tail, 128, s, variant);
  }
  GemmArgs all;
  all.n = 0;
  all.tile_start[0] = 0;
  for (int i = 1; i <= MAX_SUB; ++i) all.tile_start[i] = 0;
  for (int i = 0; i < n; ++i) plan_add(all, problems[i], 0, choice == 0 ? 256 : 128);
  return launch_plan(all, choice == 0 ? 256 : 128, s, variant);
}

