"""Full-depth parity (SURVEY 8d last row): the engine against the fp32 oracle at 19 + 38 full-width
blocks, S = 2560, over the whole 28-step trajectory, on identical weights and inputs -- oracle/parity.py is the harness, the
same one bench.py prints as `parity`.

Stated tolerances (the north star asks 1e-3 rel-err):
  * bf16 mode (the throughput mode: bf16 MFMA operands, fp32 accumulate, fp32 residual stream): what it MEASURES on an MI355X is
    recorded in DESIGN.md section 4; asserted here at the north star's 1e-3 on the edited latents and 6e-3 per forward;
  * fp16 operand mode (model_config["operands"] = "fp16" / dtype=torch.float16: fp16 GEMM operand images on v_mfma_f32_*_f16, bf16
    attention operands; round 5): <= 1e-3 PER FORWARD -- the north star's figure at the bf16 mode's matrix rate -- and <= 2e-4 on the
    edited latents, asserted;
  * precise mode (model_config / LxFluxTransformer(precise=True): split-bf16 MFMA GEMMs + fp32 attention): <= 1e-3, asserted.
"""
import json

import pytest
import torch

pytestmark = pytest.mark.gpu

# measured on MI355X (rounds 2-4): noise_pred 4.5e-3 ... 5.3e-3 per forward, final latents 8.0e-4 ... 8.6e-4.
# The final-latent bound IS the north star's 1e-3 (the measured margin is 14 %); the per-forward bound sits 13 % above the largest
# value seen. Where the 4.6e-3 per forward comes from is measured (tools/bf16_ablation.py, profiles/r04c_bf16_ablation.txt): the bf16
# A operands of the GEMMs, every kind in proportion to its share of the flops (ff1 2.3e-3, ff2 2.3e-3, single proj_out 1.7e-3, fused
# single projection 1.5e-3, to_out 1.1e-3, q/k/v 3.5e-4 alone; attention's q / k / v / P together 4.1e-4), adding in quadrature -- no
# subset cheaper than precise mode brings it under 2e-3.
from loongx_amd.tolerances import TOLERANCES  # noqa: E402  (the stated numbers live in ONE module: bench.py stamps its legs with the same ones)

BF16_NOISE_PRED_MAX = TOLERANCES["bf16"]["per_forward_max"]          # 6.0e-3
BF16_FINAL_LATENT = TOLERANCES["bf16"]["final"]                      # 1.0e-3


def test_full_depth_parity_bf16():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle.parity import full_depth_parity
    rec = full_depth_parity("cuda:0", steps=28, every=3)
    print("PARITY_BF16 " + json.dumps(rec))
    assert rec["noise_pred_relerr_max"] < BF16_NOISE_PRED_MAX, rec
    assert rec["final_latent_relerr"] < BF16_FINAL_LATENT, rec
    assert rec["final_latent_cosine"] > 0.9995, rec


@pytest.mark.parametrize("brain", ["eeg", "all"])
def test_full_depth_parity_bf16_with_the_brain_side(brain):
    """The composition bench.py times, at full depth: raw signals -> the product's CS3 encoders (+ DGF fusion for "all") -> its
    57-block DiT over 28 steps, against oracle/cs3.py -> oracle/flux_ref.py. "eeg" = BASELINE configs[1] (EEG-only, per-stream
    rule), "all" = configs[2]'s four modalities with fuse_flag=True. Same bounds as the DiT-only run; the encoders alone are fp32."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle.parity import full_depth_parity
    rec = full_depth_parity("cuda:0", steps=28, every=9, brain=brain)
    print(f"PARITY_BF16_BRAIN_{brain} " + json.dumps(rec))
    assert rec["brain_embeds_relerr"] is not None and rec["brain_embeds_relerr"] < 5e-6, rec      # measured 1.1e-6 (eeg) / 9e-8 (all): fp32 kernels
    assert rec["noise_pred_relerr_max"] < BF16_NOISE_PRED_MAX, rec
    assert rec["final_latent_relerr"] < BF16_FINAL_LATENT and rec["final_latent_cosine"] > 0.9995, rec


def test_full_depth_parity_independent_condition():
    """model_config["independent_condition"] (a reference option: block.py:115-120 masks the condition queries from text / image keys) is a
    SUPPORTED product mode with its own execution plan: the condition stream is step-invariant, so the engine computes it in the first
    denoise step and keeps its keys / values per layer (27 of 28 steps run 1536 instead of 2560 rows; the long-K projections of those steps
    take lx_gemm4_kernel's three-way split form). Full depth, 28 steps, against the oracle run with the same option: the bf16 mode's stated
    bounds -- the cached forwards (steps 1..27, teacher-forced at 9, 18, 27) included."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle.parity import full_depth_parity
    rec = full_depth_parity("cuda:0", steps=28, every=9, model_config={"union_cond_attn": True, "independent_condition": True})
    print("PARITY_BF16_INDEPENDENT_CONDITION " + json.dumps(rec))
    assert rec["noise_pred_relerr_max"] < BF16_NOISE_PRED_MAX, rec
    assert rec["final_latent_relerr"] < BF16_FINAL_LATENT and rec["final_latent_cosine"] > 0.9995, rec


# ---- fp16 operand images (round 5) -----------------------------------------------------------------------------------------------
# tools/bf16_ablation.py (profiles/r05a_fp16_ablation.txt) predicts 7.0e-4 per forward for fp16 GEMM A operands + bf16 attention operands
# from the fp32 oracle alone; the north star's bound is 1e-3 per forward.
F16_NOISE_PRED_MAX = TOLERANCES["fp16"]["per_forward_max"]           # 1.0e-3
F16_FINAL_LATENT = TOLERANCES["fp16"]["final"]                       # 2.0e-4


@pytest.mark.parametrize("brain", [None, "eeg"])
def test_full_depth_parity_fp16_operands(brain):
    """The north star's 1e-3 per velocity prediction, at full depth, in a mode that runs at the matrix rate of the bf16 mode: DiT alone,
    and BASELINE configs[1]'s composition (EEG -> CS3 encoder -> per-stream replacement -> 57 blocks x 28 steps). No operand image may
    have saturated (fp16's range) on the way."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle.parity import full_depth_parity
    rec = full_depth_parity("cuda:0", steps=28, every=3 if brain is None else 9, model_config={"union_cond_attn": True, "operands": "fp16"}, brain=brain)
    print(f"PARITY_FP16_{brain} " + json.dumps(rec))
    assert rec["noise_pred_relerr_max"] < F16_NOISE_PRED_MAX, rec
    assert rec["final_latent_relerr"] < F16_FINAL_LATENT and rec["final_latent_cosine"] > 0.99999, rec
    assert rec["f16_saturated_waves"] == 0 and rec["f16_weights_inexact_share"] < 1e-3, rec


# ---- statistics of a trained checkpoint (round 5) --------------------------------------------------------------------------------------
# Every figure above is on N(0, 0.02^2) weights with unit q / k norms and zero biases: all 57 layers then run the bounded-score attention
# kernel and no activation is far from its row's typical size. oracle.parity.realistic_stats_ adds what a real checkpoint has: per-layer
# q / k norm gains that put about half of the layers over the score bound (a MIXED bounded / max-tracking plan in one step, peaked softmax
# where the gain is large), outlier channels in the residual stream (x100-x1000) and in every MLP hidden layer, non-zero biases everywhere.
# Bounds = 2x what the MI355X measured (round 5, profiles/r05g_parity_realistic.txt), per mode.
# Measured: bf16 2.48e-3 per forward (max) / 3.9e-4 final latents; fp16 3.05e-4 / 3.9e-5; precise 4e-6 / 1e-6; 35 of 57 layers bounded.
REALISTIC_BOUNDS = {m: (TOLERANCES["realistic_" + m]["per_forward_max"], TOLERANCES["realistic_" + m]["final"]) for m in ("bf16", "fp16", "precise")}
# = {"bf16": (5.0e-3, 8.0e-4), "fp16": (6.5e-4, 1.0e-4), "precise": (2.0e-5, 5.0e-6)}      (per forward max, final latents)


@pytest.mark.parametrize("mode", ["bf16", "fp16", "precise"])
def test_full_depth_parity_realistic_stats(mode):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle.parity import full_depth_parity
    mc = {"union_cond_attn": True}
    if mode == "fp16":
        mc["operands"] = "fp16"
    rec = full_depth_parity("cuda:0", steps=28, every=9, model_config=mc, precise=mode == "precise", realistic=True)
    print(f"PARITY_REALISTIC_{mode} " + json.dumps(rec))
    bl = rec["bounded_score_layers"]
    if mode != "precise" or bl["layers"]:
        assert 0 < bl["bounded"] < bl["layers"] == 57, rec                  # the plan IS mixed: both attention kernels run in every step
    per_fwd, final = REALISTIC_BOUNDS[mode]
    assert rec["noise_pred_relerr_max"] < per_fwd and rec["final_latent_relerr"] < final, rec
    assert rec["final_latent_cosine"] > 0.9995, rec
    if mode == "fp16":
        assert rec["f16_saturated_waves"] == 0, rec                         # outliers of 900 / 200 are far inside fp16's range


# ---- BASELINE configs[4]'s mode: fp8 (e4m3) attention, bf16 GEMMs ------------------------------------------------------------------
# The reference has no fp8 path (block.py:129 is plain SDPA), so the contract is the bf16 result within a STATED tolerance:
#   <= 1e-2 per velocity prediction on average over the trajectory (<= 1.1e-2 at any single step), <= 2e-3 on the final latents
#   (full depth, against the fp32 oracle).
# Measured on MI355X (round 3): text-embedding conditioning 8.1e-3 mean / 8.9e-3 max per forward, 1.6e-3 final latents; with the
# conditioning coming from the CS3 encoders + DGF fusion (brain="all", profiles/r03k_bench_line.json) 9.4e-3 / 1.02e-2 / 1.7e-3;
# 1024x1024: 6.3e-3 / 6.6e-3 / 1.4e-3.
# The e4m3 GEMMs (model_config gemm_fp8, `bench.py --fp8`) do NOT hold it -- 1.0e-1 per forward, whatever the scaling recipe
# (tools/fp8_ablation.py, profiles/r03a_fp8_ablation.json) -- and are kept as an explicitly lossy option.
FP8_ATTN_NOISE_PRED_MEAN = TOLERANCES["attn_fp8"]["per_forward_mean"]      # 1.0e-2
FP8_ATTN_NOISE_PRED_MAX = TOLERANCES["attn_fp8"]["per_forward_max"]        # 1.1e-2
FP8_ATTN_FINAL_LATENT = TOLERANCES["attn_fp8"]["final"]                    # 2.0e-3


@pytest.mark.parametrize("brain", [None, "all"])
def test_full_depth_parity_fp8_attention_512(brain):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle.parity import full_depth_parity
    rec = full_depth_parity("cuda:0", steps=28, every=3, model_config={"union_cond_attn": True, "attn_fp8": True}, brain=brain)
    print(f"PARITY_FP8ATTN_512_{brain} " + json.dumps(rec))
    assert rec["noise_pred_relerr_mean"] <= FP8_ATTN_NOISE_PRED_MEAN and rec["noise_pred_relerr_max"] <= FP8_ATTN_NOISE_PRED_MAX, rec
    assert rec["final_latent_relerr"] <= FP8_ATTN_FINAL_LATENT and rec["final_latent_cosine"] > 0.99999, rec


def test_full_depth_parity_fp8_attention_1024_precheck_8_steps():
    """A quick pre-check of the shape configs[4] names (1024x1024, S = 8704, all 57 blocks) on an 8-step schedule: the per-forward figures
    do not depend on the schedule and are held to the stated bounds; the free-running latents do (eight steps of 1/8 carry each
    forward's error further than 28 of 1/28: measured 2.2e-3), so THEIR stated bound is asserted by the 28-step test below, not here."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle.parity import full_depth_parity
    rec = full_depth_parity("cuda:0", steps=8, every=2, hw=64, model_config={"union_cond_attn": True, "attn_fp8": True})
    print("PARITY_FP8ATTN_1024_8STEP " + json.dumps(rec))
    assert rec["noise_pred_relerr_mean"] <= FP8_ATTN_NOISE_PRED_MEAN and rec["noise_pred_relerr_max"] <= FP8_ATTN_NOISE_PRED_MAX, rec
    assert rec["final_latent_cosine"] > 0.99999, rec


def test_full_depth_parity_fp8_attention_1024():
    """The shape configs[4] names -- 1024x1024 (S = 8704), all 57 blocks -- on the metric's 28-step schedule, held to the STATED bounds
    (round 5 had cut this to 8 steps with widened bounds to save 33 s; the stated numbers are asserted again): teacher-forced comparison
    at every 3rd step + the free-running loop. An fp32 oracle forward at this size is 1.5 s on the GPU: ~45 s, once per process."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle.parity import full_depth_parity
    rec = full_depth_parity("cuda:0", steps=28, every=3, hw=64, model_config={"union_cond_attn": True, "attn_fp8": True})
    print("PARITY_FP8ATTN_1024 " + json.dumps(rec))
    assert rec["noise_pred_relerr_mean"] <= FP8_ATTN_NOISE_PRED_MEAN and rec["noise_pred_relerr_max"] <= FP8_ATTN_NOISE_PRED_MAX, rec
    assert rec["final_latent_relerr"] <= FP8_ATTN_FINAL_LATENT and rec["final_latent_cosine"] > 0.99999, rec


def test_fp8_gemm_mode_is_lossy_and_says_so():
    """gemm_fp8 (e4m3 operands in every block GEMM): ~1e-1 per forward at full depth. Asserted as a band so that neither a silent
    regression nor a silent 'improvement' of the documented figure goes unnoticed."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle.parity import full_depth_parity
    rec = full_depth_parity("cuda:0", steps=28, every=9, model_config={"union_cond_attn": True, "gemm_fp8": True})
    print("PARITY_FP8GEMM_512 " + json.dumps(rec))
    assert 5e-2 < rec["noise_pred_relerr_mean"] < 2e-1, rec
