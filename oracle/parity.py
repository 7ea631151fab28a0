"""Full-depth parity harness: the MI355X engine against the fp32 oracle, same weights, same inputs, same GPU.

TEST INFRASTRUCTURE (imported by tests/ and by bench.py's `parity_check` leg only, as the checker -- never as the thing
measured or shipped).  The oracle is `oracle/flux_ref.py` (the restatement of the reference's src/flux/transformer.py:47-252
and src/flux/block.py, pinned to reference-generated goldens) evaluated in fp32 by torch-ROCm on the SAME GPU, which is what
makes 57 full-width blocks x 28 steps affordable (37.7 TFLOP per forward: ~0.5 s on the fp32 matrix path, hours on host cores).

What is compared (last row):
  * teacher-forced: at every step i of the oracle's own denoise trajectory, the engine's `noise_pred` for the ORACLE's
    latents x_i against the oracle's (isolates the error of one forward at full depth);
  * free-running: the product's `generate()` from the same start -> final latents rel-err and cosine (error as it compounds
    over 28 Euler steps).
Weights: synthetic N(0, 0.02^2) as BASELINE.md section 4; the base weights are rounded to bf16-representable values first
(`bf16_exact_base=True`), which is what FLUX.1-dev checkpoints hold (the reference upcasts a bf16 checkpoint to its
configured dtype, train/config/seed_512.yaml:2), so both sides compute with the SAME weights and the number measures
arithmetic, not weight quantisation. LoRA matrices stay fp32 on the oracle side.
"""
from __future__ import annotations

import time
from typing import Dict, Optional

import numpy as np
import torch

from . import flux_modules as fm
from . import flux_ref as fr


def relerr(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def cosine(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-30))


@torch.no_grad()
def realistic_stats_(tr, seed: int = 0) -> Dict:
    """Statistics a trained checkpoint has and N(0, 0.02^2) weights with unit norms and zero biases do not (round 5; the reference loads
    FLUX.1-dev + a LoongX checkpoint, inference.py:24-60, src/train/model.py:399-401 -- neither is available offline):
      (a) per-layer q / k RMSNorm weights with gains 0.5 ... 2.5 and 15 % channel jitter: 16.33 max|w_q| max|w_k| runs from ~6 to ~200,
          so about half of the layers exceed the bounded-score attention's limit of 100 and keep the max-tracking kernel -- a MIXED
          plan inside one denoise step -- and the softmax is peaked where the gain is large;
      (b) outlier channels: four channels of the residual stream carry a constant 100-1000x the typical activation (biases of
          x_embedder / context_embedder: FLUX's "massive activations"), and three channels of every MLP hidden layer sit at 50-200
          (biases of ff.net[0] / proj_mlp) -- the dynamic range inside a row that a 16-bit operand image has to carry;
      (c) every other bias non-zero, N(0, 0.05^2).
    Values are bf16-representable where the base weights are (biases and norm weights are fp32 on both sides anyway).
    Returns what it did (for the bench line)."""
    g = torch.Generator().manual_seed(9000 + seed)
    blocks = list(tr.transformer_blocks) + list(tr.single_transformer_blocks)
    gains = []
    for li, b in enumerate(blocks):
        gain = 0.5 + 2.0 * ((li * 7) % 10) / 9.0
        gains.append(gain)
        for nm in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
            m = getattr(b.attn, nm, None)
            if m is not None:
                m.weight.copy_((gain * (1.0 + 0.15 * torch.randn(m.weight.shape[0], generator=g))).to(m.weight.device))
    for name, p in tr.named_parameters():
        if name.endswith("bias") and p.dim() == 1:
            p.copy_((0.05 * torch.randn(p.shape[0], generator=g)).to(p.device))
    D = tr.inner_dim
    res_ch = [17, 1031, 2049, 3001]
    res_val = [120.0, -300.0, 60.0, 900.0]
    for emb in (tr.x_embedder, tr.context_embedder):
        lin = getattr(emb, "base_layer", emb)
        for c, v in zip(res_ch, res_val):
            if c < D:
                lin.bias[c] = v
    hid_val = [50.0, 200.0, -90.0]
    for li, b in enumerate(blocks):
        ffs = [b.ff.net[0].proj, b.ff_context.net[0].proj] if hasattr(b, "ff") else [getattr(b.proj_mlp, "base_layer", b.proj_mlp)]
        for lin in ffs:
            n = lin.bias.shape[0]
            for k, v in enumerate(hid_val):
                lin.bias[(li * 131 + k * 4099 + 7) % n] = v
    return {"norm_gain_range": [min(gains), max(gains)], "residual_outlier_channels": dict(zip(res_ch, res_val)), "mlp_hidden_outliers": hid_val,
            "bias_std": 0.05}


def build_pair(device, num_layers: int = 19, num_single_layers: int = 38, heads: int = 24, seed: int = 0, std: float = 0.02,
               bf16_exact_base: bool = True, joint_dim: int = 4096, pooled_dim: int = 768, precise: bool = False, realistic: bool = False,
               engine: bool = True):
    """-> (oracle FluxTransformer2DModel on `device`, LxFluxTransformer packed from ITS state dict).
    realistic: realistic_stats_() on top of the synthetic weights (mixed bounded / max-tracking attention plan, outlier channels, biases)."""
    from loongx_amd.flux.transformer import LxFluxTransformer
    from loongx_amd.flux.weights import FluxConfig
    with torch.device(device):
        tr = fm.FluxTransformer2DModel(num_layers=num_layers, num_single_layers=num_single_layers, heads=heads, head_dim=128,
                                       in_channels=64, joint_dim=joint_dim, pooled_dim=pooled_dim, guidance_embeds=True, lora=True)
    g = torch.Generator(device=device).manual_seed(seed)
    with torch.no_grad():
        for name, p in tr.named_parameters():
            if p.dim() >= 2:
                p.normal_(0.0, std, generator=g)
                if bf16_exact_base and ".lora_" not in name:
                    p.copy_(p.to(torch.bfloat16).float())
            elif name.endswith("bias"):
                p.zero_()
            else:
                p.fill_(1.0)
    if realistic:
        realistic_stats_(tr, seed)
    tr.eval()
    if not engine:
        return tr, None
    return tr, engine_for(tr, device, num_layers, num_single_layers, heads=heads, joint_dim=joint_dim, pooled_dim=pooled_dim, precise=precise)


def engine_for(tr, device, num_layers: int = 19, num_single_layers: int = 38, heads: int = 24, joint_dim: int = 4096, pooled_dim: int = 768,
               precise: bool = False):
    """The product's transformer packed from the oracle model's state dict (identical weights on both sides)."""
    from loongx_amd.flux.transformer import LxFluxTransformer
    from loongx_amd.flux.weights import FluxConfig
    cfg = FluxConfig(num_layers=num_layers, num_single_layers=num_single_layers, num_attention_heads=heads, in_channels=64,
                     joint_attention_dim=joint_dim, pooled_projection_dim=pooled_dim, guidance_embeds=True)
    kw = dict(precise=True) if precise else {}
    return LxFluxTransformer.from_state_dict(tr.state_dict(), cfg, device, **kw)


# The oracle side of a parity run depends on the weights recipe, the shapes and the oracle-relevant model_config -- not on the engine's
# arithmetic mode. The 13 full-depth parity tests (and bench.py's seven parity legs) share five such combinations: the oracle model and its
# 28-step trajectory are built once per combination and per process (the fp32 model stays resident: 48 GB of the 288), only the engine side
# is rebuilt per mode. clear_cache() drops everything.
_ORACLE_MODELS: Dict = {}
_TRAJECTORIES: Dict = {}
_ENGINE_ONLY_KEYS = ("operands", "attn_fp8", "attn_fp8_exp2", "gemm_fp8", "precise", "f16_overflow")


def clear_cache() -> None:
    _ORACLE_MODELS.clear()
    _TRAJECTORIES.clear()
    torch.cuda.empty_cache()


def _oracle_model(device, num_layers, num_single_layers, seed, realistic):
    key = (str(device), num_layers, num_single_layers, seed, bool(realistic))
    if key not in _ORACLE_MODELS:
        tr, _ = build_pair(device, num_layers, num_single_layers, seed=seed, realistic=realistic, engine=False)
        _ORACLE_MODELS[key] = tr
    return _ORACLE_MODELS[key]


@torch.no_grad()
def _oracle_trajectory(dev, tr, steps, hw, n_txt, seed, brain, omc):
    """Inputs + the oracle's own denoise trajectory: latents and noise prediction of every step, and the final latents."""
    N = hw * hw
    g = torch.Generator(device=dev).manual_seed(4321 + seed)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)
    lat0, cond, pe, pooled = r(1, N, 64), r(1, N, 64), r(1, n_txt, 4096) * 0.1, r(1, 768)
    ent = {"lat0": lat0, "cond": cond, "pe_text": pe, "pooled_text": pooled, "signals": {}, "ref_cs3_sd": None}
    if brain is not None:
        from . import cs3 as ocs3
        if brain not in ("eeg", "all"):
            raise ValueError("brain must be None, 'eeg' or 'all'")
        torch.manual_seed(seed)
        ref_cs3 = ocs3.CS3DGF(seed=seed).eval()
        signals = {"eeg": r(1, 4, 4096)}
        if brain == "all":
            signals.update(fnirs=r(1, 6, 512), ppg=r(1, 4, 256), motion=r(1, 6, 128))
        cpu = {k: v.float().cpu() for k, v in signals.items()}
        rpe, rpool = ref_cs3.brain_embeds(pe.float().cpu(), pooled.float().cpu(), cpu.get("eeg"), cpu.get("fnirs"), cpu.get("ppg"), cpu.get("motion"),
                                          fuse_flag=brain == "all", per_stream=True)
        pe, pooled = rpe.to(dev), rpool.to(dev)                 # what the ORACLE's DiT is conditioned on
        ent.update(signals=signals, ref_cs3_sd=ref_cs3.state_dict())
    ids = fm.prepare_latent_image_ids(hw, hw).to(dev)
    cids = ids.clone()
    cids[:, 2] -= hw
    txt_ids = torch.zeros(n_txt, 3, device=dev)
    guidance = torch.full((1,), 3.5, device=dev)
    sch = fm.FlowMatchEulerDiscreteScheduler()
    sig = np.linspace(1.0, 1 / steps, steps)
    mu = fm.calculate_shift(N, sch.config.base_image_seq_len, sch.config.max_image_seq_len, sch.config.base_shift, sch.config.max_shift)
    timesteps, _ = fm.retrieve_timesteps(sch, steps, dev, None, sig, mu=mu)
    lat = lat0.clone()
    lats, wants, tss = [], [], []
    torch.cuda.synchronize(dev); t1 = time.time()
    for i, t in enumerate(timesteps):
        ts = t.expand(1).to(lat.dtype) / 1000
        want = fr.tranformer_forward(tr, cond, cids, None, omc, hidden_states=lat, encoder_hidden_states=pe, pooled_projections=pooled, timestep=ts,
                                     img_ids=ids, txt_ids=txt_ids, guidance=guidance)[0]
        lats.append(lat.clone()); wants.append(want.clone()); tss.append(ts)
        lat = sch.step(want, t, lat)[0]
    torch.cuda.synchronize(dev)
    ent.update(pe=pe, pooled=pooled, ids=ids, cids=cids, txt_ids=txt_ids, guidance=guidance, lats=lats, wants=wants, tss=tss, final=lat,
               oracle_s_per_forward=(time.time() - t1) / len(timesteps))
    return ent


@torch.no_grad()
def full_depth_parity(device="cuda:0", steps: int = 28, num_layers: int = 19, num_single_layers: int = 38, hw: int = 32,
                      n_txt: int = 512, seed: int = 0, precise: bool = False, model_config: Optional[Dict] = None,
                      every: int = 1, brain: Optional[str] = None, realistic: bool = False) -> Dict:
    """Runs both sides at batch 1 on identical synthetic inputs (BASELINE.md section 4 shapes) and returns the parity record
    that bench.py prints as `parity`. `every`: compare the teacher-forced noise_pred at every `every`-th step (the oracle still
    runs all steps).
    `brain`: None = the DiT alone on given text embeddings; "eeg" = BASELINE configs[1] as bench.py times it (EEG-only CS3
    conditioning, per-stream replacement rule: the EEG encoder's output IS the prompt-embedding stream); "all" = all four modalities
    through the CS3 encoders + DGF fusion (fuse_flag=True: configs[2]'s composition). The oracle side is oracle/cs3.py (full-size
    encoders, on the host) feeding oracle/flux_ref.py; the product side is generate() with the signals, i.e. its own CS3 / DGF
    kernels feeding its own DiT: the free-running figures then cover the COMPOSITION, and `brain_embeds_relerr` the encoders alone."""
    from loongx_amd.flux.condition import Condition
    from loongx_amd.flux.generate import generate
    from loongx_amd.flux.pipeline import LxFluxPipeline
    from loongx_amd.flux.transformer import tranformer_forward
    dev = torch.device(device)
    t0 = time.time()
    mc = dict(model_config or {"union_cond_attn": True})
    omc = {k: v for k, v in mc.items() if k not in _ENGINE_ONLY_KEYS}
    tr = _oracle_model(dev, num_layers, num_single_layers, seed, realistic)
    tk = (str(dev), num_layers, num_single_layers, seed, bool(realistic), hw, n_txt, steps, brain, tuple(sorted(omc.items())))
    cached = tk in _TRAJECTORIES
    if not cached:
        _TRAJECTORIES[tk] = _oracle_trajectory(dev, tr, steps, hw, n_txt, seed, brain, omc)
    T = _TRAJECTORIES[tk]
    lx = engine_for(tr, dev, num_layers, num_single_layers, precise=precise)
    N = hw * hw
    lat0, cond, pe, pooled, pe_text, pooled_text, signals = T["lat0"], T["cond"], T["pe"], T["pooled"], T["pe_text"], T["pooled_text"], T["signals"]
    ids, cids, txt_ids, guidance = T["ids"], T["cids"], T["txt_ids"], T["guidance"]
    model, brain_rec = None, {}
    if brain is not None:
        from loongx_amd.flux.pipeline import LxFluxPipeline as _Pipe
        from loongx_amd.train.model import OminiModel
        model = OminiModel.from_pipe(_Pipe(lx), T["ref_cs3_sd"], dict(model_config or {"union_cond_attn": True}), dev)

    # ---- teacher-forced engine predictions on the oracle's trajectory ------------------------------------------------
    per_step = []
    for i in range(steps):
        if i % every == 0 or i == steps - 1:
            kw = dict(hidden_states=T["lats"][i], encoder_hidden_states=pe, pooled_projections=pooled, timestep=T["tss"][i], img_ids=ids, txt_ids=txt_ids,
                      guidance=guidance)
            got = tranformer_forward(lx, cond, cids, None, mc, return_dict=False, **kw)[0]
            per_step.append((i, relerr(got, T["wants"][i])))
    final_oracle = T["final"]
    t_oracle_per_fwd = T["oracle_s_per_forward"]

    # ---- free-running product loop ---------------------------------------------------------------------------------
    lx.invalidate_conditioning()
    c = Condition("subject", latents=cond, latent_hw=(hw, hw), position_delta=[0, -hw])
    if brain is None:
        pipe = LxFluxPipeline(lx)
        final = generate(None, pipe, conditions=[c], height=16 * hw, width=16 * hw, num_inference_steps=steps, latents=lat0.clone(),
                         prompt_embeds=pe, pooled_prompt_embeds=pooled, output_type="latent", model_config=mc, default_lora=True,
                         use_brain_condition=False, guidance_scale=3.5).images
    else:
        # the product's own CS3 / DGF kernels make the embeddings from the raw signals (text embeddings in, as bench.py passes them)
        pipe = model.flux_pipe
        seen = {}

        def grab(_pipe, i, t, kw):
            if i == 0:
                seen["pe"] = kw["prompt_embeds"].float().clone()
            return {}
        final = generate(model, pipe, conditions=[c], height=16 * hw, width=16 * hw, num_inference_steps=steps, latents=lat0.clone(),
                         prompt_embeds=pe_text, pooled_prompt_embeds=pooled_text, output_type="latent", model_config=mc, default_lora=True,
                         additional_condition1=signals.get("eeg"), additional_condition2=signals.get("fnirs"), additional_condition3=signals.get("ppg"),
                         additional_condition4=signals.get("motion"), use_brain_condition=True, fuse_flag=brain == "all",
                         brain_replace="per_stream", guidance_scale=3.5, callback_on_step_end=grab,
                         callback_on_step_end_tensor_inputs=["prompt_embeds"]).images
        brain_rec = {"brain": "EEG-only CS3 conditioning (per-stream rule)" if brain == "eeg" else "EEG+fNIRS+PPG+motion CS3 + DGF fusion",
                     "brain_embeds_relerr": round(relerr(seen["pe"], pe), 8) if "pe" in seen else None}
    errs = [e for _, e in per_step]
    mode = "precise (split-bf16 MFMA, fp32 attention)" if precise else "bf16 MFMA operands, fp32 accumulate / residual"
    if mc.get("gemm_fp8") or mc.get("attn_fp8"):
        mode = "fp8 e4m3 operands (" + " + ".join(k for k in ("gemm_fp8", "attn_fp8") if mc.get(k)) + "), fp32 accumulate / residual"
    f16 = str(mc.get("operands", "bf16")).lower() in ("fp16", "f16", "float16") and not precise and not mc.get("gemm_fp8")
    if f16:
        mode = ("fp16 MFMA GEMM operands (bf16 attention operands), fp32 accumulate / residual" if not mc.get("attn_fp8")
                else "fp16 MFMA GEMM operands + e4m3 attention operands, fp32 accumulate / residual")
        brain_rec = dict(brain_rec, f16_saturated_waves=lx.engine.f16_overflow_count(), f16_weights_inexact_share=lx.engine.w16_inexact_share)
    rec = {"mode": mode,
           "oracle": "oracle/flux_ref.py fp32 on the same GPU (torch-ROCm), identical weights and inputs",
           "blocks": [num_layers, num_single_layers], "steps": steps, "tokens": [n_txt, N, N],
           "noise_pred_relerr_first": round(errs[0], 6), "noise_pred_relerr_max": round(max(errs), 6),
           "noise_pred_relerr_mean": round(float(np.mean(errs)), 6), "noise_pred_relerr_last": round(errs[-1], 6),
           "steps_compared": len(errs),
           "final_latent_relerr": round(relerr(final, final_oracle), 6), "final_latent_cosine": round(cosine(final, final_oracle), 8),
           "oracle_s_per_forward": round(t_oracle_per_fwd, 3), "oracle_trajectory": "cached" if cached else "computed",
           "wall_s": round(time.time() - t0, 1)}
    rec.update(brain_rec)
    if realistic:
        eng = lx.engine
        tab = getattr(eng.w, "q_log2", None) or {}
        bounds = [e["bound"] for e in tab.values()]
        rec["weights"] = "realistic_stats_: mixed q/k norm gains, outlier channels, non-zero biases (oracle/parity.py)"
        rec["bounded_score_layers"] = {"bounded": sum(1 for b_ in bounds if b_ <= getattr(eng, "_nomax_room", 100.0)), "layers": len(bounds),
                                       "largest_score_bound_log2": round(max(bounds), 1) if bounds else None}
    del lx, pipe, model
    torch.cuda.empty_cache()
    return rec
