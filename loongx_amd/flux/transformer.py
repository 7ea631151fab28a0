"""Mirror of the reference's src/flux/transformer.py: `tranformer_forward` (sic, transformer.py:47-252) and
`prepare_params` (:18-44), driving the MI355X DiT engine.

`transformer` is an `LxFluxTransformer` (the MI355X stand-in for diffusers' FluxTransformer2DModel: it owns the packed
device weights and exposes `.config`, `.transformer_blocks`, `.single_transformer_blocks`).  Everything that does not
depend on the timestep (prompt/condition embedders, RoPE tables, guidance/text embeddings, the condition stream's
modulations) is computed once per distinct set of conditioning tensors and reused across the 28 calls of a denoise loop.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Any, Dict, Optional

import torch

from .block import LxBlock
from .engine import DiTEngine
from .weights import FluxConfig, PackedWeights, pack_state_dict, synthetic_weights


class Transformer2DModelOutput(SimpleNamespace):
    pass


def prepare_params(hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor = None,
                   pooled_projections: torch.Tensor = None, timestep: torch.Tensor = None, img_ids: torch.Tensor = None,
                   txt_ids: torch.Tensor = None, guidance: torch.Tensor = None,
                   joint_attention_kwargs: Optional[Dict[str, Any]] = None, controlnet_block_samples=None,
                   controlnet_single_block_samples=None, return_dict: bool = True, **kwargs):
    return (hidden_states, encoder_hidden_states, pooled_projections, timestep, img_ids, txt_ids, guidance,
            joint_attention_kwargs, controlnet_block_samples, controlnet_single_block_samples, return_dict)


class LxFluxTransformer:
    """Holds the fused weights on one MI355X and the engine that runs them."""

    def __init__(self, weights: PackedWeights, device="cuda", precise: bool = False, operands: str = "bf16"):
        """precise=True: fp32-class arithmetic by default (the reference's shipped `dtype: float32`, train/config/seed_512.yaml:2):
        split-bf16 MFMA GEMMs + fp32 attention; a call's model_config["precise"] overrides the default either way.
        operands="fp16" (dtype=torch.float16): the GEMM operand images are IEEE fp16 instead of bf16 by default (same speed class, 1/8 of
        the rounding error: DiTEngine.operands_default; model_config["operands"] overrides per call)."""
        cfg = weights.cfg
        self.engine = DiTEngine(weights, device)
        self.engine.precise_default = bool(precise)
        self.engine.operands_default = operands
        self.config = SimpleNamespace(in_channels=cfg.in_channels, num_layers=cfg.num_layers,
                                      num_single_layers=cfg.num_single_layers, attention_head_dim=cfg.attention_head_dim,
                                      num_attention_heads=cfg.num_attention_heads, joint_attention_dim=cfg.joint_attention_dim,
                                      pooled_projection_dim=cfg.pooled_projection_dim, guidance_embeds=cfg.guidance_embeds,
                                      axes_dims_rope=cfg.axes_dims_rope)
        self.transformer_blocks = [LxBlock(self.engine, "double", i) for i in range(cfg.num_layers)]
        self.single_transformer_blocks = [LxBlock(self.engine, "single", j) for j in range(cfg.num_single_layers)]
        self.device = self.engine.device
        self.dtype = torch.bfloat16
        self.training = False
        self.gradient_checkpointing = False
        self.c_factor: Optional[float] = None      # generate(condition_scale != 1) sets this (generate.py:90-94)
        self._cond_key = None
        self._cond_refs = None                      # strong references to the tensors _cond_key was computed from

    def invalidate_conditioning(self) -> None:
        """Forget the step-invariant conditioning cache (generate() calls this at the start and end of every image)."""
        self._cond_key = self._cond_refs = None
        self.engine.cond_ready = False
        self.engine.sched = None

    @classmethod
    def from_state_dict(cls, sd, cfg: FluxConfig, device="cuda", lora_scale: float = 1.0, prefix: str = "", precise: bool = False,
                        operands: str = "bf16"):
        return cls(pack_state_dict(sd, cfg, device, lora_scale, prefix, precise=precise), device, precise=precise, operands=operands)

    @classmethod
    def synthetic(cls, cfg: Optional[FluxConfig] = None, device="cuda", seed: int = 0, precise: bool = False, operands: str = "bf16"):
        return cls(synthetic_weights(cfg or FluxConfig(), device, seed), device, precise=precise, operands=operands)

    def named_modules(self):
        for i, b in enumerate(self.transformer_blocks):
            yield f"transformer_blocks.{i}.attn", b.attn
        for i, b in enumerate(self.single_transformer_blocks):
            yield f"single_transformer_blocks.{i}.attn", b.attn

    def eval(self):
        return self

    def to(self, *a, **k):
        return self


def _tkey(t: Optional[torch.Tensor]):
    """Identity of a conditioning tensor for the step-invariant cache. Identity (address, shape, dtype, version) is only a valid
    stand-in for content while the tensor is ALIVE -- a freed tensor's address is handed to the next allocation of that size --
    so the cache also keeps strong references to every keyed tensor (LxFluxTransformer._cond_refs), and generate() drops the
    cache at the start and end of every image. What identity cannot see is an in-place write that bypasses torch's version
    counter (a raw kernel through the C ABI): callers that do that between calls must call invalidate_conditioning()."""
    return None if t is None else (t.data_ptr(), tuple(t.shape), t.dtype, t._version)


def tranformer_forward(transformer: LxFluxTransformer, condition_latents: torch.Tensor, condition_ids: torch.Tensor,
                       condition_type_ids: torch.Tensor, model_config: Optional[Dict[str, Any]] = {}, c_t=0, **params):
    """One velocity prediction. Same keyword surface as the reference (prepare_params); `condition_type_ids` is
    accepted and ignored exactly as in the reference (transformer.py:133 is commented out there)."""
    (hidden_states, encoder_hidden_states, pooled_projections, timestep, img_ids, txt_ids, guidance,
     joint_attention_kwargs, controlnet_block_samples, controlnet_single_block_samples, return_dict) = prepare_params(**params)
    if controlnet_block_samples is not None or controlnet_single_block_samples is not None:
        raise NotImplementedError("controlnet residuals are not part of the LoongX path (unused hooks at transformer.py:173-181,231-239)")
    eng = transformer.engine
    use_condition = condition_latents is not None
    if txt_ids.ndim == 3:
        txt_ids = txt_ids[0]
    if img_ids.ndim == 3:
        img_ids = img_ids[0]
    if eng.cfg.guidance_embeds and guidance is None:
        raise ValueError("guidance_embeds=True transformer called without guidance")
    if not eng.cfg.guidance_embeds:
        guidance = None
    mc = dict(model_config or {})
    key = (_tkey(encoder_hidden_states), _tkey(pooled_projections), _tkey(guidance), _tkey(txt_ids), _tkey(img_ids),
           _tkey(condition_latents), _tkey(condition_ids), float(c_t), tuple(sorted(mc.items())), transformer.c_factor,
           hidden_states.shape[0], getattr(eng.w, "weights_version", 0), getattr(eng.w, "lora_version", 0))      # (a weight broadcast or a new
    # adapter after the conditioning: the cached embedder outputs and modulations were computed from the old weights)
    if key != transformer._cond_key or not eng.cond_ready:
        eng.set_conditioning(encoder_hidden_states, pooled_projections, guidance, txt_ids, img_ids,
                             condition_latents if use_condition else None, condition_ids if use_condition else None,
                             c_t=float(c_t), model_config=mc, c_factor=transformer.c_factor)
        transformer._cond_key = key
        transformer._cond_refs = (encoder_hidden_states, pooled_projections, guidance, txt_ids, img_ids, condition_latents, condition_ids)
    # Extension (the reference swallows unknown kwargs): lx_schedule=(i, timesteps) tells the engine that `timestep` is entry
    # i of a known schedule (same units as `timestep`), so all steps' modulation vectors come from one weight pass.
    step_index = None
    sched = params.get("lx_schedule")
    if sched is not None:
        step_index, ts = int(sched[0]), tuple(float(v) for v in sched[1])
        if eng.sched is None or eng.sched[0] != ts:
            eng.prepare_schedule(torch.tensor(ts, dtype=torch.float32))
    out = eng.forward(hidden_states.to(device=eng.device, dtype=torch.float32), timestep.to(eng.device), step_index=step_index)
    out = out.to(hidden_states.dtype) if hidden_states.dtype != torch.float32 else out.clone()
    if not return_dict:
        return (out,)
    return Transformer2DModelOutput(sample=out)
