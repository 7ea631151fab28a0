"""Mirror of the reference's src/flux/generate.py: `generate()` (:72-394), `prepare_params` (:25-65), `get_config`
(:16-22), `seed_everything` (:68-71) -- the LoongX denoise driver on MI355X.

Differences from the reference, all deliberate and recorded in DESIGN.md:
  Q1  brain signals reach the CS3 encoders as [B,C,L] (what OminiModel.step does, model.py:659-673), not flatten(1);
  Q2  with fuse_flag=False the reference replaces the text embeddings only when BOTH brain sides exist (generate.py:252-255);
      that literal rule is the default here (`brain_replace="both"`). `brain_replace="per_stream"` is the opt-in extension that
      lets each side replace its own embedding (EEG[+PPG] -> prompt_embeds, fNIRS[+Motion] -> pooled), which is what makes
      EEG-only conditioning (BASELINE configs[1]) take effect; bench.py / inference.py --synthetic ask for it explicitly;
  Q5  signals may carry a batch dimension ([B,C,L]); a [C,L] tensor is treated as batch 1 like the reference.
  Q6  fp16 operand mode (model_config["operands"] = "fp16" / dtype float16): where the reference clips fp16 activations silently
      (block.py:275-276, 336-337) the kernels saturate AND count; generate() reads the counter once per image and applies
      model_config["f16_overflow"]: "raise" (default; F16OverflowError), "warn", or "fallback" (the image is computed again from the
      same start latents with bf16 operands).
"""
from __future__ import annotations

import os
import warnings
from typing import Any, Callable, Dict, List, Optional, Union

import numpy as np
import torch
import yaml

from .condition import Condition
from .pipeline import FluxPipelineOutput, calculate_shift, retrieve_timesteps
from .transformer import tranformer_forward


F16_OVERFLOW_POLICIES = ("raise", "warn", "fallback")


class F16OverflowError(RuntimeError):
    """An operand of the fp16 operand mode left fp16's range (+-65504) and was saturated: the image is not what the mode promises."""


def get_config(config_path: str = None):
    config_path = config_path or os.environ.get("XFL_CONFIG")
    if not config_path:
        return {}
    with open(config_path, "r") as f:
        return yaml.safe_load(f)


def prepare_params(prompt: Union[str, List[str]] = None, prompt_2=None, height: Optional[int] = 512, width: Optional[int] = 512,
                   num_inference_steps: int = 28, timesteps: List[int] = None, guidance_scale: float = 3.5,
                   num_images_per_prompt: Optional[int] = 1, generator=None, latents: Optional[torch.Tensor] = None,
                   prompt_embeds: Optional[torch.Tensor] = None, pooled_prompt_embeds: Optional[torch.Tensor] = None,
                   output_type: Optional[str] = "pil", return_dict: bool = True,
                   joint_attention_kwargs: Optional[Dict[str, Any]] = None,
                   callback_on_step_end: Optional[Callable[[int, int, Dict], None]] = None,
                   callback_on_step_end_tensor_inputs: List[str] = ["latents"], max_sequence_length: int = 512, **kwargs):
    return (prompt, prompt_2, height, width, num_inference_steps, timesteps, guidance_scale, num_images_per_prompt, generator,
            latents, prompt_embeds, pooled_prompt_embeds, output_type, return_dict, joint_attention_kwargs, callback_on_step_end,
            callback_on_step_end_tensor_inputs, max_sequence_length)


def seed_everything(seed: int = 42):
    torch.manual_seed(seed)
    np.random.seed(seed)


def _signal(x, device, dtype, fixed_len, model):
    """[C,L] or [B,C,L] array-like -> padded/truncated [B,C,fixed_len] (generate.py:170-211)."""
    if x is None:
        return None
    if not isinstance(x, torch.Tensor):
        x = torch.tensor(np.asarray(x))
    if x.dim() == 2:
        x = x.unsqueeze(0)
    if x.dim() != 3:
        raise ValueError(f"brain signal must be [C,L] or [B,C,L], got {tuple(x.shape)}")
    return model.spatial_pyramid_pooling(x.to(device).to(dtype), fixed_len)


@torch.no_grad()
def generate(model, pipeline, conditions: List[Condition] = None, config_path: str = None,
             model_config: Optional[Dict[str, Any]] = {}, condition_scale: float = 1.0, default_lora: bool = False,
             additional_condition1: Optional[torch.Tensor] = None,   # EEG
             additional_condition2: Optional[torch.Tensor] = None,   # fNIRS
             additional_condition3: Optional[torch.Tensor] = None,   # PPG
             additional_condition4: Optional[torch.Tensor] = None,   # Motion
             use_brain_condition: bool = True, fuse_flag: bool = True, brain_replace: str = "both", **params):
    model_config = model_config or get_config(config_path).get("model", {})
    self = pipeline
    if brain_replace not in ("both", "per_stream"):
        raise ValueError(f"brain_replace must be 'both' (the reference rule) or 'per_stream', got {brain_replace!r}")
    # the step-invariant conditioning cache of the transformer lives for the steps of ONE image: every tensor made below is
    # freed on return and its address may come back with different content on the next call
    if hasattr(pipeline.transformer, "invalidate_conditioning"):
        pipeline.transformer.invalidate_conditioning()
    if condition_scale != 1:
        pipeline.transformer.c_factor = float(condition_scale)
    (prompt, prompt_2, height, width, num_inference_steps, timesteps, guidance_scale, num_images_per_prompt, generator, latents,
     prompt_embeds, pooled_prompt_embeds, output_type, return_dict, joint_attention_kwargs, callback_on_step_end,
     callback_on_step_end_tensor_inputs, max_sequence_length) = prepare_params(**params)
    height = height or self.default_sample_size * self.vae_scale_factor
    width = width or self.default_sample_size * self.vae_scale_factor
    self.check_inputs(prompt, prompt_2, height, width, prompt_embeds=prompt_embeds, pooled_prompt_embeds=pooled_prompt_embeds,
                      callback_on_step_end_tensor_inputs=callback_on_step_end_tensor_inputs, max_sequence_length=max_sequence_length)
    self._guidance_scale, self._joint_attention_kwargs, self._interrupt = guidance_scale, joint_attention_kwargs, False
    if prompt is not None and isinstance(prompt, str):
        batch_size = 1
    elif prompt is not None and isinstance(prompt, list):
        batch_size = len(prompt)
    else:
        batch_size = prompt_embeds.shape[0]
    device = self._execution_device
    lora_scale = self.joint_attention_kwargs.get("scale", None) if self.joint_attention_kwargs is not None else None
    prompt_embeds, pooled_prompt_embeds, text_ids = self.encode_prompt(
        prompt=prompt, prompt_2=prompt_2, prompt_embeds=prompt_embeds, pooled_prompt_embeds=pooled_prompt_embeds, device=device,
        num_images_per_prompt=num_images_per_prompt, max_sequence_length=max_sequence_length, lora_scale=lora_scale)

    # ---- CS3 encoders + DGF fusion: once per image, outside the loop (generate.py:168-258) ----------
    if use_brain_condition and any(c is not None for c in (additional_condition1, additional_condition2,
                                                           additional_condition3, additional_condition4)):
        f32 = torch.float32
        eeg = _signal(additional_condition1, device, f32, model.eeg_fixed_length, model)
        fnirs = _signal(additional_condition2, device, f32, model.fnirs_fixed_length, model)
        ppg = _signal(additional_condition3, device, f32, model.ppg_fixed_length, model)
        motion = _signal(additional_condition4, device, f32, model.motion_fixed_length, model)
        prompt_embeds_brain = pooled_prompt_embeds_brain = None
        if eeg is not None:
            eeg_features = model.eeg_projection(eeg)
            prompt_embeds_brain = model.fuse_eeg(eeg_features, model.ppg_projection(ppg)) if ppg is not None else eeg_features
        if fnirs is not None:
            fnirs_features = model.fnirs_projection(fnirs)
            pooled_prompt_embeds_brain = (model.fuse_fnirs(fnirs_features, model.motion_projection(motion))
                                          if motion is not None else fnirs_features)
        for name, t, ref in (("prompt", prompt_embeds_brain, prompt_embeds), ("pooled", pooled_prompt_embeds_brain, pooled_prompt_embeds)):
            if t is not None and t.shape[0] != ref.shape[0]:
                if t.shape[0] != 1:
                    raise ValueError(f"brain {name} embeddings have batch {t.shape[0]}, prompt batch is {ref.shape[0]}")
        if fuse_flag and prompt_embeds_brain is not None and pooled_prompt_embeds_brain is not None:
            prompt_embeds = model.duan_norm_prompt(prompt_embeds.float(), prompt_embeds_brain.expand_as(prompt_embeds).contiguous())
            pooled_prompt_embeds = model.duan_norm_pooled(
                pooled_prompt_embeds.float().unsqueeze(1),
                pooled_prompt_embeds_brain.expand_as(pooled_prompt_embeds).contiguous().unsqueeze(1)).squeeze(1)
        elif brain_replace == "per_stream":
            if prompt_embeds_brain is not None:
                prompt_embeds = prompt_embeds_brain.expand(prompt_embeds.shape[0], -1, -1)
            if pooled_prompt_embeds_brain is not None:
                pooled_prompt_embeds = pooled_prompt_embeds_brain.expand(pooled_prompt_embeds.shape[0], -1)
        elif prompt_embeds_brain is not None and pooled_prompt_embeds_brain is not None:
            prompt_embeds, pooled_prompt_embeds = prompt_embeds_brain, pooled_prompt_embeds_brain

    # ---- latents + condition tokens ----------------------------------------------------------------------
    num_channels_latents = self.transformer.config.in_channels // 4
    latents, latent_image_ids = self.prepare_latents(batch_size * num_images_per_prompt, num_channels_latents, height, width,
                                                     torch.float32, device, generator, latents)
    condition_latents, condition_ids, condition_type_ids = ([] for _ in range(3))
    use_condition = conditions is not None or []
    if use_condition:
        assert len(conditions) <= 1, "Only one condition is supported for now."
        if not default_lora:
            pipeline.set_adapters(conditions[0].condition_type)
        for condition in conditions:
            tokens, ids, type_id = condition.encode(self)
            condition_latents.append(tokens)
            condition_ids.append(ids)
            condition_type_ids.append(type_id)
        condition_latents = torch.cat(condition_latents, dim=1)
        condition_ids = torch.cat(condition_ids, dim=0)
        condition_type_ids = torch.cat(condition_type_ids, dim=0)
        if condition_latents.shape[0] == 1 and latents.shape[0] > 1:
            condition_latents = condition_latents.expand(latents.shape[0], -1, -1).contiguous()

    # ---- sigma schedule (generate.py:290-310) ---------------------------------------------------------------
    sigmas = np.linspace(1.0, 1 / num_inference_steps, num_inference_steps)
    cfg = self.scheduler.config
    mu = calculate_shift(latents.shape[1], cfg.base_image_seq_len, cfg.max_image_seq_len, cfg.base_shift, cfg.max_shift)
    timesteps, num_inference_steps = retrieve_timesteps(self.scheduler, num_inference_steps, device, timesteps, sigmas, mu=mu)
    num_warmup_steps = max(len(timesteps) - num_inference_steps * self.scheduler.order, 0)
    self._num_timesteps = len(timesteps)
    guidance = None
    if self.transformer.config.guidance_embeds:
        guidance = torch.full((latents.shape[0],), float(guidance_scale), device=device, dtype=torch.float32)
    prompt_embeds = prompt_embeds.contiguous()
    pooled_prompt_embeds = pooled_prompt_embeds.contiguous()

    # ---- denoise loop (generate.py:313-369) ------------------------------------------------------------------
    # the whole schedule is known here: hand it to the engine so the AdaLN modulation weights stream once per image
    # (host copy of the same fp32 values: a .tolist() on the device tensor would drain the stream here and keep the host from
    #  preparing this image while the GPU is still busy with the previous one)
    th = getattr(self.scheduler, "timesteps_host", None)
    if th is not None and latents.dtype == torch.float32 and len(th) == len(timesteps):
        sched_ts = (th / np.float32(1000)).tolist()
    else:
        sched_ts = (timesteps.to(latents.dtype) / 1000).tolist()
    engine = getattr(pipeline.transformer, "engine", None)
    policy = str((model_config or {}).get("f16_overflow", "raise"))
    if policy not in F16_OVERFLOW_POLICIES:
        raise ValueError(f'model_config["f16_overflow"] = {policy!r}: one of {F16_OVERFLOW_POLICIES}')
    start_latents, operands_used = latents, None

    def denoise(latents, prompt_embeds, model_config):
        with self.progress_bar(total=num_inference_steps) as progress_bar:
            for i, t in enumerate(timesteps):
                if self.interrupt:
                    continue
                timestep = t.expand(latents.shape[0]).to(latents.dtype)
                noise_pred = tranformer_forward(
                    self.transformer, model_config=model_config,
                    condition_latents=condition_latents if use_condition else None,
                    condition_ids=condition_ids if use_condition else None,
                    condition_type_ids=condition_type_ids if use_condition else None,
                    hidden_states=latents, timestep=timestep / 1000, guidance=guidance, pooled_projections=pooled_prompt_embeds,
                    encoder_hidden_states=prompt_embeds, txt_ids=text_ids, img_ids=latent_image_ids,
                    joint_attention_kwargs=self.joint_attention_kwargs, return_dict=False, lx_schedule=(i, sched_ts))[0]
                latents = self.scheduler.step(noise_pred, t, latents, return_dict=False)[0]
                if callback_on_step_end is not None:
                    scope = dict(locals())                                  # (a comprehension has its own locals() before 3.12)
                    callback_kwargs = {}
                    for k in callback_on_step_end_tensor_inputs:
                        callback_kwargs[k] = scope[k]
                    callback_outputs = callback_on_step_end(self, i, t, callback_kwargs)
                    latents = callback_outputs.pop("latents", latents)
                    prompt_embeds = callback_outputs.pop("prompt_embeds", prompt_embeds)
                if i == len(timesteps) - 1 or ((i + 1) > num_warmup_steps and (i + 1) % self.scheduler.order == 0):
                    progress_bar.update()
        return latents

    latents = denoise(start_latents, prompt_embeds, model_config)
    if engine is not None and engine.f16:
        # Q6: the saturation counter of this image's fp16 operand images. "fallback" has to know NOW, and a decode drains the stream anyway:
        # one 4-byte synchronous read. Otherwise (latent output: the host runs ahead of the GPU) the pinned-host asynchronous read of
        # check_status(sync=False) -- an event then surfaces at the next image's check; callers that keep latents end their loop with
        # engine.f16_overflow_poll(sync=True) (inference.py does, per image, before it saves).
        operands_used = "fp16"
        n_sat = engine.f16_overflow_poll(sync=(policy == "fallback" or output_type != "latent"))
        if n_sat:
            what = (f"fp16 operand mode: {n_sat} saturation event(s) (producer waves that clipped an operand to +-65504, or clipped weights: "
                    f"{getattr(engine, 'w16_clipped', 0)})")
            if policy == "raise":
                raise F16OverflowError(what + '; model_config["f16_overflow"] = "fallback" recomputes such images with bf16 operands, "warn" keeps them')
            if policy == "warn":
                warnings.warn(what + "; the image was kept", RuntimeWarning, stacklevel=2)
            else:
                warnings.warn(what + "; recomputing this image with bf16 operands", RuntimeWarning, stacklevel=2)
                pipeline.transformer.invalidate_conditioning()
                timesteps, num_inference_steps = retrieve_timesteps(self.scheduler, num_inference_steps, device, None, sigmas, mu=mu)      # (step index back to 0)
                latents = denoise(start_latents, prompt_embeds, dict(model_config or {}, operands="bf16"))
                operands_used = "bf16 (fp16 overflow fallback)"

    if output_type == "latent":
        image = latents
    else:
        if self.vae is None or self.image_processor is None:
            raise NotImplementedError("decoding to pixels needs a VAE (outside the MI355X hot path): use output_type='latent' "
                                      "or construct LxFluxPipeline(vae=..., image_processor=...)")
        if hasattr(pipeline.transformer, "engine"):
            # a split-K pair time-out invalidates this image's latents: find out BEFORE it is decoded and handed back (the decode
            # drains the stream anyway, so the synchronous check costs one 4-byte read)
            pipeline.transformer.engine.check_status(sync=True)
        z = self._unpack_latents(latents, height, width, self.vae_scale_factor)
        z = (z / self.vae.config.scaling_factor) + self.vae.config.shift_factor
        image = self.vae.decode(z, return_dict=False)[0]
        image = self.image_processor.postprocess(image, output_type=output_type)
    self.maybe_free_model_hooks()
    if condition_scale != 1:
        pipeline.transformer.c_factor = None
    if hasattr(pipeline.transformer, "invalidate_conditioning"):
        pipeline.transformer.invalidate_conditioning()
        if output_type == "latent":
            # latents stay on the device and the host runs ahead: no synchronisation here. A pair time-out surfaces at the next
            # image's check; callers that keep latents must call engine.check_status() once at the end of their loop
            # (inference.py does, per shard)
            pipeline.transformer.engine.check_status(sync=False)
    if not return_dict:
        return (image,)
    return FluxPipelineOutput(images=image, lx_operands=operands_used)
