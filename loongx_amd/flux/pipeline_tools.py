"""Mirror of the reference's src/flux/pipeline_tools.py (encode_images :7-30, prepare_text_input :33-52)."""
from __future__ import annotations

import torch


def encode_images(pipeline, images):
    """VAE-encode + (x - shift) * scale + 2x2 pack + ids.  Needs `pipeline.vae`; VAE-free callers hand packed
    latents to `Condition(latents=...)` instead (the VAE is outside the denoise hot path)."""
    if pipeline.vae is None or pipeline.image_processor is None:
        raise NotImplementedError("encode_images needs a VAE: construct LxFluxPipeline(vae=..., image_processor=...) or "
                                  "pass pre-encoded packed latents via Condition(latents=...)")
    images = pipeline.image_processor.preprocess(images)
    images = images.to(pipeline.device).to(pipeline.dtype)
    z = pipeline.vae.encode(images).latent_dist.sample()
    z = (z - pipeline.vae.config.shift_factor) * pipeline.vae.config.scaling_factor
    tokens = pipeline._pack_latents(z, *z.shape)
    ids = pipeline._prepare_latent_image_ids(z.shape[0], z.shape[2], z.shape[3], pipeline.device, torch.float32)
    if tokens.shape[1] != ids.shape[0]:   # pipelines whose id helper takes pre-halved sizes (reference :22-29)
        ids = pipeline._prepare_latent_image_ids(z.shape[0], z.shape[2] // 2, z.shape[3] // 2, pipeline.device, torch.float32)
    return tokens, ids


def prepare_text_input(pipeline, prompts, max_sequence_length: int = 512):
    return pipeline.encode_prompt(prompt=prompts, prompt_2=None, prompt_embeds=None, pooled_prompt_embeds=None,
                                  device=pipeline.device, num_images_per_prompt=1, max_sequence_length=max_sequence_length,
                                  lora_scale=None)
