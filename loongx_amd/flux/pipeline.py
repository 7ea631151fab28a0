"""`LxFluxPipeline`: the slice of diffusers' FluxPipeline that the reference's generate() touches
(src/flux/generate.py:72-394), for MI355X.  Restated from diffusers==0.31.0 (train/requirements.txt:1): latent packing,
latent image ids, calculate_shift, FlowMatchEulerDiscreteScheduler.  The Euler update runs in HIP (lx_euler_step).

Text encoders (T5/CLIP) and the VAE are outside the denoise hot path ("next"): pass `prompt_embeds`,
`pooled_prompt_embeds` and `output_type="latent"`, or plug callables in via `text_encoder=` / `vae=`.
"""
from __future__ import annotations

import math
from contextlib import contextmanager
from types import SimpleNamespace
from typing import List, Optional

import numpy as np
import torch

from .. import ops


class FluxPipelineOutput(SimpleNamespace):
    pass


def calculate_shift(image_seq_len, base_seq_len: int = 256, max_seq_len: int = 4096, base_shift: float = 0.5,
                    max_shift: float = 1.16):
    m = (max_shift - base_shift) / (max_seq_len - base_seq_len)
    return image_seq_len * m + (base_shift - m * base_seq_len)


class FlowMatchEulerDiscreteScheduler:
    """FLUX.1-dev scheduler config; sigma schedule on the host in fp64 -> fp32 like diffusers, step on the GPU."""
    order = 1

    def __init__(self, num_train_timesteps=1000, shift=3.0, use_dynamic_shifting=True, base_shift=0.5, max_shift=1.15,
                 base_image_seq_len=256, max_image_seq_len=4096):
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, shift=shift, use_dynamic_shifting=use_dynamic_shifting,
                                      base_shift=base_shift, max_shift=max_shift, base_image_seq_len=base_image_seq_len,
                                      max_image_seq_len=max_image_seq_len)
        self.timesteps = self.sigmas = None
        self._step_index = None

    def set_timesteps(self, num_inference_steps=None, device=None, sigmas=None, mu=None):
        if sigmas is None:
            sigmas = np.linspace(self.config.num_train_timesteps, 1, num_inference_steps) / self.config.num_train_timesteps
        sigmas = np.asarray(sigmas, dtype=np.float64)
        if self.config.use_dynamic_shifting:
            if mu is None:
                raise ValueError("use_dynamic_shifting=True needs mu")
            sigmas = math.exp(mu) / (math.exp(mu) + (1.0 / sigmas - 1.0))
        else:
            s = self.config.shift
            sigmas = s * sigmas / (1 + (s - 1) * sigmas)
        sig = sigmas.astype(np.float32)
        self._sig_host = np.concatenate([sig, np.zeros(1, np.float32)])
        self.timesteps_host = sig * np.float32(self.config.num_train_timesteps)      # same fp32 values as self.timesteps, no device sync
        self.timesteps = torch.from_numpy(self.timesteps_host).to(device)
        self.sigmas = torch.from_numpy(self._sig_host).to(device)
        self._step_index = 0

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, return_dict: bool = False):
        """prev = sample(fp32) + (sigma_next - sigma) * model_output, cast to model_output.dtype (diffusers semantics)."""
        i = self._step_index
        ds = float(np.float32(self._sig_host[i + 1]) - np.float32(self._sig_host[i]))
        x = sample.to(torch.float32).clone() if (sample.dtype != torch.float32 or not sample.is_contiguous()) else sample.clone()
        v = model_output if model_output.dtype in (torch.float32, torch.bfloat16) else model_output.float()
        ops.euler_step(x, v.contiguous(), ds)
        self._step_index += 1
        return (x.to(model_output.dtype),)


def retrieve_timesteps(scheduler, num_inference_steps=None, device=None, timesteps=None, sigmas=None, **kwargs):
    if timesteps is not None:
        raise ValueError("custom `timesteps` are not supported by the flow-match scheduler")
    scheduler.set_timesteps(num_inference_steps, device=device, sigmas=sigmas, **kwargs)
    return scheduler.timesteps, len(scheduler.timesteps)


class LxFluxPipeline:
    vae_scale_factor = 16
    default_sample_size = 64

    def __init__(self, transformer, scheduler=None, vae=None, text_encoder=None, image_processor=None):
        """vae: an `LxAutoencoderKL` (loongx_amd/vae.py) or any object with its encode / decode / config surface;
        text_encoder: a callable (prompt, prompt_2, max_sequence_length) -> (prompt_embeds, pooled), e.g. `FluxTextEncoders`;
        image_processor defaults to the diffusers-0.31 `VaeImageProcessor(vae_scale_factor=16)` when a VAE is given."""
        self.transformer = transformer
        self.scheduler = scheduler or FlowMatchEulerDiscreteScheduler()
        if vae is not None and image_processor is None:
            from ..vae import VaeImageProcessor
            image_processor = VaeImageProcessor(vae_scale_factor=self.vae_scale_factor)
        self.vae, self.text_encoder, self.image_processor = vae, text_encoder, image_processor
        self.device = transformer.device
        self.dtype = torch.float32
        self._guidance_scale, self._joint_attention_kwargs, self._interrupt, self._num_timesteps = 3.5, None, False, 0

    # ---- properties generate() reads ------------------------------------------------------------
    @property
    def _execution_device(self):
        return self.device

    @property
    def interrupt(self):
        return self._interrupt

    @property
    def joint_attention_kwargs(self):
        return self._joint_attention_kwargs

    def to(self, *a, **k):
        return self

    @classmethod
    def from_pretrained(cls, path: str, device="cuda", dtype=torch.bfloat16, flux_config=None, lora_scale: float = 1.0,
                        load_vae: bool = True, load_text_encoders: bool = True):
        """A LOCAL diffusers-format FLUX.1 directory (the reference passes a hub id to FluxPipeline.from_pretrained,
        src/train/model.py:399-401; the box has no hub access): `transformer/` (safetensors, sharded or single) is packed for the
        MI355X engine; `vae/` becomes an `LxAutoencoderKL`; `text_encoder*/` + `tokenizer*/` a `FluxTextEncoders` (transformers on
        ROCm). Missing optional parts leave the corresponding slot empty. dtype float32 selects the engine's precise mode, float16 its
        fp16 operand mode (bfloat16: bf16 operands)."""
        import json
        import os
        from safetensors.torch import load_file
        from .transformer import LxFluxTransformer
        from .weights import FluxConfig

        def load_dir(d):
            idx = [f for f in os.listdir(d) if f.endswith(".safetensors.index.json")]
            if idx:
                files = sorted(set(json.load(open(os.path.join(d, idx[0])))["weight_map"].values()))
            else:
                files = sorted(f for f in os.listdir(d) if f.endswith(".safetensors"))
            if not files:
                raise FileNotFoundError(f"no .safetensors weights in {d}")
            sd = {}
            for f in files:
                sd.update(load_file(os.path.join(d, f)))
            return sd

        tdir = os.path.join(path, "transformer")
        if not os.path.isdir(tdir):
            raise FileNotFoundError(f"{path}: no transformer/ directory (expected a diffusers-format FLUX.1 checkpoint)")
        cfg = flux_config
        cj = os.path.join(tdir, "config.json")
        if cfg is None and os.path.isfile(cj):
            c = json.load(open(cj))
            cfg = FluxConfig(num_layers=c.get("num_layers", 19), num_single_layers=c.get("num_single_layers", 38),
                             num_attention_heads=c.get("num_attention_heads", 24), attention_head_dim=c.get("attention_head_dim", 128),
                             in_channels=c.get("in_channels", 64), joint_attention_dim=c.get("joint_attention_dim", 4096),
                             pooled_projection_dim=c.get("pooled_projection_dim", 768), guidance_embeds=c.get("guidance_embeds", True),
                             axes_dims_rope=tuple(c.get("axes_dims_rope", (16, 56, 56))))
        tsd = load_dir(tdir)
        tr = LxFluxTransformer.from_state_dict(tsd, cfg or FluxConfig.from_state_dict(tsd), device, lora_scale, precise=dtype == torch.float32,
                                              operands="fp16" if dtype == torch.float16 else "bf16")
        vae = text = None
        vdir = os.path.join(path, "vae")
        if load_vae and os.path.isdir(vdir):
            from ..vae import LxAutoencoderKL
            vc = json.load(open(os.path.join(vdir, "config.json"))) if os.path.isfile(os.path.join(vdir, "config.json")) else {}
            keep = ("in_channels", "out_channels", "latent_channels", "block_out_channels", "layers_per_block", "norm_num_groups",
                    "scaling_factor", "shift_factor")
            vae = LxAutoencoderKL(load_dir(vdir), {k: vc[k] for k in keep if k in vc}, device)
        if load_text_encoders and all(os.path.isdir(os.path.join(path, d)) for d in ("text_encoder", "tokenizer", "text_encoder_2", "tokenizer_2")):
            from .text import FluxTextEncoders
            text = FluxTextEncoders.from_pretrained(path, device, dtype)
        return cls(tr, vae=vae, text_encoder=text)

    def load_lora_weights(self, path: str, weight_name: str = "pytorch_lora_weights.safetensors", lora_scale: float = 1.0, **kwargs):
        """diffusers `FluxPipeline.load_lora_weights` for the transformer (model.py:472): `path` is the directory the
        reference saves with `save_lora` (model.py:526-531) or the .safetensors file itself. Replaces the adapters of the
        packed model in place and drops every cached graph / conditioning that baked the old ones in."""
        import os
        from .weights import install_lora
        f = os.path.join(path, weight_name) if os.path.isdir(path) else path
        if not os.path.isfile(f):
            raise FileNotFoundError(f"no LoRA weights at {f}")
        if f.endswith(".safetensors"):
            from safetensors.torch import load_file
            sd = load_file(f)
        else:
            sd = torch.load(f, map_location="cpu")
        eng = self.transformer.engine
        n = install_lora(eng.w, sd, lora_scale=lora_scale)
        eng.graphs = {}
        eng.shape = None               # the adapter rank sizes the modulation scratch: re-run setup()
        self.transformer.invalidate_conditioning()
        return n

    def set_adapters(self, *a, **k):
        """Adapters are baked into the packed weights (one LoRA per checkpoint, inference.py:114 default_lora=True)."""

    def maybe_free_model_hooks(self):
        pass

    @contextmanager
    def progress_bar(self, total=None):
        yield SimpleNamespace(update=lambda *a, **k: None)

    # ---- input handling ---------------------------------------------------------------------------
    def check_inputs(self, prompt, prompt_2, height, width, prompt_embeds=None, pooled_prompt_embeds=None,
                     callback_on_step_end_tensor_inputs=None, max_sequence_length=None):
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if prompt is not None and prompt_embeds is not None:
            raise ValueError("Cannot forward both `prompt` and `prompt_embeds`.")
        if prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`.")
        if prompt is not None and not isinstance(prompt, (str, list)):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        if prompt_embeds is not None and pooled_prompt_embeds is None:
            raise ValueError("If `prompt_embeds` are provided, `pooled_prompt_embeds` also have to be passed.")
        if max_sequence_length is not None and max_sequence_length > 512:
            raise ValueError(f"`max_sequence_length` cannot be greater than 512 but is {max_sequence_length}")

    def encode_prompt(self, prompt=None, prompt_2=None, prompt_embeds=None, pooled_prompt_embeds=None, device=None,
                      num_images_per_prompt: int = 1, max_sequence_length: int = 512, lora_scale=None):
        if prompt_embeds is None:
            if self.text_encoder is None:
                raise NotImplementedError("T5/CLIP text encoding is outside the MI355X hot path: pass prompt_embeds / "
                                          "pooled_prompt_embeds, or construct LxFluxPipeline(text_encoder=callable)")
            prompt_embeds, pooled_prompt_embeds = self.text_encoder(prompt, prompt_2, max_sequence_length)
        if num_images_per_prompt != 1:
            prompt_embeds = prompt_embeds.repeat_interleave(num_images_per_prompt, dim=0)
            pooled_prompt_embeds = pooled_prompt_embeds.repeat_interleave(num_images_per_prompt, dim=0)
        text_ids = torch.zeros(prompt_embeds.shape[1], 3, device=device or self.device, dtype=torch.float32)
        return prompt_embeds.to(device or self.device), pooled_prompt_embeds.to(device or self.device), text_ids

    # ---- latents -----------------------------------------------------------------------------------
    @staticmethod
    def _pack_latents(latents, batch_size, num_channels_latents, height, width):
        x = latents.view(batch_size, num_channels_latents, height // 2, 2, width // 2, 2).permute(0, 2, 4, 1, 3, 5)
        return x.reshape(batch_size, (height // 2) * (width // 2), num_channels_latents * 4)

    @staticmethod
    def _unpack_latents(latents, height, width, vae_scale_factor):
        b, n, ch = latents.shape
        h, w = height // vae_scale_factor, width // vae_scale_factor
        x = latents.view(b, h, w, ch // 4, 2, 2).permute(0, 3, 1, 4, 2, 5)
        return x.reshape(b, ch // 4, h * 2, w * 2)

    @staticmethod
    def _prepare_latent_image_ids(batch_size, height, width, device, dtype):
        """diffusers 0.31: `height`/`width` are the UNPACKED latent sizes; ids live on the (h/2, w/2) grid."""
        h2, w2 = height // 2, width // 2
        ids = torch.zeros(h2, w2, 3)
        ids[..., 1] += torch.arange(h2)[:, None]
        ids[..., 2] += torch.arange(w2)[None, :]
        return ids.reshape(h2 * w2, 3).to(device=device, dtype=dtype)

    def prepare_latents(self, batch_size, num_channels_latents, height, width, dtype, device, generator, latents=None):
        height = 2 * (int(height) // self.vae_scale_factor)
        width = 2 * (int(width) // self.vae_scale_factor)
        ids = self._prepare_latent_image_ids(batch_size, height, width, device, torch.float32)
        if latents is not None:
            return latents.to(device=device, dtype=dtype), ids
        shape = (batch_size, num_channels_latents, height, width)
        gdev = generator.device if isinstance(generator, torch.Generator) else device
        noise = torch.randn(shape, generator=generator if isinstance(generator, torch.Generator) else None, device=gdev, dtype=dtype)
        return self._pack_latents(noise.to(device), batch_size, num_channels_latents, height, width), ids
