"""Oracle: diffusers==0.31.0 arithmetic used by the LoongX denoise path, restated.

TEST INFRASTRUCTURE.  diffusers is pinned at 0.31.0 by the reference
(`train/requirements.txt:1`) but is NOT vendored under /root/reference and is not
installed in the build image, so every class below is a restatement of the
published diffusers algorithm ("parity unpinned" at this boundary).  Each class
cites the reference call site that consumes it.

The classes are duck-typed to the attribute surface the reference functions touch,
so `oracle/make_goldens.py` can plug them into the REAL `src/flux/block.py` /
`transformer.py` functions and pin `oracle/flux_ref.py` against those outputs.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------
# LoRA (peft) -- reference: src/flux/lora_controller.py:1-42 drives `scaling`/`scale_layer`
# ----------------------------------------------------------------------------------------
class BaseTunerLayer:
    """Marker base, stands in for peft.tuners.tuners_utils.BaseTunerLayer."""


class LoraLinear(nn.Module, BaseTunerLayer):
    """peft.tuners.lora.Linear restated: y = base(x) + scaling * B(A(x)).

    `enable_lora` (lora_controller.py:21-28) calls `scale_layer(0)` which multiplies
    every active adapter's scaling by the factor, and restores `scaling[...]` on exit.
    """

    def __init__(self, in_features: int, out_features: int, r: int = 4, lora_alpha: float = 4.0,
                 bias: bool = True, adapter: str = "default"):
        super().__init__()
        self.base_layer = nn.Linear(in_features, out_features, bias=bias)
        self.lora_A = nn.ModuleDict({adapter: nn.Linear(in_features, r, bias=False)})
        self.lora_B = nn.ModuleDict({adapter: nn.Linear(r, out_features, bias=False)})
        self.scaling = {adapter: lora_alpha / r}
        self.active_adapters = [adapter]
        self.in_features, self.out_features, self.r = in_features, out_features, r

    @property
    def weight(self):
        return self.base_layer.weight

    @property
    def bias(self):
        return self.base_layer.bias

    def scale_layer(self, scale: float) -> None:
        if scale == 1:
            return
        for a in self.active_adapters:
            self.scaling[a] *= scale

    def forward(self, x):
        y = self.base_layer(x)
        for a in self.active_adapters:
            s = self.scaling[a]
            y = y + self.lora_B[a](self.lora_A[a](x)) * s
        return y


def make_linear(i: int, o: int, lora: bool, r: int = 4, alpha: float = 4.0) -> nn.Module:
    return LoraLinear(i, o, r=r, lora_alpha=alpha) if lora else nn.Linear(i, o)


# ----------------------------------------------------------------------------------------
# normalisation
# ----------------------------------------------------------------------------------------
class RMSNorm(nn.Module):
    """diffusers.models.normalization.RMSNorm(dim, eps, elementwise_affine=True).
    Consumed at block.py:38-41, 60-67, 92-95 (attn.norm_q/k/added_q/added_k)."""

    def __init__(self, dim: int, eps: float = 1e-6):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        in_dtype = x.dtype
        var = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
        x = x * torch.rsqrt(var + self.eps)
        if self.weight.dtype in (torch.float16, torch.bfloat16):
            x = x.to(self.weight.dtype)
        x = x * self.weight
        if self.weight.dtype not in (torch.float16, torch.bfloat16):
            x = x.to(in_dtype)
        return x


class AdaLayerNormZero(nn.Module):
    """block.py:192-207: norm1 / norm1_context; returns 5-tuple."""

    def __init__(self, dim: int, lora: bool = False):
        super().__init__()
        self.silu = nn.SiLU()
        self.linear = make_linear(dim, 6 * dim, lora)
        self.norm = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)

    def forward(self, x, emb=None):
        emb = self.linear(self.silu(emb))
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = emb.chunk(6, dim=1)
        x = self.norm(x) * (1 + scale_msa[:, None]) + shift_msa[:, None]
        return x, gate_msa, shift_mlp, scale_mlp, gate_mlp


class AdaLayerNormZeroSingle(nn.Module):
    """block.py:301,305: single-block norm; returns (x, gate)."""

    def __init__(self, dim: int, lora: bool = False):
        super().__init__()
        self.silu = nn.SiLU()
        self.linear = make_linear(dim, 3 * dim, lora)
        self.norm = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)

    def forward(self, x, emb=None):
        emb = self.linear(self.silu(emb))
        shift_msa, scale_msa, gate_msa = emb.chunk(3, dim=1)
        x = self.norm(x) * (1 + scale_msa[:, None]) + shift_msa[:, None]
        return x, gate_msa


class AdaLayerNormContinuous(nn.Module):
    """transformer.py:243 norm_out(hidden, temb): chunk order is (scale, shift)."""

    def __init__(self, dim: int, cond_dim: int):
        super().__init__()
        self.silu = nn.SiLU()
        self.linear = nn.Linear(cond_dim, 2 * dim)
        self.norm = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)

    def forward(self, x, conditioning_embedding):
        emb = self.linear(self.silu(conditioning_embedding).to(x.dtype))
        scale, shift = torch.chunk(emb, 2, dim=1)
        return self.norm(x) * (1 + scale)[:, None, :] + shift[:, None, :]


# ----------------------------------------------------------------------------------------
# feed-forward
# ----------------------------------------------------------------------------------------
class GELUProj(nn.Module):
    """diffusers.models.activations.GELU(dim_in, dim_out, approximate="tanh")."""

    def __init__(self, i: int, o: int):
        super().__init__()
        self.proj = nn.Linear(i, o)

    def forward(self, x):
        return F.gelu(self.proj(x), approximate="tanh")


class FeedForward(nn.Module):
    """block.py:256-266 ff / ff_context; LoRA target is net[2] (seed_512.yaml:38)."""

    def __init__(self, dim: int, mult: int = 4, lora_out: bool = False):
        super().__init__()
        self.net = nn.ModuleList([GELUProj(dim, dim * mult), nn.Dropout(0.0),
                                  make_linear(dim * mult, dim, lora_out)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


# ----------------------------------------------------------------------------------------
# embeddings
# ----------------------------------------------------------------------------------------
def get_timestep_embedding(t: torch.Tensor, dim: int = 256, flip_sin_to_cos: bool = True,
                           downscale_freq_shift: float = 0.0, scale: float = 1.0,
                           max_period: int = 10000) -> torch.Tensor:
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device)
    exponent = exponent / (half - downscale_freq_shift)
    emb = torch.exp(exponent)
    emb = t[:, None].float() * emb[None, :]
    emb = scale * emb
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


class TimestepEmbedding(nn.Module):
    def __init__(self, i: int, o: int):
        super().__init__()
        self.linear_1 = nn.Linear(i, o)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(o, o)

    def forward(self, x):
        return self.linear_2(self.act(self.linear_1(x)))


class CombinedTimestepGuidanceTextProjEmbeddings(nn.Module):
    """transformer.py:102-114 time_text_embed(timestep, guidance, pooled)."""

    def __init__(self, dim: int, pooled_dim: int):
        super().__init__()
        self.timestep_embedder = TimestepEmbedding(256, dim)
        self.guidance_embedder = TimestepEmbedding(256, dim)
        self.text_embedder = TimestepEmbedding(pooled_dim, dim)  # PixArtAlphaTextProjection(act="silu")

    def forward(self, timestep, guidance, pooled_projection):
        t = self.timestep_embedder(get_timestep_embedding(timestep).to(pooled_projection.dtype))
        g = self.guidance_embedder(get_timestep_embedding(guidance).to(pooled_projection.dtype))
        return t + g + self.text_embedder(pooled_projection)


class CombinedTimestepTextProjEmbeddings(nn.Module):
    """guidance_embeds=False variant (FLUX.1-schnell): time_text_embed(timestep, pooled)."""

    def __init__(self, dim: int, pooled_dim: int):
        super().__init__()
        self.timestep_embedder = TimestepEmbedding(256, dim)
        self.text_embedder = TimestepEmbedding(pooled_dim, dim)

    def forward(self, timestep, pooled_projection):
        t = self.timestep_embedder(get_timestep_embedding(timestep).to(pooled_projection.dtype))
        return t + self.text_embedder(pooled_projection)


def rope_tables(ids: torch.Tensor, axes_dim: Sequence[int] = (16, 56, 56), theta: float = 10000.0
                ) -> Tuple[torch.Tensor, torch.Tensor]:
    """FluxPosEmbed.forward restated (transformer.py:131,134): float64 frequencies,
    repeat_interleave(2) real layout, cast to float32."""
    pos = ids.float()
    cos_out, sin_out = [], []
    for a, d in enumerate(axes_dim):
        freqs = 1.0 / (theta ** (torch.arange(0, d, 2, dtype=torch.float64, device=pos.device)[: d // 2] / d))
        f = torch.outer(pos[:, a].to(torch.float64), freqs)
        cos_out.append(f.cos().repeat_interleave(2, dim=1).float())
        sin_out.append(f.sin().repeat_interleave(2, dim=1).float())
    return torch.cat(cos_out, dim=-1), torch.cat(sin_out, dim=-1)


class FluxPosEmbed(nn.Module):
    def __init__(self, theta: float = 10000.0, axes_dim: Sequence[int] = (16, 56, 56)):
        super().__init__()
        self.theta, self.axes_dim = theta, tuple(axes_dim)

    def forward(self, ids):
        return rope_tables(ids, self.axes_dim, self.theta)


def apply_rotary_emb(x: torch.Tensor, freqs_cis, use_real: bool = True, use_real_unbind_dim: int = -1):
    """diffusers.models.embeddings.apply_rotary_emb (block.py:75-78, 97-99)."""
    cos, sin = freqs_cis
    cos, sin = cos[None, None].to(x.device), sin[None, None].to(x.device)
    x_real, x_imag = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    x_rot = torch.stack([-x_imag, x_real], dim=-1).flatten(3)
    return (x.float() * cos + x_rot.float() * sin).to(x.dtype)


# ----------------------------------------------------------------------------------------
# attention module + blocks + transformer
# ----------------------------------------------------------------------------------------
class Attention(nn.Module):
    """Attribute surface used by block.py:7-176."""

    def __init__(self, dim: int, heads: int, head_dim: int, pre_only: bool = False, lora: bool = False):
        super().__init__()
        self.heads = heads
        inner = heads * head_dim
        self.to_q = make_linear(dim, inner, lora)
        self.to_k = make_linear(dim, inner, lora)
        self.to_v = make_linear(dim, inner, lora)
        self.norm_q = RMSNorm(head_dim)
        self.norm_k = RMSNorm(head_dim)
        if not pre_only:
            self.add_q_proj = nn.Linear(dim, inner)
            self.add_k_proj = nn.Linear(dim, inner)
            self.add_v_proj = nn.Linear(dim, inner)
            self.norm_added_q = RMSNorm(head_dim)
            self.norm_added_k = RMSNorm(head_dim)
            self.to_out = nn.ModuleList([make_linear(inner, dim, lora), nn.Dropout(0.0)])
            self.to_add_out = nn.Linear(inner, dim)


class FluxTransformerBlock(nn.Module):
    def __init__(self, dim: int, heads: int, head_dim: int, lora: bool = False):
        super().__init__()
        self.norm1 = AdaLayerNormZero(dim, lora)
        self.norm1_context = AdaLayerNormZero(dim, False)
        self.attn = Attention(dim, heads, head_dim, pre_only=False, lora=lora)
        self.norm2 = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.ff = FeedForward(dim, 4, lora_out=lora)
        self.norm2_context = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.ff_context = FeedForward(dim, 4, lora_out=False)


class FluxSingleTransformerBlock(nn.Module):
    def __init__(self, dim: int, heads: int, head_dim: int, lora: bool = False):
        super().__init__()
        self.norm = AdaLayerNormZeroSingle(dim, lora)
        self.proj_mlp = make_linear(dim, 4 * dim, lora)
        self.act_mlp = nn.GELU(approximate="tanh")
        self.proj_out = make_linear(5 * dim, dim, lora)
        self.attn = Attention(dim, heads, head_dim, pre_only=True, lora=lora)


class FluxTransformer2DModel(nn.Module):
    """Attribute surface used by transformer.py:47-252 (+ `.config` read at generate.py:262,322)."""

    def __init__(self, num_layers: int = 19, num_single_layers: int = 38, heads: int = 24,
                 head_dim: int = 128, in_channels: int = 64, joint_dim: int = 4096,
                 pooled_dim: int = 768, guidance_embeds: bool = True,
                 axes_dims_rope: Sequence[int] = (16, 56, 56), lora: bool = False):
        super().__init__()
        dim = heads * head_dim
        self.config = SimpleNamespace(num_layers=num_layers, num_single_layers=num_single_layers,
                                      num_attention_heads=heads, attention_head_dim=head_dim,
                                      in_channels=in_channels, joint_attention_dim=joint_dim,
                                      pooled_projection_dim=pooled_dim, guidance_embeds=guidance_embeds,
                                      axes_dims_rope=tuple(axes_dims_rope))
        self.inner_dim = dim
        self.pos_embed = FluxPosEmbed(10000.0, axes_dims_rope)
        self.time_text_embed = (CombinedTimestepGuidanceTextProjEmbeddings(dim, pooled_dim)
                                if guidance_embeds else CombinedTimestepTextProjEmbeddings(dim, pooled_dim))
        self.context_embedder = nn.Linear(joint_dim, dim)
        self.x_embedder = make_linear(in_channels, dim, lora)
        self.transformer_blocks = nn.ModuleList(
            [FluxTransformerBlock(dim, heads, head_dim, lora) for _ in range(num_layers)])
        self.single_transformer_blocks = nn.ModuleList(
            [FluxSingleTransformerBlock(dim, heads, head_dim, lora) for _ in range(num_single_layers)])
        self.norm_out = AdaLayerNormContinuous(dim, dim)
        self.proj_out = nn.Linear(dim, in_channels)
        self.gradient_checkpointing = False


def init_synthetic_(model: nn.Module, seed: int = 0, std: float = 0.02, bias_std: float = 0.0,
                    norm_jitter: float = 0.0) -> nn.Module:
    """Synthetic weights per BASELINE.md §4: Linear ~ N(0, std^2), bias 0 (or N(0,bias_std^2)),
    RMSNorm weights 1 (+jitter for tests that must catch a dropped weight)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() >= 2:
                p.copy_(torch.randn(p.shape, generator=g) * std)
            elif name.endswith("bias"):
                p.copy_(torch.randn(p.shape, generator=g) * bias_std if bias_std else torch.zeros_like(p))
            else:  # 1-D norm weights
                p.copy_(1.0 + norm_jitter * torch.randn(p.shape, generator=g))
    return model


# ----------------------------------------------------------------------------------------
# scheduler + latent utilities (generate.py:261-310, 349, 371-378; pipeline_tools.py:7-30)
# ----------------------------------------------------------------------------------------
def calculate_shift(image_seq_len, base_seq_len: int = 256, max_seq_len: int = 4096,
                    base_shift: float = 0.5, max_shift: float = 1.16):
    m = (max_shift - base_shift) / (max_seq_len - base_seq_len)
    b = base_shift - m * base_seq_len
    return image_seq_len * m + b


class FlowMatchEulerDiscreteScheduler:
    """FLUX.1-dev config: dynamic shifting, base_shift .5, max_shift 1.15, seq 256..4096."""

    order = 1

    def __init__(self, num_train_timesteps: int = 1000, shift: float = 3.0, use_dynamic_shifting: bool = True,
                 base_shift: float = 0.5, max_shift: float = 1.15, base_image_seq_len: int = 256,
                 max_image_seq_len: int = 4096):
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, shift=shift,
                                      use_dynamic_shifting=use_dynamic_shifting, base_shift=base_shift,
                                      max_shift=max_shift, base_image_seq_len=base_image_seq_len,
                                      max_image_seq_len=max_image_seq_len)
        self.timesteps = None
        self.sigmas = None
        self._step_index = None

    @staticmethod
    def time_shift(mu: float, sigma: float, t):
        return math.exp(mu) / (math.exp(mu) + (1 / t - 1) ** sigma)

    def set_timesteps(self, num_inference_steps=None, device=None, sigmas=None, mu=None):
        if sigmas is None:
            ts = np.linspace(self.config.num_train_timesteps, 1, num_inference_steps)
            sigmas = ts / self.config.num_train_timesteps
        sigmas = np.asarray(sigmas, dtype=np.float64)
        if self.config.use_dynamic_shifting:
            sigmas = self.time_shift(mu, 1.0, sigmas)
        else:
            s = self.config.shift
            sigmas = s * sigmas / (1 + (s - 1) * sigmas)
        sig = torch.from_numpy(np.asarray(sigmas)).to(dtype=torch.float32, device=device)
        self.timesteps = sig * self.config.num_train_timesteps
        self.sigmas = torch.cat([sig, torch.zeros(1, device=sig.device)])
        self._step_index = 0

    def step(self, model_output, timestep, sample, return_dict: bool = False):
        sample = sample.to(torch.float32)
        sigma, sigma_next = self.sigmas[self._step_index], self.sigmas[self._step_index + 1]
        prev = sample + (sigma_next - sigma) * model_output
        prev = prev.to(model_output.dtype)
        self._step_index += 1
        return (prev,)


def retrieve_timesteps(scheduler, num_inference_steps=None, device=None, timesteps=None, sigmas=None, **kw):
    if timesteps is not None:
        raise ValueError("custom timesteps are not supported by FlowMatchEulerDiscreteScheduler")
    if sigmas is not None:
        scheduler.set_timesteps(sigmas=sigmas, device=device, **kw)
    else:
        scheduler.set_timesteps(num_inference_steps, device=device, **kw)
    return scheduler.timesteps, len(scheduler.timesteps)


def pack_latents(latents: torch.Tensor) -> torch.Tensor:
    """FluxPipeline._pack_latents: [B,C,H,W] -> [B,(H/2)(W/2),4C]."""
    b, c, h, w = latents.shape
    x = latents.view(b, c, h // 2, 2, w // 2, 2).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(b, (h // 2) * (w // 2), c * 4)


def unpack_latents(latents: torch.Tensor, height: int, width: int, vae_scale_factor: int = 16) -> torch.Tensor:
    b, n, ch = latents.shape
    h, w = height // vae_scale_factor, width // vae_scale_factor
    x = latents.view(b, h, w, ch // 4, 2, 2).permute(0, 3, 1, 4, 2, 5)
    return x.reshape(b, ch // 4, h * 2, w * 2)


def prepare_latent_image_ids(h2: int, w2: int, dtype=torch.float32) -> torch.Tensor:
    """ids[...,1]=row, ids[...,2]=col on the packed (h2 x w2) grid -> [h2*w2, 3]."""
    ids = torch.zeros(h2, w2, 3)
    ids[..., 1] = ids[..., 1] + torch.arange(h2)[:, None]
    ids[..., 2] = ids[..., 2] + torch.arange(w2)[None, :]
    return ids.reshape(h2 * w2, 3).to(dtype)
