"""Oracle: diffusers==0.31.0 `AutoencoderKL` as FLUX.1 configures it, restated with torch.nn -- TEST INFRASTRUCTURE.

The reference calls it at src/flux/generate.py:375-380 (`vae.decode(latents / scaling_factor + shift_factor)`) and
src/flux/pipeline_tools.py:8-14 (`vae.encode(images).latent_dist.sample()`, then `(x - shift_factor) * scaling_factor`); the
class itself lives in diffusers, which is absent from /root/reference and from this image, so this is a restatement of the
published architecture -- **parity unpinned** (no reference-generated golden can exist for it). Module and parameter names
are diffusers' (`encoder.down_blocks.N.resnets.M.conv1.weight`, `decoder.up_blocks.N.upsamplers.0.conv.weight`,
`*.mid_block.attentions.0.to_q.weight` ...), so a real `vae/diffusion_pytorch_model.safetensors` loads into both sides.

FLUX.1 VAE config (vae/config.json): in/out_channels 3, latent_channels 16, block_out_channels (128, 256, 512, 512),
layers_per_block 2, norm_num_groups 32, act silu, mid_block_add_attention true, use_quant_conv / use_post_quant_conv false,
scaling_factor 0.3611, shift_factor 0.1159, force_upcast true.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F


class ResnetBlock2D(nn.Module):
    """ResnetBlock2D(temb_channels=None, groups=32, eps=1e-6, output_scale_factor=1)."""

    def __init__(self, cin: int, cout: int, groups: int = 32, eps: float = 1e-6):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps, affine=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h


class AttnBlock(nn.Module):
    """diffusers Attention(heads=1, dim_head=C, norm_num_groups=32, residual_connection=True, bias=True) over H*W tokens."""

    def __init__(self, c: int, groups: int = 32, eps: float = 1e-6):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=eps, affine=True)
        self.to_q, self.to_k, self.to_v = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Dropout(0.0)])

    def forward(self, x):
        b, c, h, w = x.shape
        t = self.group_norm(x).view(b, c, h * w).transpose(1, 2)
        q, k, v = self.to_q(t), self.to_k(t), self.to_v(t)
        o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]
        o = self.to_out[0](o).transpose(1, 2).reshape(b, c, h, w)
        return x + o


class _Sampler(nn.Module):
    def __init__(self, c: int, down: bool):
        super().__init__()
        self.down = down
        self.conv = nn.Conv2d(c, c, 3, stride=2 if down else 1, padding=0 if down else 1)

    def forward(self, x):
        if self.down:
            return self.conv(F.pad(x, (0, 1, 0, 1), mode="constant", value=0))
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class _Block(nn.Module):
    def __init__(self, cin, cout, n_res, sampler: Optional[str], groups):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, groups) for i in range(n_res)])
        if sampler == "down":
            self.downsamplers = nn.ModuleList([_Sampler(cout, True)])
        elif sampler == "up":
            self.upsamplers = nn.ModuleList([_Sampler(cout, False)])
        self.sampler = sampler

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.sampler == "down":
            x = self.downsamplers[0](x)
        elif self.sampler == "up":
            x = self.upsamplers[0](x)
        return x


class _Mid(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, groups), ResnetBlock2D(c, c, groups)])
        self.attentions = nn.ModuleList([AttnBlock(c, groups)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class Encoder(nn.Module):
    def __init__(self, cin, latent, chans: Sequence[int], layers, groups):
        super().__init__()
        self.conv_in = nn.Conv2d(cin, chans[0], 3, padding=1)
        blocks, c = [], chans[0]
        for i, co in enumerate(chans):
            blocks.append(_Block(c, co, layers, "down" if i < len(chans) - 1 else None, groups))
            c = co
        self.down_blocks = nn.ModuleList(blocks)
        self.mid_block = _Mid(c, groups)
        self.conv_norm_out = nn.GroupNorm(groups, c, eps=1e-6)
        self.conv_out = nn.Conv2d(c, 2 * latent, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        x = self.mid_block(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class Decoder(nn.Module):
    def __init__(self, cout, latent, chans: Sequence[int], layers, groups):
        super().__init__()
        rev = list(reversed(chans))
        self.conv_in = nn.Conv2d(latent, rev[0], 3, padding=1)
        self.mid_block = _Mid(rev[0], groups)
        blocks, c = [], rev[0]
        for i, co in enumerate(rev):
            blocks.append(_Block(c, co, layers + 1, "up" if i < len(rev) - 1 else None, groups))
            c = co
        self.up_blocks = nn.ModuleList(blocks)
        self.conv_norm_out = nn.GroupNorm(groups, c, eps=1e-6)
        self.conv_out = nn.Conv2d(c, cout, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class DiagonalGaussianDistribution:
    def __init__(self, parameters: torch.Tensor):
        self.mean, logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator: Optional[torch.Generator] = None, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        if noise is None:
            noise = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean


class AutoencoderKL(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, latent_channels=16, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 norm_num_groups=32, scaling_factor=0.3611, shift_factor=0.1159):
        super().__init__()
        self.encoder = Encoder(in_channels, latent_channels, block_out_channels, layers_per_block, norm_num_groups)
        self.decoder = Decoder(out_channels, latent_channels, block_out_channels, layers_per_block, norm_num_groups)
        self.config = SimpleNamespace(in_channels=in_channels, out_channels=out_channels, latent_channels=latent_channels,
                                      block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                                      norm_num_groups=norm_num_groups, scaling_factor=scaling_factor, shift_factor=shift_factor)

    def encode(self, x, return_dict: bool = True):
        d = DiagonalGaussianDistribution(self.encoder(x))
        return SimpleNamespace(latent_dist=d) if return_dict else (d,)

    def decode(self, z, return_dict: bool = True):
        y = self.decoder(z)
        return SimpleNamespace(sample=y) if return_dict else (y,)


def init_synthetic_(vae: nn.Module, seed: int = 0) -> nn.Module:
    """Variance-preserving random weights (fan-in scaled convs / linears, jittered norm affines) so that activations stay O(1)
    through the 30-odd layers and every parameter matters to the output."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in vae.named_parameters():
            if p.dim() >= 2:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g) * (0.7 / fan_in ** 0.5))
            elif name.endswith("bias"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
            else:
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
    return vae.eval()
