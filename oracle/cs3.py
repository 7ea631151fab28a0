"""Oracle: CS3 (Cross-Scale State Space) encoders + DGF (DUAN) fusion, restated on torch-CPU.

TEST INFRASTRUCTURE.  Restates /root/reference/src/train/model.py:
  EEGEncoder :16-134, PPGEncoder :137-205, FNIRSEncoder :208-274, MotionEncoder :277-343,
  FeaturePyramidPooling :345-373, OminiModel.spatial_pyramid_pooling :479-511,
  DUAN :947-1035, fuse_eeg :731-755, fuse_fnirs :757-779, and the generate()-side
  fusion branch src/flux/generate.py:213-258 (with the SURVEY Q1/Q2 decisions).
DUAN / FeaturePyramidPooling / fuse_* / spatial_pyramid_pooling are pinned by goldens made
from the real reference classes (tests/golden/cs3_*.npz); the S4 part is unpinned (oracle/s4.py).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from .s4 import S4Model


def adaptive_avg_pool1d(x: torch.Tensor, out: int) -> torch.Tensor:
    """nn.AdaptiveAvgPool1d bin rule restated: start=floor(i*L/out), end=ceil((i+1)*L/out)."""
    L = x.shape[-1]
    cols = []
    for i in range(out):
        s = (i * L) // out
        e = -((-(i + 1) * L) // out)
        cols.append(x[..., s:e].mean(-1))
    return torch.stack(cols, dim=-1)


class FeaturePyramidPooling(nn.Module):
    def __init__(self, output_sizes: Sequence[int]):
        super().__init__()
        self.output_sizes = list(output_sizes)

    def forward(self, x):
        return torch.cat([adaptive_avg_pool1d(x, s) for s in self.output_sizes], dim=-1)


def spatial_pyramid_pooling(x: torch.Tensor, output_size: int, adaptive: bool = False) -> torch.Tensor:
    b, c, l = x.shape
    if l == output_size:
        return x
    if adaptive:
        return adaptive_avg_pool1d(x, output_size)
    if l < output_size:
        return torch.cat([x, x.new_zeros(b, c, output_size - l)], dim=2)
    return x[:, :, :output_size]


def _mlp_head(d_in: int, d_hid: int, d_out: int, expand: Optional[int]) -> nn.Sequential:
    """Linear-LN-ReLU-(Dropout=id)-Linear-LN-ReLU[-Unflatten(512,8)-Linear(8,expand)]."""
    layers: List[nn.Module] = [nn.Flatten(start_dim=1), nn.Linear(d_in, d_hid), nn.LayerNorm(d_hid), nn.ReLU(),
                               nn.Dropout(0.3), nn.Linear(d_hid, d_out), nn.LayerNorm(d_out), nn.ReLU(),
                               nn.Dropout(0.3)]
    if expand:
        layers += [nn.Unflatten(1, (512, 8)), nn.Linear(8, expand)]
    return nn.Sequential(*layers)


class EEGEncoder(nn.Module):
    """[B,4,4096] -> [B,512,4096]  (model.py:16-134)."""
    fixed_length = 4096

    def __init__(self, g=None, g2=None):
        """g / g2: generators the two S4Model instances draw their S4 parameters from (g2 defaults to continuing g)."""
        super().__init__()
        self.s41 = S4Model(4, 64, 64, 2, 64, 4096, g)
        self.s42 = S4Model(4, 4, 4, 2, 4, 4096, g if g2 is None else g2)
        self.fpp = FeaturePyramidPooling([128, 256, 512, 1024, 2048])
        self.projection = _mlp_head(4 * 4096, 2048, 4096, 4096)

    def forward(self, x):
        z1 = self.s41(x.permute(0, 2, 1)).permute(0, 2, 1)          # [B,64,4096]
        z1 = adaptive_avg_pool1d(z1, 4).permute(0, 2, 1)            # [B,4,64]
        z2 = self.s42(x.permute(0, 2, 1)).permute(0, 2, 1)          # [B,4,4096]
        z2 = adaptive_avg_pool1d(z2, 64)                            # [B,4,64]
        comb = torch.cat([z1, self.fpp(x), z2], dim=-1)             # [B,4,4096]
        return self.projection(comb)


class _FlatEncoder(nn.Module):
    def __init__(self, ch, L, pool, fpp_sizes, d_hid, d_out, expand, g=None):
        super().__init__()
        self.s4 = S4Model(ch, ch, ch, 2, ch, L, g)
        self.pool_size = pool
        self.fpp = FeaturePyramidPooling(fpp_sizes)
        self.projection = _mlp_head(ch * pool + ch * sum(fpp_sizes), d_hid, d_out, expand)

    def forward(self, x):
        z = self.s4(x.permute(0, 2, 1)).permute(0, 2, 1)
        z = adaptive_avg_pool1d(z, self.pool_size)
        comb = torch.cat([z.flatten(1), self.fpp(x).flatten(1)], dim=1)
        return self.projection(comb)


class PPGEncoder(_FlatEncoder):
    """[B,4,256] -> [B,512,4096] (model.py:137-205)."""
    fixed_length = 256

    def __init__(self, g=None):
        super().__init__(4, 256, 16, [64, 128, 256], 1024, 4096, 4096, g)


class FNIRSEncoder(_FlatEncoder):
    """[B,6,512] -> [B,768] (model.py:208-274)."""
    fixed_length = 512

    def __init__(self, g=None):
        super().__init__(6, 512, 32, [128, 256, 448], 1024, 768, None, g)


class MotionEncoder(_FlatEncoder):
    """[B,6,128] -> [B,768] (model.py:277-343)."""
    fixed_length = 128

    def __init__(self, g=None):
        super().__init__(6, 128, 6, [32, 64, 124], 512, 768, None, g)


class DUAN(nn.Module):
    """DGF core (model.py:947-1035): gate-mixed instance/layer statistics, gamma/beta
    modulation from the pooled condition, top-k channel mask."""

    def __init__(self, channels: int, hidden_dim: int = 128, keep_ratio: float = 0.7, eps: float = 1e-3):
        super().__init__()
        self.channels, self.hidden_dim, self.keep_ratio, self.eps = channels, hidden_dim, keep_ratio, eps
        self.gate = nn.Sequential(nn.Conv1d(channels, hidden_dim, 1), nn.ReLU(),
                                  nn.Conv1d(hidden_dim, channels, 1), nn.Sigmoid())
        self.mlp = nn.Sequential(nn.Conv1d(channels, hidden_dim, 1), nn.ReLU(),
                                 nn.Conv1d(hidden_dim, channels * 2, 1))

    def forward(self, x16, c16, keep_ratio=None):
        x, c = x16.float(), c16.float()
        B, C, L = x.shape
        kr = self.keep_ratio if keep_ratio is None else keep_ratio
        mu_c = x.mean(2, keepdim=True)
        sig_c = torch.sqrt(x.var(2, unbiased=False, keepdim=True) + self.eps)
        mu_l = x.mean((1, 2), keepdim=True).expand(B, C, 1)
        sig_l = torch.sqrt(x.var((1, 2), unbiased=False, keepdim=True).expand(B, C, 1) + self.eps)
        g = self.gate(c).mean(2, keepdim=True)
        mu = g * mu_c + (1 - g) * mu_l
        sig = g * sig_c + (1 - g) * sig_l
        xh = (x - mu) / sig
        gb = self.mlp(c.mean(2, keepdim=True))
        gamma, beta = gb.chunk(2, dim=1)
        y = (1 + gamma) * xh + beta
        imp = y.abs().mean(2)
        k = max(1, int(C * kr))
        idx = torch.topk(imp, k, dim=1).indices
        mask = torch.zeros_like(imp).scatter_(1, idx, 1.0)
        return (y * mask.unsqueeze(2)).to(x16.dtype)


def fuse_eeg(duan1: DUAN, fusion1: nn.Module, eeg_f, ppg_f):
    """model.py:731-755."""
    f = duan1(ppg_f, eeg_f)
    f = torch.cat([eeg_f, f], dim=1).transpose(1, 2)
    return fusion1(f).transpose(1, 2)


def fuse_fnirs(duan2: DUAN, fusion2: nn.Module, fnirs_f, motion_f):
    """model.py:757-779."""
    a, m = fnirs_f.unsqueeze(1), motion_f.unsqueeze(1)
    f = duan2(a, m)
    return fusion2(torch.cat([a, f], dim=-1)).squeeze(1)


class CS3DGF(nn.Module):
    """Brain-side modules of OminiModel (model.py:430-462) with the generate() branch."""

    def __init__(self, seed: int = 0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.eeg_projection, self.ppg_projection = EEGEncoder(g), PPGEncoder(g)
        self.fnirs_projection, self.motion_projection = FNIRSEncoder(g), MotionEncoder(g)
        self.fusion1 = nn.Sequential(nn.Linear(1024, 512))
        self.fusion2 = nn.Sequential(nn.Linear(1536, 768))
        self.fusion3 = nn.Sequential(nn.Linear(1024, 512))
        self.fusion4 = nn.Sequential(nn.Linear(1536, 768))
        self.duan_norm1, self.duan_norm2 = DUAN(512), DUAN(1)
        self.duan_norm_prompt, self.duan_norm_pooled = DUAN(512), DUAN(1)
        self.eeg_fixed_length, self.fnirs_fixed_length = 4096, 512
        self.ppg_fixed_length, self.motion_fixed_length = 256, 128

    def brain_embeds(self, prompt_embeds, pooled, eeg=None, fnirs=None, ppg=None, motion=None,
                     fuse_flag: bool = False, per_stream: bool = True):
        """generate.py:168-258 restated on batched [B,C,L] signals.
        per_stream=True is SURVEY Q2 (each side replaces its own embedding);
        per_stream=False is the literal reference rule (replace only if BOTH sides exist)."""
        pe_b = pp_b = None
        if eeg is not None:
            e = self.eeg_projection(spatial_pyramid_pooling(eeg, self.eeg_fixed_length))
            if ppg is not None:
                p = self.ppg_projection(spatial_pyramid_pooling(ppg, self.ppg_fixed_length))
                pe_b = fuse_eeg(self.duan_norm1, self.fusion1, e, p)
            else:
                pe_b = e
        if fnirs is not None:
            f = self.fnirs_projection(spatial_pyramid_pooling(fnirs, self.fnirs_fixed_length))
            if motion is not None:
                m = self.motion_projection(spatial_pyramid_pooling(motion, self.motion_fixed_length))
                pp_b = fuse_fnirs(self.duan_norm2, self.fusion2, f, m)
            else:
                pp_b = f
        if fuse_flag and pe_b is not None and pp_b is not None:
            prompt_embeds = self.duan_norm_prompt(prompt_embeds, pe_b)
            pooled = self.duan_norm_pooled(pooled.unsqueeze(1), pp_b.unsqueeze(1)).squeeze(1)
        elif per_stream:
            prompt_embeds = pe_b if pe_b is not None else prompt_embeds
            pooled = pp_b if pp_b is not None else pooled
        elif pe_b is not None and pp_b is not None:
            prompt_embeds, pooled = pe_b, pp_b
        return prompt_embeds, pooled
