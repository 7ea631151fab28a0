"""FLUX VAE (diffusers `AutoencoderKL`) on MI355X -- the step either side of the denoise loop.

Reference call sites: `vae.decode(latents / scaling_factor + shift_factor)` (src/flux/generate.py:375-380) and
`vae.encode(images).latent_dist.sample()` (src/flux/pipeline_tools.py:8-14). Same surface here: `.config`, `.encode(x)`
-> object with `.latent_dist.sample() / .mode()`, `.decode(z, return_dict=False)` -> `(image,)`, NCHW tensors in and out.

Inside, activations are NHWC: the residual stream is fp32 `[B*H*W, C]`, GEMM operands bf16. Every convolution is an implicit
GEMM on the DiT's MFMA kernel (`lx_im2col3x3` rows x `[Cout, 9 Cin]` weights, bias / fp32-residual epilogues fused), the
mid-block attention is two GEMMs around `lx_softmax_rows`, GroupNorm+SiLU is `lx_groupnorm_silu` (csrc/vae.hip). torch owns
buffers and the NCHW<->NHWC layout moves only. Parameter names are diffusers' (`decoder.up_blocks.0.resnets.1.conv1.weight` ...),
so `vae/diffusion_pytorch_model.safetensors` of a FLUX.1 checkpoint loads as is.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, Optional, Tuple

import torch

from . import ops
from .ops import LX_EPI_RESID_F32, LX_EPI_STORE_BF16, LX_EPI_STORE_F32

FLUX_VAE_CONFIG = dict(in_channels=3, out_channels=3, latent_channels=16, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                       norm_num_groups=32, scaling_factor=0.3611, shift_factor=0.1159)


def _pad_to(n: int, m: int) -> int:
    return (n + m - 1) // m * m


class DiagonalGaussianDistribution:
    """diffusers DiagonalGaussianDistribution: mean / clamped logvar from the encoder's 2*latent channels."""

    def __init__(self, parameters: torch.Tensor):
        self.mean, logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator: Optional[torch.Generator] = None, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        if noise is None:
            noise = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)
        return self.mean + self.std * noise.to(self.mean.device)

    def mode(self) -> torch.Tensor:
        return self.mean


class LxAutoencoderKL:
    def __init__(self, state_dict: Dict[str, torch.Tensor], config: Optional[dict] = None, device="cuda"):
        cfg = dict(FLUX_VAE_CONFIG)
        cfg.update(config or {})
        cfg["block_out_channels"] = tuple(cfg["block_out_channels"])
        self.config = SimpleNamespace(**cfg)
        self.device = torch.device(device)
        self.dtype = torch.float32
        self.G = cfg["norm_num_groups"]
        self.w: Dict[str, torch.Tensor] = {}
        sd = state_dict
        for k, v in sd.items():
            if k.endswith(".weight") and v.dim() == 4:
                self._pack_conv(k[: -len(".weight")], v, sd.get(k[: -len("weight")] + "bias"))
            elif k.endswith(".weight") and v.dim() == 2:
                self._pack_conv(k[: -len(".weight")], v[:, :, None, None], sd.get(k[: -len("weight")] + "bias"))
            elif k.endswith(".weight") and v.dim() == 1:                        # GroupNorm affine
                self.w[k[: -len(".weight")] + ".g"] = v.detach().to(self.device, torch.float32).contiguous()
                self.w[k[: -len(".weight")] + ".b"] = sd[k[: -len("weight")] + "bias"].detach().to(self.device, torch.float32).contiguous()

    # ---- weights -------------------------------------------------------------------------------------------------------
    def _pack_conv(self, name: str, w: torch.Tensor, b: Optional[torch.Tensor]) -> None:
        """[Cout, Cin, kh, kw] -> bf16 [Npad, Kpad] with column = (dy*kw + dx)*Cin + c (the im2col order), zero padded to the
        GEMM's granules (K % 64, N % 8)."""
        co, ci, kh, kw = w.shape
        W = w.detach().to(self.device, torch.float32).permute(0, 2, 3, 1).reshape(co, kh * kw * ci)
        Np, Kp = _pad_to(co, 8), _pad_to(kh * kw * ci, 64)
        Wp = torch.zeros(Np, Kp, dtype=torch.bfloat16, device=self.device)
        Wp[:co, : kh * kw * ci] = W.to(torch.bfloat16)
        bp = torch.zeros(Np, dtype=torch.float32, device=self.device)
        if b is not None:
            bp[:co] = b.detach().to(self.device, torch.float32)
        self.w[name + ".w"], self.w[name + ".bias"] = Wp, bp
        self.w[name + ".shape"] = (co, ci, kh)

    # ---- building blocks (x: fp32 [B*H*W, C]) ----------------------------------------------------------------------------
    def _gn(self, x: torch.Tensor, B: int, name: str, silu: bool = True) -> torch.Tensor:
        C = x.shape[1]
        y = torch.empty(x.shape, dtype=torch.bfloat16, device=self.device)
        ops.groupnorm_silu(x.view(B, -1, C), self.w[name + ".g"], self.w[name + ".b"], y.view(B, -1, C), self.G, 1e-6, silu)
        return y

    def _conv3(self, h: torch.Tensor, B: int, H: int, W: int, name: str, mode: int = 0, out: Optional[torch.Tensor] = None,
               epilogue: int = LX_EPI_STORE_F32) -> Tuple[torch.Tensor, int, int]:
        """3x3 convolution of bf16 NHWC h [B*H*W, Cin] as im2col + GEMM; image by image so the im2col scratch stays one image
        large. out: fp32 [B*Ho*Wo, Npad] (epilogue STORE_F32 / RESID_F32 accumulates into it) or bf16 (STORE_BF16)."""
        Wt, bias = self.w[name + ".w"], self.w[name + ".bias"]
        Cin = h.shape[1]
        Ho, Wo = (H // 2, W // 2) if mode == 1 else ((2 * H, 2 * W) if mode == 2 else (H, W))
        Np, Kp = Wt.shape
        if out is None:
            out = torch.empty(B * Ho * Wo, Np, dtype=torch.bfloat16 if epilogue == LX_EPI_STORE_BF16 else torch.float32, device=self.device)
        cols = torch.empty(Ho * Wo, Kp, dtype=torch.bfloat16, device=self.device)
        hv = h.view(B, H, W, Cin)
        for b in range(B):
            ops.im2col3x3(hv[b:b + 1], cols, mode)
            ops.gemm([ops.gemm_desc(cols, Wt, out[b * Ho * Wo:(b + 1) * Ho * Wo], bias=bias, epilogue=epilogue)])
        return out, Ho, Wo

    def _lin(self, a: torch.Tensor, name: str, out: Optional[torch.Tensor] = None, epilogue: int = LX_EPI_STORE_BF16) -> torch.Tensor:
        Wt, bias = self.w[name + ".w"], self.w[name + ".bias"]
        if out is None:
            out = torch.empty(a.shape[0], Wt.shape[0], dtype=torch.bfloat16 if epilogue == LX_EPI_STORE_BF16 else torch.float32, device=self.device)
        ops.gemm([ops.gemm_desc(a, Wt, out, bias=bias, epilogue=epilogue)])
        return out

    def _resnet(self, x: torch.Tensor, B: int, H: int, W: int, p: str) -> torch.Tensor:
        h = self._gn(x, B, p + ".norm1")
        h1, _, _ = self._conv3(h, B, H, W, p + ".conv1")
        t = self._gn(h1, B, p + ".norm2")
        if (p + ".conv_shortcut.w") in self.w:
            xb = torch.empty(x.shape, dtype=torch.bfloat16, device=self.device)
            ops.convert(xb, x)
            x = self._lin(xb, p + ".conv_shortcut", epilogue=LX_EPI_STORE_F32)
        self._conv3(t, B, H, W, p + ".conv2", out=x, epilogue=LX_EPI_RESID_F32)          # x += conv2(t)
        return x

    def _attn(self, x: torch.Tensor, B: int, P: int, p: str) -> torch.Tensor:
        """Single-head attention over the P = H*W tokens of each image (diffusers Attention, residual_connection=True)."""
        C = x.shape[1]
        n = self._gn(x, B, p + ".group_norm", silu=False)
        q, k = self._lin(n, p + ".to_q"), self._lin(n, p + ".to_k")
        Wv, bv = self.w[p + ".to_v.w"], self.w[p + ".to_v.bias"]
        S = torch.empty(P, P, dtype=torch.float32, device=self.device)
        Pm = torch.empty(P, P, dtype=torch.bfloat16, device=self.device)
        vT = torch.empty(C, P, dtype=torch.bfloat16, device=self.device)
        o = torch.empty(B * P, C, dtype=torch.bfloat16, device=self.device)
        for b in range(B):
            r = slice(b * P, (b + 1) * P)
            ops.gemm([ops.gemm_desc(q[r], k[r], S, epilogue=LX_EPI_STORE_F32)])                       # S = q k^T
            ops.softmax_rows(S, Pm, 1.0 / math.sqrt(C))
            ops.gemm([ops.gemm_desc(Wv, n[r], vT, epilogue=LX_EPI_STORE_BF16)])                       # v^T = Wv n^T (bias added below)
            ops.gemm([ops.gemm_desc(Pm, vT, o[r], bias=bv, epilogue=LX_EPI_STORE_BF16)])              # rows of P sum to 1: P (v + 1 bv^T) = P v + bv
        self._lin(o, p + ".to_out.0", out=x, epilogue=LX_EPI_RESID_F32)
        return x

    def _mid(self, x, B, H, W, p):
        x = self._resnet(x, B, H, W, p + ".resnets.0")
        x = self._attn(x, B, H * W, p + ".attentions.0")
        return self._resnet(x, B, H, W, p + ".resnets.1")

    def _nhwc_bf16(self, t: torch.Tensor) -> torch.Tensor:
        return t.to(self.device, torch.float32).permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)

    # ---- public surface -------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = True, generator=None):
        """z [B, latent, h, w] -> image [B, 3, 8h, 8w] fp32."""
        cfg = self.config
        B, _, H, W = z.shape
        if (H * W) % 64:
            raise ValueError(f"the mid-block attention runs on the GEMM kernel: latent h*w={H * W} must be a multiple of 64")
        x, _, _ = self._conv3(self._nhwc_bf16(z).view(B * H * W, -1), B, H, W, "decoder.conv_in")
        x = self._mid(x, B, H, W, "decoder.mid_block")
        rev = list(reversed(cfg.block_out_channels))
        for i in range(len(rev)):
            for j in range(cfg.layers_per_block + 1):
                x = self._resnet(x, B, H, W, f"decoder.up_blocks.{i}.resnets.{j}")
            if i < len(rev) - 1:
                xb = torch.empty(x.shape, dtype=torch.bfloat16, device=self.device)
                ops.convert(xb, x)
                x, H, W = self._conv3(xb, B, H, W, f"decoder.up_blocks.{i}.upsamplers.0.conv", mode=2)
        h = self._gn(x, B, "decoder.conv_norm_out")
        y, _, _ = self._conv3(h, B, H, W, "decoder.conv_out")
        img = y[:, : cfg.out_channels].reshape(B, H, W, cfg.out_channels).permute(0, 3, 1, 2).contiguous()
        return SimpleNamespace(sample=img) if return_dict else (img,)

    @torch.no_grad()
    def encode(self, images: torch.Tensor, return_dict: bool = True):
        """images [B, 3, H, W] in [-1, 1] -> latent_dist over [B, latent, H/8, W/8]."""
        cfg = self.config
        B, _, H, W = images.shape
        nd = len(cfg.block_out_channels) - 1
        if H % (1 << nd) or W % (1 << nd) or ((H >> nd) * (W >> nd)) % 64:
            raise ValueError(f"image {H}x{W}: sides must be multiples of {1 << nd} and the latent grid a multiple of 64 tokens")
        x, _, _ = self._conv3(self._nhwc_bf16(images).view(B * H * W, -1), B, H, W, "encoder.conv_in")
        for i in range(len(cfg.block_out_channels)):
            for j in range(cfg.layers_per_block):
                x = self._resnet(x, B, H, W, f"encoder.down_blocks.{i}.resnets.{j}")
            if i < nd:
                xb = torch.empty(x.shape, dtype=torch.bfloat16, device=self.device)
                ops.convert(xb, x)
                x, H, W = self._conv3(xb, B, H, W, f"encoder.down_blocks.{i}.downsamplers.0.conv", mode=1)
        x = self._mid(x, B, H, W, "encoder.mid_block")
        h = self._gn(x, B, "encoder.conv_norm_out")
        y, _, _ = self._conv3(h, B, H, W, "encoder.conv_out")
        params = y[:, : 2 * cfg.latent_channels].reshape(B, H, W, 2 * cfg.latent_channels).permute(0, 3, 1, 2).contiguous()
        d = DiagonalGaussianDistribution(params)
        return SimpleNamespace(latent_dist=d) if return_dict else (d,)

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def requires_grad_(self, *a, **k):
        return self


class VaeImageProcessor:
    """diffusers VaeImageProcessor(vae_scale_factor=16) as FluxPipeline 0.31.0 constructs it: PIL / array / tensor -> [B,3,H,W] in
    [-1, 1] with both sides floored to a multiple of the scale factor; and back (denormalise, clamp, PIL)."""

    def __init__(self, vae_scale_factor: int = 16, do_resize: bool = True, do_normalize: bool = True):
        self.vae_scale_factor, self.do_resize, self.do_normalize = vae_scale_factor, do_resize, do_normalize

    def preprocess(self, image, height: Optional[int] = None, width: Optional[int] = None) -> torch.Tensor:
        import numpy as np
        if isinstance(image, torch.Tensor):
            x = image if image.dim() == 4 else image[None]
            x = x.float()
        else:
            imgs = image if isinstance(image, (list, tuple)) else [image]
            arrs = []
            for im in imgs:
                if hasattr(im, "convert"):                  # PIL
                    w, h = im.size
                    if self.do_resize:
                        w2 = (width or w) // self.vae_scale_factor * self.vae_scale_factor
                        h2 = (height or h) // self.vae_scale_factor * self.vae_scale_factor
                        if (w2, h2) != (w, h):
                            from PIL import Image
                            im = im.resize((w2, h2), resample=Image.LANCZOS)
                    arrs.append(np.asarray(im.convert("RGB"), dtype=np.float32) / 255.0)
                else:
                    arrs.append(np.asarray(im, dtype=np.float32))
            x = torch.from_numpy(np.stack(arrs)).permute(0, 3, 1, 2).contiguous()
        if self.do_normalize:
            x = 2.0 * x - 1.0
        return x

    def postprocess(self, image: torch.Tensor, output_type: str = "pil"):
        if output_type == "latent":
            return image
        x = (image.float() / 2 + 0.5).clamp(0, 1) if self.do_normalize else image.float()
        if output_type == "pt":
            return x
        arr = x.cpu().permute(0, 2, 3, 1).numpy()
        if output_type == "np":
            return arr
        from PIL import Image
        return [Image.fromarray((a * 255).round().astype("uint8")) for a in arr]
