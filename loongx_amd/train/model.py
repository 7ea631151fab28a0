"""Inference half of the reference's `src/train/model.py` on MI355X: CS3 encoders, DGF (DUAN) fusion and the
`OminiModel` attribute surface that `generate()` consumes (reference model.py:16-373, 430-462, 479-511, 731-779, 947-1035).

Everything runs fp32 and channel-major [B,C,L] (the shipped LoongX config is fp32 for these modules); the
arithmetic is in liblx_amd.so (cs3.hip / dgf.hip), torch only owns the buffers.  S4 layers are converted once at load
time to modal form (s4_params.py) and evaluated by the wavefront-scan kernel.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .. import ops
from . import s4_params


def _f32(t, device):
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


def _cplx(t: torch.Tensor) -> np.ndarray:
    """s4torch stores complex parameters as view_as_real pairs [..., 2]."""
    a = t.detach().cpu().double().numpy()
    return a[..., 0] + 1j * a[..., 1]


def resolve_s4_block_keys(sd: Dict[str, torch.Tensor], prefix: str, d_model: int) -> Dict[str, str]:
    """Names of one S4 block's tensors inside a checkpoint: -> {lam, p, q, B, Ct, D, log_step, lin_w, lin_b, ln_g, ln_b}.

    s4torch is neither pinned nor listed by the reference (src/train/model.py:14 imports it; no requirements file names it) and
    is absent offline, so the exact attribute names of its S4Block / S4Layer cannot be checked here. The names this repo's
    oracle uses (`s4._lambda_`, `s4._p`, `s4._q`, `s4._B`, `s4._Ct`, `s4.D`, `s4.log_step`, `linear.*`, `norm.*`) are tried first;
    otherwise the tensors are identified structurally: the S4 parameters by their name stem under any `s4`-like sub-module
    (with or without the leading underscore), the block's channel mixer as the only [d_model, d_model] matrix outside it (e.g.
    `pipeline.<i>.weight`), and the LayerNorm as the [d_model] weight/bias pair whose name contains `norm`. Ambiguity raises
    with the candidate keys listed."""
    keys = [k[len(prefix):] for k in sd if k.startswith(prefix)]
    if not keys:
        raise KeyError(f"no tensors under '{prefix}' in the state dict")
    out: Dict[str, str] = {}

    def pick(role, cands):
        cands = sorted(set(cands))
        if len(cands) != 1:
            raise KeyError(f"cannot identify the S4 block tensor '{role}' under '{prefix}': candidates {cands or 'none'}; "
                           f"keys present: {sorted(keys)}")
        out[role] = prefix + cands[0]

    def stem(k):          # last path component without leading / trailing underscores
        return k.split(".")[-1].strip("_").lower()

    for role, names in (("lam", ("lambda",)), ("p", ("p",)), ("q", ("q",)), ("B", ("b",)), ("Ct", ("ct", "c_tilde", "ctilde")),
                        ("D", ("d",)), ("log_step", ("log_step", "logstep"))):
        pick(role, [k for k in keys if stem(k) in names and "." in k and not k.split(".")[-2].startswith(("norm", "linear", "pipeline"))])
    s4_keys = {out[r][len(prefix):] for r in out}
    mats = [k for k in keys if k not in s4_keys and sd[prefix + k].dim() == 2 and tuple(sd[prefix + k].shape) == (d_model, d_model)]
    pick("lin_w", mats)
    lw = out["lin_w"][len(prefix):]
    pick("lin_b", [k for k in keys if k == lw[: -len("weight")] + "bias"])
    pick("ln_g", [k for k in keys if k.endswith("weight") and "norm" in k.lower() and sd[prefix + k].dim() == 1])
    pick("ln_b", [k for k in keys if k.endswith("bias") and "norm" in k.lower() and sd[prefix + k].dim() == 1])
    return out


class S4Model:
    """s4torch.S4Model(d_input, d_model, d_output, n_blocks, n, l_max) -- channel-major evaluation.
    y = enc(u); per block: y = LN(Linear(GELU(S4(y))) + y); out = dec(y)."""

    def __init__(self, sd: Dict[str, torch.Tensor], prefix: str, n_blocks: int, l_max: int, device, use_conv: bool = False):
        g = lambda k: sd[prefix + k]
        self.device, self.l_max, self.use_conv = device, l_max, use_conv
        self.enc_w, self.enc_b = _f32(g("encoder.weight"), device), _f32(g("encoder.bias"), device)
        self.dec_w, self.dec_b = _f32(g("decoder.weight"), device), _f32(g("decoder.bias"), device)
        self.d_in, self.d_model, self.d_out = self.enc_w.shape[1], self.enc_w.shape[0], self.dec_w.shape[0]
        self.blocks = []
        for i in range(n_blocks):
            kk = resolve_s4_block_keys(sd, f"{prefix}blocks.{i}.", self.d_model)
            n = sd[kk["p"]].shape[-2]
            lam = _cplx(sd[kk["lam"]]).reshape(-1, n)[0]                  # stored [1, n, 2] (or with more leading singleton dims)
            p, q = _cplx(sd[kk["p"]]).reshape(n), _cplx(sd[kk["q"]]).reshape(n)
            B, Ct = _cplx(sd[kk["B"]]).reshape(self.d_model, n), _cplx(sd[kk["Ct"]]).reshape(self.d_model, n)
            step = np.exp(sd[kk["log_step"]].detach().cpu().double().numpy().reshape(-1))
            lam_bar, w = s4_params.modal_form(lam, p, q, B, Ct, step, l_max)
            blk = dict(
                lam=torch.from_numpy(np.stack([lam_bar.real, lam_bar.imag], -1)).to(device).contiguous(),
                w=torch.from_numpy(np.stack([w.real, w.imag], -1)).to(device).contiguous(),
                D=_f32(sd[kk["D"]].reshape(-1), device),
                lin_w=_f32(sd[kk["lin_w"]], device), lin_b=_f32(sd[kk["lin_b"]], device),
                ln_g=_f32(sd[kk["ln_g"]], device), ln_b=_f32(sd[kk["ln_b"]], device))
            if use_conv:
                blk["K"] = torch.from_numpy(s4_params.kernel_from_modes(lam_bar, w, l_max)).float().to(device).contiguous()
            self.blocks.append(blk)

    def forward(self, u: torch.Tensor) -> torch.Tensor:
        """u fp32 [B, d_in, L] -> [B, d_out, L]."""
        B, _, L = u.shape
        if L != self.l_max:
            raise ValueError(f"S4 layer was built for l_max={self.l_max}, got L={L}")
        dev = u.device
        y = torch.empty(B, self.d_model, L, device=dev)
        ops.chanmix(u, self.enc_w, self.enc_b, None, None, None, y)
        z = torch.empty_like(y)
        for blk in self.blocks:
            if self.use_conv:
                ops.s4_conv(y, blk["K"], blk["D"], z)
            else:
                ops.s4_scan(y, blk["lam"], blk["w"], blk["D"], z)
            y2 = torch.empty_like(y)
            ops.chanmix(z, blk["lin_w"], blk["lin_b"], y, blk["ln_g"], blk["ln_b"], y2, act=1)
            y = y2
        out = torch.empty(B, self.d_out, L, device=dev)
        ops.chanmix(y, self.dec_w, self.dec_b, None, None, None, out)
        return out

    __call__ = forward


class _Head:
    """Flatten-Linear-LN-ReLU-Linear-LN-ReLU[-Unflatten(512,8)-Linear(8,E)] (reference model.py:60-72 etc.)."""

    def __init__(self, sd, prefix, device, expand: bool):
        g = lambda k: _f32(sd[prefix + k], device)
        self.w1, self.b1, self.g1, self.be1 = g("1.weight"), g("1.bias"), g("2.weight"), g("2.bias")
        self.w2, self.b2, self.g2, self.be2 = g("5.weight"), g("5.bias"), g("6.weight"), g("6.bias")
        self.w3, self.b3 = (g("10.weight"), g("10.bias")) if expand else (None, None)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, K = x.shape
        dev = x.device
        h1 = torch.empty(B, self.w1.shape[0], device=dev)
        ops.linear_f32(x, self.w1, self.b1, h1, M=B, N=self.w1.shape[0], K=K, ldx=x.stride(0), ldy=h1.stride(0))
        ops.layernorm_relu(h1, self.g1, self.be1)
        h2 = torch.empty(B, self.w2.shape[0], device=dev)
        ops.linear_f32(h1, self.w2, self.b2, h2, M=B, N=self.w2.shape[0], K=h1.shape[1], ldx=h1.stride(0), ldy=h2.stride(0))
        ops.layernorm_relu(h2, self.g2, self.be2)
        if self.w3 is None:
            return h2
        E = self.w3.shape[0]
        out = torch.empty(B, 512, E, device=dev)
        ops.linear_f32(h2.view(B * 512, 8), self.w3, self.b3, out, M=B * 512, N=E, K=8, ldx=8, ldy=E)
        return out


class EEGEncoder:
    """[B,4,4096] -> [B,512,4096] (reference model.py:16-134)."""
    fixed_length = 4096
    fpp_sizes = [128, 256, 512, 1024, 2048]

    def __init__(self, sd, prefix, device, use_conv=False):
        self.s41 = S4Model(sd, prefix + "s41.", 2, 4096, device, use_conv)
        self.s42 = S4Model(sd, prefix + "s42.", 2, 4096, device, use_conv)
        self.head = _Head(sd, prefix + "projection.", device, expand=True)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() != 3 or tuple(x.shape[1:]) != (4, 4096):
            raise ValueError(f"EEGEncoder expects [B,4,4096], got {tuple(x.shape)}")
        x = x.float().contiguous()
        B = x.shape[0]
        comb = torch.empty(B, 4, 4096, device=x.device)
        z1p = torch.empty(B, 64, 4, device=x.device)
        ops.pyramid_pool(self.s41(x), z1p, [4])
        comb[:, :, :64] = z1p.permute(0, 2, 1)                       # layout move only
        ops.pyramid_pool(x, comb, self.fpp_sizes, y_col0=64)
        ops.pyramid_pool(self.s42(x), comb, [64], y_col0=64 + sum(self.fpp_sizes))
        return self.head.forward(comb.view(B, 4 * 4096))

    __call__ = forward


class _FlatEncoder:
    def __init__(self, sd, prefix, device, ch, L, pool, fpp_sizes, expand, use_conv=False):
        self.ch, self.L, self.pool, self.fpp_sizes = ch, L, pool, fpp_sizes
        self.s4 = S4Model(sd, prefix + "s4.", 2, L, device, use_conv)
        self.head = _Head(sd, prefix + "projection.", device, expand)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() != 3 or tuple(x.shape[1:]) != (self.ch, self.L):
            raise ValueError(f"{type(self).__name__} expects [B,{self.ch},{self.L}], got {tuple(x.shape)}")
        x = x.float().contiguous()
        B = x.shape[0]
        nz, nf = self.ch * self.pool, self.ch * sum(self.fpp_sizes)
        z = torch.empty(B, self.ch, self.pool, device=x.device)
        ops.pyramid_pool(self.s4(x), z, [self.pool])
        f = torch.empty(B, self.ch, sum(self.fpp_sizes), device=x.device)
        ops.pyramid_pool(x, f, self.fpp_sizes)
        comb = torch.cat([z.view(B, nz), f.view(B, nf)], dim=1)          # layout move only
        return self.head.forward(comb)

    __call__ = forward


class PPGEncoder(_FlatEncoder):
    """[B,4,256] -> [B,512,4096] (reference model.py:137-205)."""
    fixed_length = 256

    def __init__(self, sd, prefix, device, use_conv=False):
        super().__init__(sd, prefix, device, 4, 256, 16, [64, 128, 256], True, use_conv)


class FNIRSEncoder(_FlatEncoder):
    """[B,6,512] -> [B,768] (reference model.py:208-274)."""
    fixed_length = 512

    def __init__(self, sd, prefix, device, use_conv=False):
        super().__init__(sd, prefix, device, 6, 512, 32, [128, 256, 448], False, use_conv)


class MotionEncoder(_FlatEncoder):
    """[B,6,128] -> [B,768] (reference model.py:277-343)."""
    fixed_length = 128

    def __init__(self, sd, prefix, device, use_conv=False):
        super().__init__(sd, prefix, device, 6, 128, 6, [32, 64, 124], False, use_conv)


class FeaturePyramidPooling:
    """reference model.py:345-373."""

    def __init__(self, output_sizes: Sequence[int] = (256, 512, 1024)):
        self.output_sizes = list(output_sizes)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = x.float().contiguous()
        y = torch.empty(*x.shape[:2], sum(self.output_sizes), device=x.device)
        ops.pyramid_pool(x, y, self.output_sizes)
        return y

    __call__ = forward


class DUAN:
    """Dynamic Gated Fusion core (reference model.py:947-1035): x content, c condition, both [B,C,L]."""

    def __init__(self, sd, prefix, device, keep_ratio: float = 0.7, eps: float = 1e-3):
        self.p = {}
        for k in ("gate.0", "gate.2", "mlp.0", "mlp.2"):
            w = sd[f"{prefix}{k}.weight"]
            self.p[k + ".weight"] = _f32(w.reshape(w.shape[0], -1), device)
            self.p[k + ".bias"] = _f32(sd[f"{prefix}{k}.bias"], device)
        self.channels = self.p["gate.2.weight"].shape[0]
        self.keep_ratio, self.eps = keep_ratio, eps

    def forward(self, x16: torch.Tensor, c16: torch.Tensor, keep_ratio: Optional[float] = None) -> torch.Tensor:
        if x16.shape != c16.shape:
            raise AssertionError("x, c must have identical shape [B,C,L]")
        x, c = x16.float().contiguous(), c16.float().contiguous()
        if x.shape[1] != self.channels:
            raise ValueError(f"DUAN built for {self.channels} channels, got {x.shape[1]}")
        kr = self.keep_ratio if keep_ratio is None else keep_ratio
        y = torch.empty_like(x)
        ops.duan_fwd(x, c, self.p, y, keep_k=max(1, int(self.channels * kr)), eps=self.eps)
        return y.to(x16.dtype)

    __call__ = forward


class _Linear:
    def __init__(self, sd, prefix, device):
        self.w, self.b = _f32(sd[prefix + "weight"], device), _f32(sd[prefix + "bias"], device)


class CS3DGF:
    """The brain-signal side of OminiModel: encoders, fusion linears, DUAN instances, and the fusion rules."""

    def __init__(self, sd: Dict[str, torch.Tensor], device="cuda", use_conv: bool = False):
        self.device = torch.device(device)
        self.eeg_projection = EEGEncoder(sd, "eeg_projection.", device, use_conv)
        self.ppg_projection = PPGEncoder(sd, "ppg_projection.", device, use_conv)
        self.fnirs_projection = FNIRSEncoder(sd, "fnirs_projection.", device, use_conv)
        self.motion_projection = MotionEncoder(sd, "motion_projection.", device, use_conv)
        self.fusion1, self.fusion2 = _Linear(sd, "fusion1.0.", device), _Linear(sd, "fusion2.0.", device)
        self.fusion3, self.fusion4 = _Linear(sd, "fusion3.0.", device), _Linear(sd, "fusion4.0.", device)
        self.duan_norm1, self.duan_norm2 = DUAN(sd, "duan_norm1.", device), DUAN(sd, "duan_norm2.", device)
        self.duan_norm_prompt, self.duan_norm_pooled = DUAN(sd, "duan_norm_prompt.", device), DUAN(sd, "duan_norm_pooled.", device)
        self.eeg_fixed_length, self.fnirs_fixed_length, self.ppg_fixed_length, self.motion_fixed_length = 4096, 512, 256, 128

    def spatial_pyramid_pooling(self, x: torch.Tensor, output_size: int, adaptive: bool = False) -> torch.Tensor:
        """reference model.py:479-511: zero-pad / truncate to a fixed length (adaptive pooling only on request)."""
        b, c, l = x.shape
        if l == output_size:
            return x
        if adaptive:
            y = torch.empty(b, c, output_size, device=x.device)
            ops.pyramid_pool(x.float().contiguous(), y, [output_size])
            return y.to(x.dtype)
        if l < output_size:
            return torch.cat([x, x.new_zeros(b, c, output_size - l)], dim=2)
        return x[:, :, :output_size]

    def fuse_eeg(self, eeg_f: torch.Tensor, ppg_f: torch.Tensor) -> torch.Tensor:
        """reference model.py:731-755: f = DUAN1(x=ppg, c=eeg); Linear(1024->512) over channels of cat([eeg, f])."""
        f = self.duan_norm1(ppg_f, eeg_f)
        B, Cc, L = eeg_f.shape
        out = torch.empty(B, self.fusion1.w.shape[0], L, device=eeg_f.device)
        w = self.fusion1.w
        # Linear(1024 -> 512) over channels at every position: out = W[:, :C] eeg + b, then += W[:, C:] f -- channel-major fp32
        # GEMMs on the f32 MFMA, the whole batch per launch (no torch.cat of [eeg, f], no transposes)
        ops.chan_gemm_f32(eeg_f.contiguous(), w, self.fusion1.b, out, N=w.shape[0], K=Cc)
        ops.chan_gemm_f32(f.contiguous(), w[:, Cc:], None, out, N=w.shape[0], K=Cc, epilogue=1, ldw=w.stride(0))
        return out

    def fuse_fnirs(self, fnirs_f: torch.Tensor, motion_f: torch.Tensor) -> torch.Tensor:
        """reference model.py:757-779."""
        f = self.duan_norm2(fnirs_f.unsqueeze(1), motion_f.unsqueeze(1)).squeeze(1).contiguous()
        B, E = fnirs_f.shape
        w = self.fusion2.w
        out = torch.empty(B, w.shape[0], device=fnirs_f.device)
        a = fnirs_f.float().contiguous()
        ops.linear_f32(a, w, self.fusion2.b, out, M=B, N=w.shape[0], K=E, ldx=E, ldy=out.stride(0))
        ops.linear_f32(f, w[:, E:], None, out, M=B, N=w.shape[0], K=E, ldx=E, ldy=out.stride(0), accumulate=True, ldw=w.stride(0))
        return out


# ---------------------------------------------------------------------------------------------------------------------
def synthetic_cs3_state_dict(seed: int = 0) -> Dict[str, torch.Tensor]:
    """Random CS3/DGF weights in the reference's module naming (OminiModel attributes, nn.Sequential indices)."""
    rng = np.random.default_rng(seed)
    sd: Dict[str, torch.Tensor] = {}
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()

    def lin(prefix, o, i, std=None):
        s = std if std is not None else 1.0 / np.sqrt(i)
        sd[prefix + "weight"] = t(rng.standard_normal((o, i)) * s)
        sd[prefix + "bias"] = t(rng.standard_normal(o) * 0.01)

    def ln(prefix, d):
        sd[prefix + "weight"], sd[prefix + "bias"] = torch.ones(d), torch.zeros(d)

    def s4model(prefix, d_in, d_model, d_out, n):
        lin(prefix + "encoder.", d_model, d_in)
        lin(prefix + "decoder.", d_out, d_model)
        for b in range(2):
            p = f"{prefix}blocks.{b}."
            lay = s4_params.init_s4_layer(d_model, n, rng)
            vr = lambda z: t(np.stack([z.real, z.imag], -1))
            sd[p + "s4._lambda_"], sd[p + "s4._p"], sd[p + "s4._q"] = vr(lay["lam"][None]), vr(lay["p"]), vr(lay["q"])
            sd[p + "s4._B"], sd[p + "s4._Ct"] = vr(lay["B"]), vr(lay["Ct"])
            sd[p + "s4.D"], sd[p + "s4.log_step"] = t(lay["D"]).reshape(1, 1, -1), t(lay["log_step"])
            lin(p + "linear.", d_model, d_model)
            ln(p + "norm.", d_model)

    def head(prefix, d_in, d_hid, d_out, expand):
        lin(prefix + "1.", d_hid, d_in); ln(prefix + "2.", d_hid)
        lin(prefix + "5.", d_out, d_hid); ln(prefix + "6.", d_out)
        if expand:
            lin(prefix + "10.", 4096, 8)

    s4model("eeg_projection.s41.", 4, 64, 64, 64)
    s4model("eeg_projection.s42.", 4, 4, 4, 4)
    head("eeg_projection.projection.", 16384, 2048, 4096, True)
    s4model("ppg_projection.s4.", 4, 4, 4, 4); head("ppg_projection.projection.", 1856, 1024, 4096, True)
    s4model("fnirs_projection.s4.", 6, 6, 6, 6); head("fnirs_projection.projection.", 5184, 1024, 768, False)
    s4model("motion_projection.s4.", 6, 6, 6, 6); head("motion_projection.projection.", 1356, 512, 768, False)
    for name, o, i in (("fusion1", 512, 1024), ("fusion2", 768, 1536), ("fusion3", 512, 1024), ("fusion4", 768, 1536)):
        lin(f"{name}.0.", o, i)
    for name, C in (("duan_norm1", 512), ("duan_norm2", 1), ("duan_norm_prompt", 512), ("duan_norm_pooled", 1)):
        for sub, o, i in (("gate.0", 128, C), ("gate.2", C, 128), ("mlp.0", 128, C), ("mlp.2", 2 * C, 128)):
            sd[f"{name}.{sub}.weight"] = t(rng.standard_normal((o, i, 1)) / np.sqrt(i))
            sd[f"{name}.{sub}.bias"] = t(rng.standard_normal(o) * 0.01)
    return sd


class OminiModel(CS3DGF):
    """Inference surface of the reference's `OminiModel` (src/train/model.py:376-511, 731-779) that `generate()` and
    `inference.py` consume: `.flux_pipe`, `.transformer`, `.model_config`, `.device`, the four `*_projection` encoders,
    `fuse_eeg` / `fuse_fnirs`, the DUAN instances, `spatial_pyramid_pooling`, `load_lora`, `load_state_dict`, `.to`, `.eval`.
    Training (LoRA init, optimizers, `training_step`) is out of scope.

    Same constructor keywords as the reference (model.py:377-389):

        OminiModel(flux_pipe_id=..., lora_path=None, lora_config=None, device="cuda", dtype=torch.bfloat16, model_config={},
                   optimizer_config=None, gradient_checkpointing=False, use_brain_condition=True, fuse_flag=True)

    `flux_pipe_id`: a local diffusers-format FLUX.1 directory (there is no hub access on the box; `transformer/`, and when
    present `vae/`, `text_encoder*/`, `tokenizer*/`, are loaded by `LxFluxPipeline.from_pretrained`), or None / "synthetic":
    the transformer and the brain-side modules are then created by `load_state_dict` from a full LoongX checkpoint
    (inference.py:46-53: keys `transformer.<diffusers names, optionally PEFT-wrapped>` + `eeg_projection.*`, `fusion1.*`,
    `duan_norm1.*` ...). `dtype=torch.float32` (the reference's shipped config, train/config/seed_512.yaml:2) selects the
    engine's precise mode; bfloat16 the bf16 MFMA mode. `lora_config` supplies the adapter scale alpha / r.
    `OminiModel.from_pipe(pipe, cs3_state_dict, model_config, device)` wraps an existing `LxFluxPipeline` and a CS3 / DGF state
    dict (the form the tests and bench.py use).

    Brain-side modules: the reference's constructor creates them randomly initialised (model.py:430-462) and a '*lora*' checkpoint
    (inference.py:43-44 -> load_lora) never fills them. Here they are built by `load_state_dict` / `from_pipe`; when neither has
    run, the first access builds them from a seeded random state dict with a warning (the reference-equivalent state), instead of
    failing with an AttributeError in the middle of generate()."""

    def __init__(self, flux_pipe_id=None, lora_path=None, lora_config: Optional[dict] = None, device="cuda",
                 dtype: torch.dtype = torch.bfloat16, model_config: Optional[dict] = None, optimizer_config: Optional[dict] = None,
                 gradient_checkpointing: bool = False, use_brain_condition: bool = True, fuse_flag: bool = True, *,
                 flux_config=None):
        if hasattr(flux_pipe_id, "transformer") or isinstance(lora_path, dict):
            raise TypeError("OminiModel(flux_pipe_id=<directory>, lora_path=<directory>, ...) takes the reference's arguments; to wrap an existing "
                            "pipeline object and a CS3 / DGF state dict use OminiModel.from_pipe(pipe, cs3_state_dict, model_config, device)")
        self.device = torch.device(device if not (device == "cuda" and not torch.cuda.is_available()) else "cpu")
        self._dtype = dtype
        self.precise = dtype == torch.float32
        self.model_config = dict(model_config or {})
        self.optimizer_config = optimizer_config
        self.lora_config = dict(lora_config or {})
        self.lora_scale = float(self.lora_config.get("lora_alpha", 4)) / float(self.lora_config.get("r", 4)) if self.lora_config else 1.0
        self.use_brain_condition, self.fuse_flag = use_brain_condition, fuse_flag
        self.eeg_fixed_length, self.fnirs_fixed_length, self.ppg_fixed_length, self.motion_fixed_length = 4096, 512, 256, 128
        self.flux_config = flux_config
        self.flux_pipe = self.transformer = None
        self._brain_ready = False
        self._brain_seed = 0
        if flux_pipe_id not in (None, "", "synthetic"):
            from ..flux.pipeline import LxFluxPipeline
            self._set_pipe(LxFluxPipeline.from_pretrained(flux_pipe_id, device=self.device, dtype=dtype, flux_config=flux_config,
                                                          lora_scale=self.lora_scale))
        if isinstance(lora_path, str) and lora_path:
            self.load_lora(lora_path)

    _BRAIN_ATTRS = frozenset(("eeg_projection", "ppg_projection", "fnirs_projection", "motion_projection", "fusion1", "fusion2", "fusion3",
                              "fusion4", "duan_norm1", "duan_norm2", "duan_norm_prompt", "duan_norm_pooled"))

    def __getattr__(self, name):
        # only reached when normal lookup fails: the brain-side modules of a model that never got a state dict for them
        if name in OminiModel._BRAIN_ATTRS and not self.__dict__.get("_brain_ready", True):
            import warnings
            warnings.warn("OminiModel: the CS3 encoders / DGF modules were never loaded (no full state dict; load_lora fills only the "
                          "transformer adapters) -- building them randomly initialised, which is the state the reference's constructor "
                          "leaves them in (src/train/model.py:430-462)", RuntimeWarning, stacklevel=2)
            self._build_brain(synthetic_cs3_state_dict(self.__dict__.get("_brain_seed", 0)))
            return self.__dict__[name]
        raise AttributeError(f"{type(self).__name__!s} object has no attribute {name!r}")

    @classmethod
    def from_pipe(cls, pipe, cs3_state_dict: Optional[Dict[str, torch.Tensor]] = None, model_config: Optional[dict] = None, device="cuda",
                  dtype: torch.dtype = torch.bfloat16, lora_config: Optional[dict] = None):
        """Wrap an existing pipeline object (anything with `.transformer`; None for a brain-side-only model) and a CS3 / DGF state
        dict in the reference's naming (`eeg_projection.*`, `fusion1.*`, `duan_norm1.*` ...)."""
        m = cls(None, lora_config=lora_config, device=device, dtype=dtype, model_config=model_config)
        if pipe is not None:
            m._set_pipe(pipe)
        if cs3_state_dict is not None:
            m._build_brain(cs3_state_dict)
        return m

    # ---- construction helpers ---------------------------------------------------------------------------------
    def _set_pipe(self, pipe) -> None:
        self.flux_pipe = pipe
        self.transformer = pipe.transformer
        if self.precise and hasattr(self.transformer, "engine"):
            self.transformer.engine.precise_default = True
        if self._dtype == torch.float16 and hasattr(self.transformer, "engine"):      # dtype=torch.float16: fp16 GEMM operand images
            self.transformer.engine.operands_default = "fp16"

    def _build_brain(self, sd: Dict[str, torch.Tensor]) -> None:
        CS3DGF.__init__(self, sd, self.device)
        self._brain_ready = True

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        """A full LoongX checkpoint in the reference's naming (what `torch.load(ckpt)["state_dict"]` holds, inference.py:46-53):
        `transformer.*` (diffusers FluxTransformer2DModel names, PEFT-wrapped where LoRA is attached) rebuilds the packed DiT
        weights, everything else the CS3 encoders / DGF modules. With strict=True keys that belong to neither raise."""
        from ..flux.pipeline import LxFluxPipeline
        from ..flux.transformer import LxFluxTransformer
        from ..flux.weights import FluxConfig
        sd = state_dict
        brain_prefixes = ("eeg_projection.", "ppg_projection.", "fnirs_projection.", "motion_projection.", "fusion1.", "fusion2.",
                          "fusion3.", "fusion4.", "duan_norm1.", "duan_norm2.", "duan_norm_prompt.", "duan_norm_pooled.")
        if strict:                                           # before anything is rebuilt
            unknown = [k for k in sd if not k.startswith(("transformer.",) + brain_prefixes)]
            if unknown:
                raise KeyError(f"load_state_dict: unexpected keys {unknown[:5]}{' ...' if len(unknown) > 5 else ''}")
        tkeys = [k for k in sd if k.startswith("transformer.")]
        if tkeys:
            cfg = self.flux_config or FluxConfig.from_state_dict(sd, "transformer.")
            tr = LxFluxTransformer.from_state_dict(sd, cfg, self.device, self.lora_scale, prefix="transformer.", precise=self.precise,
                                                   operands="fp16" if self._dtype == torch.float16 else "bf16")
            if self.flux_pipe is None:
                self._set_pipe(LxFluxPipeline(tr))
            else:                                            # keep the pipeline's VAE / text encoders, swap the transformer
                self.flux_pipe.transformer = tr
                self._set_pipe(self.flux_pipe)
        elif self.flux_pipe is None:
            raise KeyError("load_state_dict: no 'transformer.*' keys and no pipeline was constructed (flux_pipe_id=None)")
        if any(k.startswith(brain_prefixes) for k in sd):
            self._build_brain(sd)
        elif strict and not self._brain_ready:
            raise KeyError("load_state_dict: the checkpoint holds none of the brain-side modules (eeg_projection.*, fusion1.*, duan_norm1.* ...)")
        return self

    def state_dict_keys_expected(self):
        """(documentation helper) the top-level prefixes load_state_dict understands."""
        return ("transformer.", "eeg_projection.", "ppg_projection.", "fnirs_projection.", "motion_projection.", "fusion1..4.",
                "duan_norm1.", "duan_norm2.", "duan_norm_prompt.", "duan_norm_pooled.")

    @classmethod
    def from_state_dict(cls, state_dict: Dict[str, torch.Tensor], flux_config=None, model_config=None, device="cuda",
                        lora_scale: float = 1.0, dtype: torch.dtype = torch.bfloat16):
        """state_dict in the reference's checkpoint naming: `transformer.<diffusers names>` + `eeg_projection.*` ... """
        m = cls(None, device=device, dtype=dtype, model_config=model_config, flux_config=flux_config)
        m.lora_scale = lora_scale
        return m.load_state_dict(state_dict)

    @classmethod
    def synthetic(cls, flux_config=None, model_config=None, device="cuda", seed: int = 0, dtype: torch.dtype = torch.bfloat16):
        from ..flux.pipeline import LxFluxPipeline
        from ..flux.transformer import LxFluxTransformer
        tr = LxFluxTransformer.synthetic(flux_config, device, seed, precise=dtype == torch.float32,
                                         operands="fp16" if dtype == torch.float16 else "bf16")
        return cls.from_pipe(LxFluxPipeline(tr), synthetic_cs3_state_dict(seed), model_config, device, dtype=dtype)

    def load_lora(self, checkpoint_path: str):
        """model.py:463-477: load LoRA weights from a checkpoint directory into the pipeline's transformer."""
        self.flux_pipe.load_lora_weights(checkpoint_path, lora_scale=self.lora_scale)
        return self

    @property
    def dtype(self):
        return self._dtype

    def eval(self):
        return self

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError("training is outside the MI355X denoise hot path (SURVEY 2.1)")
        return self

    def to(self, *a, **k):
        """The weights live where the constructor put them (one MI355X per process); `.to("cuda")` / `.to(dtype)` are accepted
        for call-site compatibility (inference.py:55-56) and change nothing."""
        return self
