"""Weight packing for the MI355X DiT engine.

Maps a diffusers-style `FluxTransformer2DModel` state_dict (optionally PEFT-LoRA wrapped: `<mod>.base_layer.weight`,
`<mod>.lora_A.<adapter>.weight`, `<mod>.lora_B.<adapter>.weight`; the checkpoint format the reference loads at
inference.py:46-53 / model.py:464-477) to the fused device layout the HIP kernels consume, or draws synthetic
weights of the same layout directly on the GPU (N(0, 0.02^2), zero bias, unit norms).

Fused layout (D = heads*128, r = LoRA rank; all GEMM weights bf16 [out, in], biases / norm weights / LoRA-up fp32):
  double block : qkv   [3D, D] rows = [to_k; to_v; to_q]     qkv_txt [3D, D] = [add_k; add_v; add_q]
                 out   [D, D]  (to_out.0)                     out_txt [D, D]  (to_add_out)
                 ff1   [4D, D] (ff.net.0.proj)                ff1_txt          (ff_context.net.0.proj)
                 ff2   [D, 4D] (ff.net.2)                     ff2_txt          (ff_context.net.2)
  single block : fused [7D, D] rows = [to_k; to_v; to_q; proj_mlp]      out [D, 5D] (proj_out, input = [attn | mlp])
  The [k | v | q] column order lets the attention output overwrite the q slot so that [attn | mlp] is one
  contiguous K=5D operand for proj_out (no torch.cat copy, block.py:326).
  modulation   : every AdaLN linear of the model stacked into ONE [n_mod, D] matrix (one weight-streaming launch
                 per step): per double block [norm1 (6D); norm1_context (6D)], per single block [norm (3D)],
                 then norm_out (2D: scale, shift).
"""
from __future__ import annotations

import re
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch

_GEMM_WEIGHTS = re.compile(r"^(d\d+\.(qkv|qkv_txt|out|out_txt|ff1|ff1_txt|ff2|ff2_txt)|s\d+\.(fused|out)|x_embedder|context_embedder)$")


def _maybe_tile(name: str, w: torch.Tensor) -> torch.Tensor:
    """GEMM-consumed weights are stored pre-tiled (ops.tile_weight: the kernel's LDS image, one contiguous 32 KiB block
    per K tile) so they stream from HBM with DRAM-page locality."""
    if not _GEMM_WEIGHTS.match(name):
        return w
    if w.shape[0] % 256 or w.shape[1] % 64:
        return w
    from ..ops import tile_weight
    return tile_weight(w)


@dataclass
class FluxConfig:
    num_layers: int = 19
    num_single_layers: int = 38
    num_attention_heads: int = 24
    attention_head_dim: int = 128
    in_channels: int = 64
    joint_attention_dim: int = 4096
    pooled_projection_dim: int = 768
    guidance_embeds: bool = True
    axes_dims_rope: tuple = (16, 56, 56)
    lora_r: int = 4

    @property
    def inner_dim(self) -> int:
        return self.num_attention_heads * self.attention_head_dim

    @classmethod
    def from_state_dict(cls, sd, prefix: str = "", axes_dims_rope=(16, 56, 56), attention_head_dim: int = 128) -> "FluxConfig":
        """Read the architecture off a diffusers FluxTransformer2DModel state dict (block counts, widths, guidance embedder)."""
        def w(name):
            for k in (f"{prefix}{name}.weight", f"{prefix}{name}.base_layer.weight"):
                if k in sd:
                    return sd[k]
            raise KeyError(f"'{prefix}{name}.weight' not in the state dict")
        pat_d, pat_s = re.compile(re.escape(prefix) + r"transformer_blocks\.(\d+)\."), re.compile(re.escape(prefix) + r"single_transformer_blocks\.(\d+)\.")
        nd = max((int(m.group(1)) for m in map(pat_d.match, sd) if m), default=-1) + 1
        ns = max((int(m.group(1)) for m in map(pat_s.match, sd) if m), default=-1) + 1
        xe = w("x_embedder")
        return cls(num_layers=nd, num_single_layers=ns, num_attention_heads=xe.shape[0] // attention_head_dim,
                   attention_head_dim=attention_head_dim, in_channels=xe.shape[1], joint_attention_dim=w("context_embedder").shape[1],
                   pooled_projection_dim=w("time_text_embed.text_embedder.linear_1").shape[1],
                   guidance_embeds=any(k.startswith(prefix + "time_text_embed.guidance_embedder.") for k in sd),
                   axes_dims_rope=tuple(axes_dims_rope))

    @property
    def n_mod(self) -> int:
        return (12 * self.num_layers + 3 * self.num_single_layers + 2) * self.inner_dim

    def mod_base_double(self, i: int) -> int:
        return 12 * self.inner_dim * i

    def mod_base_single(self, j: int) -> int:
        return (12 * self.num_layers + 3 * j) * self.inner_dim

    @property
    def mod_base_out(self) -> int:
        return (12 * self.num_layers + 3 * self.num_single_layers) * self.inner_dim


class Lora:
    """A_down [n_mod*r, K] bf16 (rows stacked per fused module), B_up [N, r] fp32 (already times alpha/r).
    down_lo: bf16(A - bf16(A)) when the model was packed for precise mode and A is not bf16-representable, else None."""
    __slots__ = ("down", "up", "down_lo")

    def __init__(self, down: torch.Tensor, up: torch.Tensor, down_lo: Optional[torch.Tensor] = None):
        self.down, self.up, self.down_lo = down, up, down_lo


def _hi_lo(w32: torch.Tensor):
    """fp32 -> (bf16 hi, bf16 lo or None): w = hi + lo to 16 mantissa bits; lo is None when w is bf16-representable."""
    hi = w32.to(torch.bfloat16)
    lo = (w32 - hi.float()).to(torch.bfloat16)
    return hi.contiguous(), (lo.contiguous() if bool(lo.any()) else None)


@dataclass
class PackedWeights:
    """t[name + ".w"]: the bf16 weight (GEMM weights pre-tiled). Packed with precise=True, weights that are NOT
    bf16-representable also get their rounding residual: GEMM weights as t[name + ".w2"] = [W_hi | W_lo] ([N, 2K], the operand
    of a k_segs = 3 GEMM), weight-streaming ones as t[name + ".w_lo"]. bf16-representable weights (FLUX.1 checkpoints, the
    synthetic weights) need neither: precise mode then costs 2 K-segments instead of 3."""
    cfg: FluxConfig
    t: Dict[str, torch.Tensor] = field(default_factory=dict)     # name -> tensor
    lora: Dict[str, Lora] = field(default_factory=dict)          # name -> Lora (absent => no adapter)
    precise_ready: bool = False                                   # every weight is bf16-exact or carries its residual
    lora_version: int = 0                                         # bumped by install_lora

    def nbytes(self) -> int:
        n = sum(v.numel() * v.element_size() for v in self.t.values())
        n += sum(l.down.numel() * 2 + l.up.numel() * 4 for l in self.lora.values())
        return n


def _check_lora_ranks(cfg: FluxConfig, ranks: Dict[str, int], group_sizes: Dict[str, int]) -> None:
    """One rank r for every adapter of a checkpoint, and r * (modules fused into one GEMM) <= 16: lx_lora_down produces at most
    16 columns per launch (the TL slabs of the engine are 16 columns wide) and the fused single-block GEMM evaluates four
    adapters (to_k, to_v, to_q, proj_mlp) from one slab. Sets cfg.lora_r, which sizes the engine's modulation scratch."""
    if not ranks:
        return
    rs = sorted(set(ranks.values()))
    if len(rs) != 1:
        bad = {k: v for k, v in ranks.items() if v != rs[0]}
        raise ValueError(f"LoRA adapters must share one rank; found ranks {rs} (e.g. {list(bad.items())[:3]})")
    r = rs[0]
    for name, n in group_sizes.items():
        if n * r > 16:
            raise ValueError(f"LoRA rank {r} is too large for the fused GEMM '{name}' ({n} adapters share one down-projection "
                             f"launch: needs {n} * r <= 16, i.e. r <= {16 // n})")
    cfg.lora_r = r


# --------------------------------------------------------------------------------------------------------------
def _sd_get(sd, name: str, what: str) -> Optional[torch.Tensor]:
    for k in (f"{name}.{what}", f"{name}.base_layer.{what}"):
        if k in sd:
            return sd[k]
    return None


def _sd_lora(sd, name: str):
    a = b = None
    pa, pb = re.compile(re.escape(name) + r"\.lora_A\.[^.]+\.weight$"), re.compile(re.escape(name) + r"\.lora_B\.[^.]+\.weight$")
    for k in sd:
        if pa.match(k):
            a = sd[k]
        elif pb.match(k):
            b = sd[k]
    return (a, b) if a is not None and b is not None else None


def pack_state_dict(sd: Dict[str, torch.Tensor], cfg: FluxConfig, device, lora_scale: float = 1.0,
                    prefix: str = "", precise: bool = False) -> PackedWeights:
    """sd: diffusers FluxTransformer2DModel names (optionally under `prefix`, e.g. 'transformer.').
    precise=True keeps the bf16 rounding residual of every weight that has one (see PackedWeights)."""
    if prefix:
        sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    pw = PackedWeights(cfg)
    pw.precise_ready = bool(precise)
    D = cfg.inner_dim
    ranks: Dict[str, int] = {}
    groups: Dict[str, int] = {}

    def W32(names: List[str]) -> torch.Tensor:
        ws = []
        for n in names:
            w = _sd_get(sd, n, "weight")
            if w is None:
                raise KeyError(f"missing weight for '{n}' in state_dict")
            ws.append(w.float())
        return torch.cat(ws, 0).to(device=device).contiguous()

    def Bv(names: List[str], sizes: List[int]) -> torch.Tensor:
        bs = []
        for n, s in zip(names, sizes):
            b = _sd_get(sd, n, "bias")
            bs.append(b.float() if b is not None else torch.zeros(s))
        return torch.cat(bs, 0).to(device=device, dtype=torch.float32).contiguous()

    def Lr(names: List[str]) -> Optional[Lora]:
        parts = [_sd_lora(sd, n) for n in names]
        if all(p is None for p in parts):
            return None
        if any(p is None for p in parts):
            raise ValueError(f"LoRA adapters must cover all of {names} or none")
        for n, p in zip(names, parts):
            ranks[n] = p[0].shape[0]
        groups[names[0]] = len(names)
        down, down_lo = _hi_lo(torch.cat([p[0].float() for p in parts], 0).to(device))
        up = torch.cat([p[1].float() * lora_scale for p in parts], 0).to(device=device, dtype=torch.float32).contiguous()
        return Lora(down, up, down_lo if precise else None)

    def put(name, w_names, out_sizes):
        hi, lo = _hi_lo(W32(w_names))
        pw.t[name + ".w"] = _maybe_tile(name, hi)
        if precise and lo is not None:
            if name.startswith("tte."):                    # weight-streaming linears: a second accumulate pass over the residual
                pw.t[name + ".w_lo"] = lo
            else:                                          # GEMM weights: [W_hi | W_lo], the operand of a k_segs = 3 launch
                pw.t[name + ".w2"] = _maybe_tile(name, torch.cat([hi, lo], 1).contiguous())
        pw.t[name + ".b"] = Bv(w_names, out_sizes)
        l = Lr(w_names)
        if l is not None:
            pw.lora[name] = l

    def vec(name, key):
        pw.t[name] = sd[key].float().to(device).contiguous()

    mod_w, mod_b, mod_lora = [], [], []
    for i in range(cfg.num_layers):
        p = f"transformer_blocks.{i}."
        put(f"d{i}.qkv", [p + "attn.to_k", p + "attn.to_v", p + "attn.to_q"], [D] * 3)
        put(f"d{i}.qkv_txt", [p + "attn.add_k_proj", p + "attn.add_v_proj", p + "attn.add_q_proj"], [D] * 3)
        put(f"d{i}.out", [p + "attn.to_out.0"], [D])
        put(f"d{i}.out_txt", [p + "attn.to_add_out"], [D])
        put(f"d{i}.ff1", [p + "ff.net.0.proj"], [4 * D])
        put(f"d{i}.ff1_txt", [p + "ff_context.net.0.proj"], [4 * D])
        put(f"d{i}.ff2", [p + "ff.net.2"], [D])
        put(f"d{i}.ff2_txt", [p + "ff_context.net.2"], [D])
        vec(f"d{i}.wq", p + "attn.norm_q.weight"); vec(f"d{i}.wk", p + "attn.norm_k.weight")
        vec(f"d{i}.wq_txt", p + "attn.norm_added_q.weight"); vec(f"d{i}.wk_txt", p + "attn.norm_added_k.weight")
        for n, sz in ((p + "norm1.linear", 6 * D), (p + "norm1_context.linear", 6 * D)):
            mod_w.append(_sd_get(sd, n, "weight").float()); mod_b.append(_sd_get(sd, n, "bias").float())
        mod_lora.append(_sd_lora(sd, p + "norm1.linear"))
    for j in range(cfg.num_single_layers):
        p = f"single_transformer_blocks.{j}."
        put(f"s{j}.fused", [p + "attn.to_k", p + "attn.to_v", p + "attn.to_q", p + "proj_mlp"], [D, D, D, 4 * D])
        put(f"s{j}.out", [p + "proj_out"], [D])
        vec(f"s{j}.wq", p + "attn.norm_q.weight"); vec(f"s{j}.wk", p + "attn.norm_k.weight")
        mod_w.append(_sd_get(sd, p + "norm.linear", "weight").float()); mod_b.append(_sd_get(sd, p + "norm.linear", "bias").float())
        mod_lora.append(_sd_lora(sd, p + "norm.linear"))
    mod_w.append(sd["norm_out.linear.weight"].float()); mod_b.append(sd["norm_out.linear.bias"].float())
    pw.t["mod.w"], mlo = _hi_lo(torch.cat(mod_w, 0).to(device))
    if precise and mlo is not None:
        pw.t["mod.w_lo"] = mlo
    pw.t["mod.b"] = torch.cat(mod_b, 0).to(device=device, dtype=torch.float32).contiguous()
    if any(m is not None for m in mod_lora):
        if any(m is None for m in mod_lora):
            raise ValueError("LoRA must cover every norm1.linear / norm.linear or none")
        for idx, m in enumerate(mod_lora):
            ranks[f"mod.{idx}"] = m[0].shape[0]
        pw.t["mod.lora_down"], dlo = _hi_lo(torch.cat([m[0].float() for m in mod_lora], 0).to(device))
        if precise and dlo is not None:
            pw.t["mod.lora_down_lo"] = dlo
        for idx, m in enumerate(mod_lora):
            pw.t[f"mod.lora_up.{idx}"] = (m[1].float() * lora_scale).to(device).contiguous()
    put("x_embedder", ["x_embedder"], [D])
    put("context_embedder", ["context_embedder"], [D])
    put("proj_out", ["proj_out"], [cfg.in_channels])
    emb = ["timestep_embedder", "text_embedder"] + (["guidance_embedder"] if cfg.guidance_embeds else [])
    for e in emb:
        for l in ("linear_1", "linear_2"):
            put(f"tte.{e}.{l}", [f"time_text_embed.{e}.{l}"], [D])
    _check_lora_ranks(cfg, ranks, groups)
    return pw


def lora_layout(cfg: FluxConfig):
    """(fused name, [diffusers module names]) of every GEMM that may carry a LoRA adapter, and the per-block modulation
    Linear names in packed order (mod.lora_down / mod.lora_up.<idx>) -- the same grouping pack_state_dict uses."""
    fused, mods = [], []
    for i in range(cfg.num_layers):
        p = f"transformer_blocks.{i}."
        fused += [(f"d{i}.qkv", [p + "attn.to_k", p + "attn.to_v", p + "attn.to_q"]),
                  (f"d{i}.qkv_txt", [p + "attn.add_k_proj", p + "attn.add_v_proj", p + "attn.add_q_proj"]),
                  (f"d{i}.out", [p + "attn.to_out.0"]), (f"d{i}.out_txt", [p + "attn.to_add_out"]),
                  (f"d{i}.ff1", [p + "ff.net.0.proj"]), (f"d{i}.ff1_txt", [p + "ff_context.net.0.proj"]),
                  (f"d{i}.ff2", [p + "ff.net.2"]), (f"d{i}.ff2_txt", [p + "ff_context.net.2"])]
        mods.append(p + "norm1.linear")
    for j in range(cfg.num_single_layers):
        p = f"single_transformer_blocks.{j}."
        fused += [(f"s{j}.fused", [p + "attn.to_k", p + "attn.to_v", p + "attn.to_q", p + "proj_mlp"]), (f"s{j}.out", [p + "proj_out"])]
        mods.append(p + "norm.linear")
    fused += [("x_embedder", ["x_embedder"]), ("context_embedder", ["context_embedder"]), ("proj_out", ["proj_out"])]
    return fused, mods


def install_lora(pw: PackedWeights, lora_sd: Dict[str, torch.Tensor], lora_scale: float = 1.0, prefix: str = "transformer.") -> int:
    """Install (replace) the LoRA adapters of an already packed model from a LoRA-only state dict -- the content of the
    `pytorch_lora_weights.safetensors` the reference loads through `flux_pipe.load_lora_weights` (model.py:463-477):
    keys `[transformer.]<module>.lora_A[.<adapter>].weight` / `lora_B...`, optional `<module>.alpha` (then the up matrix is
    scaled by alpha / r, as diffusers / peft do). Returns the number of adapted Linear modules. Adapters that the fused GEMMs
    evaluate together (q/k/v[/proj_mlp] of one block) must be given for all of them or none."""
    sd = {}
    for k, v in lora_sd.items():
        if prefix and k.startswith(prefix):
            k = k[len(prefix):]
        sd[re.sub(r"\.lora_([AB])\.[^.]+\.weight$", r".lora_\1.weight", k)] = v
    dev = pw.t["mod.w"].device
    used = set()

    def get(name):
        a, b = sd.get(name + ".lora_A.weight"), sd.get(name + ".lora_B.weight")
        if a is None or b is None:
            return None
        used.update((name + ".lora_A.weight", name + ".lora_B.weight"))
        sc = lora_scale
        if name + ".alpha" in sd:
            used.add(name + ".alpha")
            sc *= float(sd[name + ".alpha"]) / a.shape[0]
        return a.float(), b.float() * sc

    fused, mods = lora_layout(pw.cfg)
    n = 0
    new_lora = {}
    ranks: Dict[str, int] = {}
    groups: Dict[str, int] = {}
    for fname, names in fused:
        parts = [get(m) for m in names]
        if all(q is None for q in parts):
            continue
        if any(q is None for q in parts):
            raise ValueError(f"LoRA adapters must cover all of {names} or none")
        for m, q in zip(names, parts):
            ranks[m] = q[0].shape[0]
        groups[fname] = len(names)
        n += len(parts)
        # a model packed for precise mode keeps the adapters' bf16 rounding residuals too (as pack_state_dict does), so the
        # same checkpoint gives the same precise-mode numerics whichever way it was loaded
        down, down_lo = _hi_lo(torch.cat([q[0] for q in parts], 0).to(dev))
        new_lora[fname] = Lora(down, torch.cat([q[1] for q in parts], 0).to(device=dev, dtype=torch.float32).contiguous(),
                               down_lo if pw.precise_ready else None)
    mparts = [get(m) for m in mods]
    new_t = {}
    if any(q is not None for q in mparts):
        if any(q is None for q in mparts):
            raise ValueError("LoRA must cover every norm1.linear / norm.linear or none")
        n += len(mparts)
        for m, q in zip(mods, mparts):
            ranks[m] = q[0].shape[0]
        new_t["mod.lora_down"], dlo = _hi_lo(torch.cat([q[0] for q in mparts], 0).to(dev))
        if pw.precise_ready and dlo is not None:
            new_t["mod.lora_down_lo"] = dlo
        for idx, q in enumerate(mparts):
            new_t[f"mod.lora_up.{idx}"] = q[1].to(dev).contiguous()
    unknown = [k for k in sd if k not in used and ".lora_" in k]
    if unknown:
        raise KeyError(f"LoRA keys that match no module of this transformer: {unknown[:4]}{' ...' if len(unknown) > 4 else ''}")
    if n == 0:
        raise ValueError("no LoRA adapter found in the state dict")
    _check_lora_ranks(pw.cfg, ranks, groups)
    pw.lora.clear()
    pw.lora.update(new_lora)
    for k in [k for k in pw.t if k.startswith("mod.lora_")]:
        del pw.t[k]
    pw.t.update(new_t)
    pw.lora_version += 1                 # engines rebuild what they derived from the adapters (the fp16 images: DiTEngine._setup_f16)
    return n


@torch.no_grad()
def realistic_stats_(pw: "PackedWeights", seed: int = 0) -> Dict:
    """Give synthetic packed weights the statistics a trained checkpoint has and N(0, std^2) weights with unit norms and zero biases do
    not (bench.py's `realistic_stats` leg; the parity harness applies the same recipe to the oracle model, oracle/parity.py):
      (a) per-layer q / k RMSNorm weights with gains 0.5 ... 2.5 and 15 % channel jitter -- about half of the layers then exceed the
          bounded-score attention's limit (16.33 max|w_q| max|w_k| <= 100) and keep the max-tracking kernel: a MIXED plan in one step;
      (b) four residual-stream channels carrying a constant 60 ... 900 (x_embedder / context_embedder biases: "massive activations") and
          three MLP-hidden channels per block at 50 ... 200 (ff1 / proj_mlp biases);
      (c) every other bias N(0, 0.05^2)."""
    cfg = pw.cfg
    D = cfg.inner_dim
    g = torch.Generator().manual_seed(9000 + seed)
    dev = next(iter(pw.t.values())).device
    gains = []
    for li in range(cfg.num_layers + cfg.num_single_layers):
        gain = 0.5 + 2.0 * ((li * 7) % 10) / 9.0
        gains.append(gain)
        p = f"d{li}" if li < cfg.num_layers else f"s{li - cfg.num_layers}"
        for n in ("wq", "wk", "wq_txt", "wk_txt"):
            t = pw.t.get(f"{p}.{n}")
            if t is not None:
                t.copy_((gain * (1.0 + 0.15 * torch.randn(t.shape[0], generator=g))).to(dev))
    for name, t in pw.t.items():
        if name.endswith(".b") and t.dim() == 1:
            t.copy_((0.05 * torch.randn(t.shape[0], generator=g)).to(dev))
    res_ch, res_val = [17, 1031, 2049, 3001], [120.0, -300.0, 60.0, 900.0]
    for emb in ("x_embedder", "context_embedder"):
        for c, v in zip(res_ch, res_val):
            if c < D:
                pw.t[emb + ".b"][c] = v
    hid_val = [50.0, 200.0, -90.0]
    for li in range(cfg.num_layers + cfg.num_single_layers):
        if li < cfg.num_layers:
            targets = [(pw.t[f"d{li}.ff1.b"], 0), (pw.t[f"d{li}.ff1_txt.b"], 0)]
        else:
            targets = [(pw.t[f"s{li - cfg.num_layers}.fused.b"], 3 * D)]          # fused columns [k | v | q | mlp]
        for b, off in targets:
            n = b.shape[0] - off
            for k, v in enumerate(hid_val):
                b[off + (li * 131 + k * 4099 + 7) % n] = v
    if getattr(pw, "q_log2", None):
        from ..dist import refresh_q_log2
        refresh_q_log2(pw)
    pw.weights_version = getattr(pw, "weights_version", 0) + 1
    return {"norm_gain_range": [min(gains), max(gains)], "residual_outlier_channels": dict(zip(res_ch, res_val)), "mlp_hidden_outliers": hid_val, "bias_std": 0.05}


def synthetic_weights(cfg: FluxConfig, device, seed: int = 0, std: float = 0.02, lora: bool = True, fill: bool = True) -> PackedWeights:
    """Random weights of the fused layout generated on the GPU (full FLUX.1-dev scale = 23.8 GB bf16). Drawn in bf16, hence
    bf16-representable: precise mode needs no residuals for them. fill=False allocates the same tensors without drawing them
    (ranks that receive the weights by broadcast: dist.broadcast_packed_weights overwrites every byte)."""
    g = torch.Generator(device=device).manual_seed(seed)
    pw = PackedWeights(cfg)
    pw.precise_ready = True
    D, r = cfg.inner_dim, cfg.lora_r

    def rn(*shape, dtype=torch.bfloat16, s=std):
        out = torch.empty(*shape, dtype=dtype, device=device)
        if fill:
            out.normal_(0.0, s, generator=g)
        return out

    def put(name, n_out, n_in, n_lora_mod=0):
        pw.t[name + ".w"] = _maybe_tile(name, rn(n_out, n_in))
        pw.t[name + ".b"] = torch.zeros(n_out, dtype=torch.float32, device=device)
        if lora and n_lora_mod:
            pw.lora[name] = Lora(rn(n_lora_mod * r, n_in), rn(n_out, r, dtype=torch.float32))

    for i in range(cfg.num_layers):
        put(f"d{i}.qkv", 3 * D, D, 3); put(f"d{i}.qkv_txt", 3 * D, D)
        put(f"d{i}.out", D, D, 1); put(f"d{i}.out_txt", D, D)
        put(f"d{i}.ff1", 4 * D, D); put(f"d{i}.ff1_txt", 4 * D, D)
        put(f"d{i}.ff2", D, 4 * D, 1); put(f"d{i}.ff2_txt", D, 4 * D)
        for n in ("wq", "wk", "wq_txt", "wk_txt"):
            pw.t[f"d{i}.{n}"] = torch.ones(128, dtype=torch.float32, device=device)
    for j in range(cfg.num_single_layers):
        put(f"s{j}.fused", 7 * D, D, 4); put(f"s{j}.out", D, 5 * D, 1)
        pw.t[f"s{j}.wq"] = torch.ones(128, dtype=torch.float32, device=device)
        pw.t[f"s{j}.wk"] = torch.ones(128, dtype=torch.float32, device=device)
    pw.t["mod.w"] = rn(cfg.n_mod, D)
    pw.t["mod.b"] = torch.zeros(cfg.n_mod, dtype=torch.float32, device=device)
    if lora:
        nb = cfg.num_layers + cfg.num_single_layers
        pw.t["mod.lora_down"] = rn(nb * r, D)
        for idx in range(nb):
            pw.t[f"mod.lora_up.{idx}"] = rn(6 * D if idx < cfg.num_layers else 3 * D, r, dtype=torch.float32)
    put("x_embedder", D, cfg.in_channels, 1)
    put("context_embedder", D, cfg.joint_attention_dim)
    put("proj_out", cfg.in_channels, D)
    emb = [("timestep_embedder", 256), ("text_embedder", cfg.pooled_projection_dim)]
    if cfg.guidance_embeds:
        emb.append(("guidance_embedder", 256))
    for e, k in emb:
        put(f"tte.{e}.linear_1", D, k); put(f"tte.{e}.linear_2", D, D)
    return pw
