"""Mirror of the reference's src/flux/block.py on the MI355X DiT engine.

Same function names, argument meaning, return conventions and error behaviour as
  attn_forward (block.py:7-176), block_forward (:179-278), single_block_forward (:281-339);
the `attn` / `self` handles are `LxAttention` / `LxBlock` objects that index into a `DiTEngine` instead of
diffusers modules.  Tensors go in and come out as ordinary [B, L, D] torch tensors; the arithmetic runs in
liblx_amd.so.  These are the per-block entry points (tests, drop-in use); `tranformer_forward` drives the engine
directly and never round-trips activations through torch.
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import torch


class LxAttention:
    def __init__(self, engine, kind: str, idx: int):
        self.engine, self.kind, self.idx = engine, kind, idx
        self.heads = engine.cfg.num_attention_heads

    def _c_factor(self) -> Optional[float]:
        cf = getattr(self, "c_factor", None)
        return None if cf is None else float(torch.as_tensor(cf).flatten()[0])


class LxBlock:
    def __init__(self, engine, kind: str, idx: int):
        self.engine, self.kind, self.idx = engine, kind, idx
        self.attn = LxAttention(engine, kind, idx)


def _shape(hidden_states, encoder_hidden_states, condition_latents, single: bool):
    B = hidden_states.shape[0]
    C = 0 if condition_latents is None else condition_latents.shape[1]
    if single:
        return B, None, hidden_states.shape[1], C      # text length is resolved from the rope split below
    return B, encoder_hidden_states.shape[1], hidden_states.shape[1], C


def _configure(eng, B, T, N, C, model_config, c_factor, image_rotary_emb, cond_rotary_emb):
    eng.configure(B, T, N, C, model_config, c_factor, image_rotary_emb, cond_rotary_emb if C else None)


def attn_forward(attn: LxAttention, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor = None,
                 condition_latents: torch.Tensor = None, attention_mask: Optional[torch.Tensor] = None,
                 image_rotary_emb=None, cond_rotary_emb=None, model_config: Optional[Dict[str, Any]] = {}):
    """Inputs are the already-normalised streams. Double-block handles return (hidden, encoder[, condition]) after
    to_out / to_add_out; single-block handles return hidden or (hidden, condition) straight from attention.
    For single-block handles `hidden_states` is the [text; image] concatenation and `attn.text_len` gives the split
    (the reference needs no split because its rows stay concatenated)."""
    if attention_mask is not None:
        raise NotImplementedError("explicit attention_mask tensors are not supported; masks come from model_config / c_factor")
    eng = attn.engine
    D = eng.cfg.inner_dim
    dt = hidden_states.dtype
    if attn.kind == "double":
        if encoder_hidden_states is None:
            raise ValueError("double-stream attention needs encoder_hidden_states")
        B, T, N = hidden_states.shape[0], encoder_hidden_states.shape[1], hidden_states.shape[1]
        enc, hid = encoder_hidden_states, hidden_states
    else:
        T = int(getattr(attn, "text_len", 0))
        B, N = hidden_states.shape[0], hidden_states.shape[1] - T
        enc, hid = hidden_states[:, :T], hidden_states[:, T:]
    C = 0 if condition_latents is None else condition_latents.shape[1]
    _configure(eng, B, T, N, C, model_config, attn._c_factor(), image_rotary_emb, cond_rotary_emb)
    eng.load_streams(enc if T else None, hid, condition_latents, dst="XN")
    eng.attention_module(attn.kind, attn.idx, project_out=attn.kind == "double")
    if attn.kind == "double":
        h = eng.read_stream("img", N).to(dt)
        e = eng.read_stream("txt", T).to(dt)
        if C:
            return h, e, eng.read_stream("cond", C).to(dt)
        return h, e
    q = slice(2 * D, 3 * D)
    h = eng.read_stream("img", N, "Y", q)
    if T:
        h = torch.cat([eng.read_stream("txt", T, "Y", q), h], dim=1)
    h = h.to(dt)
    if C:
        return h, eng.read_stream("cond", C, "Y", q).to(dt)
    return h


def block_forward(self: LxBlock, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor,
                  condition_latents: torch.Tensor, temb: torch.Tensor, cond_temb: torch.Tensor, cond_rotary_emb=None,
                  image_rotary_emb=None, model_config: Optional[Dict[str, Any]] = {}):
    """-> (encoder_hidden_states, hidden_states, condition_latents | None)   (block.py:278)."""
    eng = self.engine
    use_cond = condition_latents is not None
    B, T, N = hidden_states.shape[0], encoder_hidden_states.shape[1], hidden_states.shape[1]
    C = condition_latents.shape[1] if use_cond else 0
    dt = hidden_states.dtype
    _configure(eng, B, T, N, C, model_config, self.attn._c_factor(), image_rotary_emb, cond_rotary_emb)
    eng.load_streams(encoder_hidden_states, hidden_states, condition_latents if use_cond else None)
    eng.block_mods("double", self.idx, temb, cond_temb if use_cond else None)
    eng.double_block(self.idx)
    out_e, out_h = eng.read_stream("txt", T).to(dt), eng.read_stream("img", N).to(dt)
    if out_e.dtype == torch.float16:
        out_e = out_e.clip(-65504, 65504)
    return out_e, out_h, (eng.read_stream("cond", C).to(dt) if use_cond else None)


def single_block_forward(self: LxBlock, hidden_states: torch.Tensor, temb: torch.Tensor, image_rotary_emb=None,
                         condition_latents: torch.Tensor = None, cond_temb: torch.Tensor = None, cond_rotary_emb=None,
                         model_config: Optional[Dict[str, Any]] = {}):
    """hidden_states is the [text; image] concatenation (transformer.py:182); -> hidden | (hidden, condition)."""
    eng = self.engine
    using_cond = condition_latents is not None
    T = int(getattr(self, "text_len", getattr(self.attn, "text_len", 0)))
    B, N = hidden_states.shape[0], hidden_states.shape[1] - T
    C = condition_latents.shape[1] if using_cond else 0
    dt = hidden_states.dtype
    _configure(eng, B, T, N, C, model_config, self.attn._c_factor(), image_rotary_emb, cond_rotary_emb)
    eng.load_streams(hidden_states[:, :T] if T else None, hidden_states[:, T:], condition_latents if using_cond else None)
    eng.block_mods("single", self.idx, temb, cond_temb if using_cond else None)
    eng.single_block(self.idx)
    h = eng.read_stream("img", N)
    if T:
        h = torch.cat([eng.read_stream("txt", T), h], dim=1)
    h = h.to(dt)
    if h.dtype == torch.float16:
        h = h.clip(-65504, 65504)
    return (h, eng.read_stream("cond", C).to(dt)) if using_cond else h
