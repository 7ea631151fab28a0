"""Oracle: the reference's own three-stream DiT forward, restated on torch-CPU.

TEST INFRASTRUCTURE.  Restates /root/reference/src/flux/block.py (attn_forward :7-176,
block_forward :179-278, single_block_forward :281-339) and src/flux/transformer.py
(tranformer_forward :47-252) as straight-line tensor code.  Pinned by
tests/golden/flux_*.npz, which `oracle/make_goldens.py` produced by running the REAL
reference functions (imported from /root/reference in the build container) over the
`oracle/flux_modules.py` sub-modules.

Stream semantics restated here:
* three token streams: text ("enc"), image ("hid") and condition ("cond");
* the condition stream reuses the image-stream modules WITH LoRA active, the image
  stream runs them with LoRA scaled to 0 unless model_config["latent_lora"]
  (lora_controller.py:21-28);
* joint attention is over the concatenation [text, image, cond] (block.py:69-72,101-104);
* masks: union_cond_attn=False blocks cond<->rest both ways (block.py:106-114);
  independent_condition blocks cond queries from seeing the rest (:115-120);
  attn.c_factor adds log(c_factor) on the cond<->rest blocks (:121-128) and, being
  evaluated last, replaces any boolean mask.
"""
from __future__ import annotations

import math
from typing import Any, Dict, Optional, Tuple

import torch
import torch.nn.functional as F

from .flux_modules import LoraLinear, apply_rotary_emb


def _linear(mod, x, lora_on: bool):
    """Evaluate a (possibly LoRA-wrapped) Linear with the adapter on or off."""
    if isinstance(mod, LoraLinear):
        y = mod.base_layer(x)
        if lora_on:
            for a in mod.active_adapters:
                y = y + mod.lora_B[a](mod.lora_A[a](x)) * mod.scaling[a]
        return y
    return mod(x)


def _heads(x, h):
    b, l, d = x.shape
    return x.view(b, l, h, d // h).transpose(1, 2)


def attention_mask(n_q: int, n_cond: int, model_config: Dict[str, Any], c_factor, dtype, device=None):
    """block.py:106-128 -> additive/bool mask or None."""
    mask = None
    if not model_config.get("union_cond_attn", True):
        mask = torch.ones(n_q, n_q, dtype=torch.bool, device=device)
        mask[-n_cond:, :-n_cond] = False
        mask[:-n_cond, -n_cond:] = False
    elif model_config.get("independent_condition", False):
        mask = torch.ones(n_q, n_q, dtype=torch.bool, device=device)
        mask[-n_cond:, :-n_cond] = False
    if c_factor is not None:
        mask = torch.zeros(n_q, n_q, dtype=dtype, device=device)
        bias = math.log(float(c_factor))
        mask[-n_cond:, :-n_cond] = bias
        mask[:-n_cond, -n_cond:] = bias
    return mask


def joint_attention(attn, hid, enc, cond, rope_main, rope_cond, model_config) -> Tuple:
    """Returns per-stream attention outputs BEFORE the output projections:
    (hid_o, enc_o|None, cond_o|None), each [B, L, D]."""
    H = attn.heads
    latent_lora = bool(model_config.get("latent_lora", False))
    q = attn.norm_q(_heads(_linear(attn.to_q, hid, latent_lora), H))
    k = attn.norm_k(_heads(_linear(attn.to_k, hid, latent_lora), H))
    v = _heads(_linear(attn.to_v, hid, latent_lora), H)
    n_enc = 0
    if enc is not None:
        n_enc = enc.shape[1]
        eq = attn.norm_added_q(_heads(attn.add_q_proj(enc), H))
        ek = attn.norm_added_k(_heads(attn.add_k_proj(enc), H))
        ev = _heads(attn.add_v_proj(enc), H)
        q, k, v = torch.cat([eq, q], 2), torch.cat([ek, k], 2), torch.cat([ev, v], 2)
    if rope_main is not None:
        q, k = apply_rotary_emb(q, rope_main), apply_rotary_emb(k, rope_main)
    n_cond = 0
    if cond is not None:
        n_cond = cond.shape[1]
        cq = attn.norm_q(_heads(_linear(attn.to_q, cond, True), H))
        ck = attn.norm_k(_heads(_linear(attn.to_k, cond, True), H))
        cv = _heads(_linear(attn.to_v, cond, True), H)
        if rope_cond is not None:
            cq, ck = apply_rotary_emb(cq, rope_cond), apply_rotary_emb(ck, rope_cond)
        q, k, v = torch.cat([q, cq], 2), torch.cat([k, ck], 2), torch.cat([v, cv], 2)
    c_factor = getattr(attn, "c_factor", None)
    if c_factor is not None:
        c_factor = float(torch.as_tensor(c_factor).flatten()[0])
    mask = attention_mask(q.shape[2], n_cond, model_config, c_factor, q.dtype, q.device) if n_cond else None
    o = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=0.0, is_causal=False)
    b = o.shape[0]
    o = o.transpose(1, 2).reshape(b, -1, H * o.shape[-1]).to(q.dtype)
    enc_o = o[:, :n_enc] if n_enc else None
    end = o.shape[1] - n_cond
    hid_o = o[:, n_enc:end]
    cond_o = o[:, end:] if n_cond else None
    return hid_o, enc_o, cond_o


def attn_forward(attn, hidden_states, encoder_hidden_states=None, condition_latents=None,
                 attention_mask=None, image_rotary_emb=None, cond_rotary_emb=None, model_config={}):
    """Same return convention as block.py:162-176."""
    hid_o, enc_o, cond_o = joint_attention(attn, hidden_states, encoder_hidden_states, condition_latents,
                                           image_rotary_emb, cond_rotary_emb, model_config)
    if encoder_hidden_states is not None:
        hid_o = _linear(attn.to_out[0], hid_o, bool(model_config.get("latent_lora", False)))
        enc_o = attn.to_add_out(enc_o)
        if cond_o is not None:
            cond_o = _linear(attn.to_out[0], cond_o, True)
            return hid_o, enc_o, cond_o
        return hid_o, enc_o
    if cond_o is not None:
        return hid_o, cond_o
    return hid_o


def _ada_zero(norm, x, emb, lora_on):
    e = _linear(norm.linear, F.silu(emb), lora_on)
    sh_a, sc_a, g_a, sh_m, sc_m, g_m = e.chunk(6, dim=1)
    return norm.norm(x) * (1 + sc_a[:, None]) + sh_a[:, None], g_a, sh_m, sc_m, g_m


def _ff(ff, x, lora_on):
    return _linear(ff.net[2], ff.net[0](x), lora_on)


def block_forward(block, hidden_states, encoder_hidden_states, condition_latents, temb, cond_temb,
                  cond_rotary_emb=None, image_rotary_emb=None, model_config={}):
    ll = bool(model_config.get("latent_lora", False))
    use_cond = condition_latents is not None
    nh, g_a, sh_m, sc_m, g_m = _ada_zero(block.norm1, hidden_states, temb, ll)
    ne, cg_a, csh_m, csc_m, cg_m = _ada_zero(block.norm1_context, encoder_hidden_states, temb, False)
    nc = None
    if use_cond:
        nc, kg_a, ksh_m, ksc_m, kg_m = _ada_zero(block.norm1, condition_latents, cond_temb, True)
    res = attn_forward(block.attn, nh, ne, nc, None, image_rotary_emb,
                       cond_rotary_emb if use_cond else None, model_config)
    hidden_states = hidden_states + g_a[:, None] * res[0]
    encoder_hidden_states = encoder_hidden_states + cg_a[:, None] * res[1]
    if use_cond:
        c_attn = kg_a[:, None] * res[2]
        condition_latents = condition_latents + c_attn
        if model_config.get("add_cond_attn", False):
            hidden_states = hidden_states + c_attn
    nh = block.norm2(hidden_states) * (1 + sc_m[:, None]) + sh_m[:, None]
    ne = block.norm2_context(encoder_hidden_states) * (1 + csc_m[:, None]) + csh_m[:, None]
    hidden_states = hidden_states + g_m[:, None] * _ff(block.ff, nh, ll)
    encoder_hidden_states = encoder_hidden_states + cg_m[:, None] * _ff(block.ff_context, ne, False)
    if use_cond:
        nc = block.norm2(condition_latents) * (1 + ksc_m[:, None]) + ksh_m[:, None]
        condition_latents = condition_latents + kg_m[:, None] * _ff(block.ff, nc, True)
    if encoder_hidden_states.dtype == torch.float16:
        encoder_hidden_states = encoder_hidden_states.clip(-65504, 65504)
    return encoder_hidden_states, hidden_states, (condition_latents if use_cond else None)


def _ada_single(norm, x, emb, lora_on):
    e = _linear(norm.linear, F.silu(emb), lora_on)
    sh, sc, g = e.chunk(3, dim=1)
    return norm.norm(x) * (1 + sc[:, None]) + sh[:, None], g


def single_block_forward(block, hidden_states, temb, image_rotary_emb=None, condition_latents=None,
                         cond_temb=None, cond_rotary_emb=None, model_config={}):
    ll = bool(model_config.get("latent_lora", False))
    use_cond = condition_latents is not None
    nh, gate = _ada_single(block.norm, hidden_states, temb, ll)
    mlp = F.gelu(_linear(block.proj_mlp, nh, ll), approximate="tanh")
    nc = None
    if use_cond:
        nc, cgate = _ada_single(block.norm, condition_latents, cond_temb, True)
        cmlp = F.gelu(_linear(block.proj_mlp, nc, True), approximate="tanh")
    hid_o, _, cond_o = joint_attention(block.attn, nh, None, nc, image_rotary_emb,
                                       cond_rotary_emb if use_cond else None, model_config)
    out = hidden_states + gate[:, None] * _linear(block.proj_out, torch.cat([hid_o, mlp], 2), ll)
    if out.dtype == torch.float16:
        out = out.clip(-65504, 65504)
    if not use_cond:
        return out
    cout = condition_latents + cgate[:, None] * _linear(block.proj_out, torch.cat([cond_o, cmlp], 2), True)
    return out, cout


def tranformer_forward(transformer, condition_latents, condition_ids, condition_type_ids=None,
                       model_config={}, c_t=0, *, hidden_states, encoder_hidden_states, pooled_projections,
                       timestep, img_ids, txt_ids, guidance=None, **_ignored):
    """One velocity prediction (transformer.py:47-252). Returns (sample,)."""
    tr = transformer
    ll = bool(model_config.get("latent_lora", False))
    use_cond = condition_latents is not None
    hid = _linear(tr.x_embedder, hidden_states, ll)
    cond = _linear(tr.x_embedder, condition_latents, True) if use_cond else None
    t = timestep.to(hid.dtype) * 1000
    ct = torch.ones_like(t) * c_t * 1000
    if guidance is not None:
        g = guidance.to(hid.dtype) * 1000
        temb = tr.time_text_embed(t, g, pooled_projections)
        cond_temb = tr.time_text_embed(ct, g, pooled_projections)
    else:
        temb = tr.time_text_embed(t, pooled_projections)
        cond_temb = tr.time_text_embed(ct, pooled_projections)
    enc = tr.context_embedder(encoder_hidden_states)
    if txt_ids.ndim == 3:
        txt_ids = txt_ids[0]
    if img_ids.ndim == 3:
        img_ids = img_ids[0]
    rope_main = tr.pos_embed(torch.cat((txt_ids, img_ids), dim=0))
    rope_cond = tr.pos_embed(condition_ids) if use_cond else None
    for blk in tr.transformer_blocks:
        enc, hid, cond = block_forward(blk, hid, enc, cond, temb, cond_temb if use_cond else None,
                                       rope_cond, rope_main, model_config)
    n_txt = enc.shape[1]
    hid = torch.cat([enc, hid], dim=1)
    for blk in tr.single_transformer_blocks:
        r = single_block_forward(blk, hid, temb, rope_main, cond, cond_temb if use_cond else None,
                                 rope_cond, model_config)
        hid, cond = r if use_cond else (r, None)
    hid = hid[:, n_txt:]
    return (tr.proj_out(tr.norm_out(hid, temb)),)


def denoise_loop(transformer, scheduler, latents, prompt_embeds, pooled, txt_ids, img_ids,
                 condition_latents, condition_ids, num_inference_steps=28, guidance_scale=3.5,
                 model_config={}):
    """generate.py:290-369 restated for pre-encoded inputs (output_type='latent')."""
    import numpy as np
    from .flux_modules import calculate_shift, retrieve_timesteps
    sigmas = np.linspace(1.0, 1 / num_inference_steps, num_inference_steps)
    cfg = scheduler.config
    mu = calculate_shift(latents.shape[1], cfg.base_image_seq_len, cfg.max_image_seq_len,
                         cfg.base_shift, cfg.max_shift)
    timesteps, _ = retrieve_timesteps(scheduler, num_inference_steps, latents.device, None, sigmas, mu=mu)
    for t in timesteps:
        ts = t.expand(latents.shape[0]).to(latents.dtype)
        guidance = None
        if transformer.config.guidance_embeds:
            guidance = torch.tensor([guidance_scale], device=latents.device).expand(latents.shape[0])
        v = tranformer_forward(transformer, condition_latents, condition_ids, None, model_config,
                               hidden_states=latents, encoder_hidden_states=prompt_embeds,
                               pooled_projections=pooled, timestep=ts / 1000, img_ids=img_ids,
                               txt_ids=txt_ids, guidance=guidance)[0]
        latents = scheduler.step(v, t, latents)[0]
    return latents
