#!/usr/bin/env python3
"""Full-depth bf16 error budget of the DiT forward: WHICH rounding to bf16 carries the headline mode's 4.9e-3 per forward?

    python tools/bf16_ablation.py [--out gpurun_out/bf16_ablation.json] [--steps 0,13,27] [--quick]

Measurement tool (GPU only; not part of the product or of the tests), the bf16 counterpart of tools/fp8_ablation.py. The product's
bf16 mode keeps the residual stream, LayerNorm / modulation, RMSNorm / RoPE arithmetic, softmax statistics and every accumulation in
fp32; what it rounds to bf16 are the matrix-pipe operands. This tool puts exactly those roundings, ONE SITE AT A TIME, into the fp32
oracle (oracle/flux_ref.py, torch on this GPU; weights are bf16-representable in both) at FULL depth (19 + 38 blocks, S = 2560) on
the oracle's own trajectory, and measures the per-forward relative error each site alone produces:

  A operand of a GEMM kind   d_qkv / d_ff1 / s_fused: the LayerNorm-modulated activations (lx_ln_modulate writes bf16);
                             d_out / s_out: the attention output (and, for s_out, the GELU(MLP) half) as stored by the producer;
                             d_ff2: the GELU output of ff1
  attention operands         q, k after RMSNorm + RoPE;  v;  P (the unnormalised probabilities that enter P.V; l stays fp32)
  everything                 all sites together = the model of the product's bf16 mode (compared with the engine itself at the end)

"leave one out" rows (all sites but one) say what a mode that keeps ONE site at 16 mantissa bits (split-bf16 hi + lo, as precise mode
does everywhere) would buy."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.nn.functional as TF

from loongx_amd.flux.transformer import tranformer_forward
from oracle import flux_modules as fm
from oracle import flux_ref as fr
from oracle.parity import build_pair, relerr

ACTIVE = set()
FMT = {"gemm": torch.bfloat16, "attn": torch.bfloat16}     # the 16-bit format of the GEMM A operands / of q, k, v, P (--fmt)


def r16(x, which="gemm"):
    if FMT[which] == torch.float16:      # saturating, as the product's converts are
        x = x.clamp(-65504.0, 65504.0)
    return x.to(FMT[which]).to(x.dtype)


def hook_inputs(mod, kind):
    def pre(_m, inp):
        if kind in ACTIVE:
            return (r16(inp[0]),) + tuple(inp[1:])
        return None
    mod.register_forward_pre_hook(pre)


def tag(tr):
    """forward pre-hooks on every Linear whose A operand the product stores in bf16 (LoRA wrappers: the base layer and the adapter's
    down-projection read the same bf16 buffer)"""
    def lin(m, kind):
        if isinstance(m, fm.LoraLinear):
            hook_inputs(m.base_layer, kind)
            for a in m.lora_A.values():
                hook_inputs(a, kind)
        else:
            hook_inputs(m, kind)
    for b in tr.transformer_blocks:
        for m in (b.attn.to_q, b.attn.to_k, b.attn.to_v, b.attn.add_q_proj, b.attn.add_k_proj, b.attn.add_v_proj):
            lin(m, "d_qkv")
        lin(b.attn.to_out[0], "d_out"); lin(b.attn.to_add_out, "d_out")
        for ff in (b.ff, b.ff_context):
            hook_inputs(ff.net[0], "d_ff1")          # GELU(tanh) o Linear: its input is ff1's A operand
            lin(ff.net[2], "d_ff2")
    for b in tr.single_transformer_blocks:
        for m in (b.attn.to_q, b.attn.to_k, b.attn.to_v, b.proj_mlp):
            lin(m, "s_fused")
        lin(b.proj_out, "s_out")
    # round 5: the three GEMMs outside the blocks whose A operand the product also stores in 16 bits (the first pass of this tool
    # left them out: they are the 2e-3 between its "every site" row and the engine)
    lin(tr.x_embedder, "x_emb")                      # latents / condition latents -> D
    hook_inputs(tr.context_embedder, "ctx_emb")      # T5 states -> D
    hook_inputs(tr.proj_out, "final")                # norm_out's output -> 64 channels


class FProxy:
    """torch.nn.functional with scaled_dot_product_attention replaced by an explicit softmax that can round q / k, v and P"""

    def __getattr__(self, name):
        return getattr(TF, name)

    @staticmethod
    def scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False):
        if not ({"attn_qk", "attn_v", "attn_p", "attn_explicit"} & ACTIVE):
            return TF.scaled_dot_product_attention(q, k, v, attn_mask=attn_mask)
        if "attn_qk" in ACTIVE:
            q, k = r16(q, "attn"), r16(k, "attn")
        if "attn_v" in ACTIVE:
            v = r16(v, "attn")
        out = torch.empty_like(q)
        scale = q.shape[-1] ** -0.5
        for h0 in range(0, q.shape[1], 4):                       # four heads at a time: [4, S, S] fp32 scores
            s = (q[:, h0:h0 + 4] @ k[:, h0:h0 + 4].transpose(-1, -2)) * scale
            if attn_mask is not None:
                s = s + attn_mask if attn_mask.dtype != torch.bool else s.masked_fill(~attn_mask, float("-inf"))
            p = torch.exp(s - s.amax(dim=-1, keepdim=True))      # (rounding to bf16 is scale-free up to the position in the binade)
            l = p.sum(dim=-1, keepdim=True)
            if "attn_p" in ACTIVE:
                p = r16(p, "attn")
            out[:, h0:h0 + 4] = (p @ v[:, h0:h0 + 4]) / l
        return out


SITES = ["d_qkv", "d_out", "d_ff1", "d_ff2", "s_fused", "s_out", "attn_qk", "attn_v", "attn_p", "x_emb", "ctx_emb", "final"]
GEMM_SITES = SITES[:6] + SITES[9:]
ATTN_SITES = SITES[6:9]


@torch.no_grad()
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "bf16_ablation.json"))
    ap.add_argument("--steps", default="0,13,27")
    ap.add_argument("--quick", action="store_true", help="2 + 2 blocks (plumbing check)")
    ap.add_argument("--fmt", default="bf16", choices=["bf16", "fp16", "both"], help="16-bit operand format of the GEMM A operands")
    ap.add_argument("--short", action="store_true", help="skip the leave-one-out rows")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    nl, ns = (2, 2) if a.quick else (19, 38)
    steps, hw, n_txt = 28, 32, 512
    N = hw * hw
    cmp_steps = sorted({int(s) for s in a.steps.split(",")})
    tr, lx = build_pair(dev, nl, ns)
    tag(tr)
    fr.F = FProxy()
    g = torch.Generator(device=dev).manual_seed(4321)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)
    lat0, cond, pe, pooled = r(1, N, 64), r(1, N, 64), r(1, n_txt, 4096) * 0.1, r(1, 768)
    ids = fm.prepare_latent_image_ids(hw, hw).to(dev)
    cids = ids.clone(); cids[:, 2] -= hw
    txt_ids = torch.zeros(n_txt, 3, device=dev)
    guidance = torch.full((1,), 3.5, device=dev)
    sch = fm.FlowMatchEulerDiscreteScheduler()
    sig = np.linspace(1.0, 1 / steps, steps)
    mu = fm.calculate_shift(N, sch.config.base_image_seq_len, sch.config.max_image_seq_len, sch.config.base_shift, sch.config.max_shift)
    timesteps, _ = fm.retrieve_timesteps(sch, steps, dev, None, sig, mu=mu)
    mc0 = {"union_cond_attn": True}

    lat, points = lat0.clone(), []
    t0 = time.time()
    ACTIVE.clear()
    for i, t in enumerate(timesteps):
        ts = t.expand(1).to(lat.dtype) / 1000
        kw = dict(hidden_states=lat, encoder_hidden_states=pe, pooled_projections=pooled, timestep=ts, img_ids=ids, txt_ids=txt_ids, guidance=guidance)
        want = fr.tranformer_forward(tr, cond, cids, None, mc0, **kw)[0]
        if i in cmp_steps:
            points.append((i, {k: v.clone() for k, v in kw.items()}, want.clone()))
        lat = sch.step(want, t, lat)[0]
        if i >= max(cmp_steps):
            break
    print(f"oracle trajectory to step {max(cmp_steps)}: {time.time() - t0:.1f}s", flush=True)

    rows = []

    def measure(label, sites):
        ACTIVE.clear(); ACTIVE.update(sites)
        errs = [relerr(fr.tranformer_forward(tr, cond, cids, None, mc0, **kw)[0], want) for _, kw, want in points]
        ACTIVE.clear()
        rec = {"label": label, "sites": sorted(sites), "relerr_per_step": [round(e, 6) for e in errs], "relerr_mean": round(float(np.mean(errs)), 6)}
        rows.append(rec)
        print(f"{label:64s} {rec['relerr_mean']:.4e}   {['%.3e' % e for e in errs]}", flush=True)
        return rec["relerr_mean"]

    measure("fp32 oracle through the explicit-softmax path (no rounding)", {"attn_explicit"})      # (sanity: the hooks themselves are exact)
    ACTIVE.clear()
    summary = {}
    for fmt in (["bf16", "fp16"] if a.fmt == "both" else [a.fmt]):
        FMT["gemm"] = FMT["attn"] = torch.float16 if fmt == "fp16" else torch.bfloat16
        one = {s: measure(f"[{fmt}] only {s} rounded", {s}) for s in SITES}
        every = measure(f"[{fmt}] every site rounded (model of the {fmt}-operand mode)", set(SITES))
        rss = float(np.sqrt(sum(v * v for v in one.values())))
        print(f"[{fmt}] root-sum-square of the single-site errors: {rss:.4e} (independent roundings add in quadrature)")
        gemm_only = measure(f"[{fmt}] GEMM A operands only ({' '.join(GEMM_SITES)})", set(GEMM_SITES))
        attn_only = measure(f"[{fmt}] attention operands only (q k v P)", set(ATTN_SITES))
        summary[fmt] = {"every": every, "rss": rss, "gemm_only": gemm_only, "attn_only": attn_only, "one": one}
        if not a.short:
            for s in SITES:
                measure(f"[{fmt}] every site but {s}", set(SITES) - {s})
            measure(f"[{fmt}] every site but the two largest", set(SITES) - set(sorted(one, key=one.get)[-2:]))
    if a.fmt in ("fp16", "both"):
        # the mode the review proposes: fp16 GEMM A operands, attention operands (q, k, v, P) still bf16
        FMT["gemm"], FMT["attn"] = torch.float16, torch.bfloat16
        summary["fp16_gemm_bf16_attn"] = measure("fp16 GEMM A operands + bf16 attention operands (the proposed mode)", set(SITES))
    FMT["gemm"] = FMT["attn"] = torch.bfloat16

    # the engine itself (bf16 mode, default plans) on the same points
    errs = []
    for _, kw, want in points:
        lx.invalidate_conditioning()
        errs.append(relerr(tranformer_forward(lx, cond, cids, None, mc0, return_dict=False, **kw)[0], want))
    eng = float(np.mean(errs))
    print(f"{'the engine (bf16 mode)':64s} {eng:.4e}   {['%.3e' % e for e in errs]}")
    out = {"blocks": [nl, ns], "tokens": [n_txt, N, N], "steps_compared": cmp_steps, "rows": rows, "summary": summary,
           "engine_bf16_relerr_mean": round(eng, 6), "engine_bf16_relerr_per_step": [round(e, 6) for e in errs]}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
