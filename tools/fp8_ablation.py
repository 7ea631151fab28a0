#!/usr/bin/env python3
"""Full-depth fp8 error budget of the DiT forward: which operands tolerate e4m3, and whether a scaling recipe changes that.

    python tools/fp8_ablation.py [--out gpurun_out/fp8_ablation.json] [--steps 0,13,27] [--quick]

Measurement tool (GPU only; not part of the product or of the tests). For BASELINE configs[4] ("fp8 MFMA attention path") the
contract is "the bf16 result within a stated tolerance" -- the reference has no fp8 path (block.py:129 is plain SDPA). This tool
measures, at FULL depth (19 + 38 blocks, S = 2560) against the fp32 oracle on the oracle's own trajectory (teacher-forced, as
oracle/parity.py), what each use of e4m3 costs:

  * per GEMM kind (double q/k/v, to_out, ff1, ff2; single fused [q|k|v|mlp], single proj_out): the bf16 kernels are fed operands
    that were rounded to e4m3 first (fake quantisation with power-of-two scales, so every operand value is exactly representable
    in bf16 and the bf16 MFMA GEMM computes exactly what an e4m3 MFMA GEMM with those scales would: same products, fp32 sums);
  * three scaling recipes: `fixed` (activation scale 16, per-output-row weight scale: what lx_gemm_fp8_kernel ships),
    `token` (per-token dynamic activation scale from the row's amax), `mx` (OCP MX: one E8M0 scale per 32 elements along K
    on both operands -- the operand form of v_mfma_scale_f32_32x32x64_f8f6f4);
  * the real fp8 kernels (model_config attn_fp8 / gemm_fp8) for comparison, which also validates the fake-quantisation model;
  * attention operands alone: q/k rounded to e4m3 (scores) and V rounded to e4m3, through the bf16 attention kernel.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("LX_GRAPH", "0")          # the hooks below change what a step launches: run the eager path

import numpy as np
import torch

from loongx_amd import ops
from loongx_amd.flux.transformer import tranformer_forward
from oracle import flux_modules as fm
from oracle import flux_ref as fr
from oracle.parity import build_pair, relerr

E4M3_MAX = 448.0


def _e4m3(x: torch.Tensor) -> torch.Tensor:
    return x.clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).float()


def _pow2_floor(x: torch.Tensor) -> torch.Tensor:
    return torch.exp2(torch.floor(torch.log2(x.clamp_min(1e-30))))


def fq_fixed(x: torch.Tensor, scale: float = 16.0) -> torch.Tensor:
    return _e4m3(x * scale) / scale


def fq_rows(x: torch.Tensor) -> torch.Tensor:
    """per-row power-of-two scale that puts the row's amax just under 448"""
    s = _pow2_floor(E4M3_MAX / x.abs().amax(dim=1, keepdim=True).clamp_min(1e-30))
    return _e4m3(x * s) / s


def fq_mx(x: torch.Tensor) -> torch.Tensor:
    """OCP MX e4m3: shared E8M0 exponent per 32 consecutive elements along K = floor(log2(amax)) - 8 (e4m3's emax)"""
    M, K = x.shape
    b = x.reshape(M, K // 32, 32)
    s = torch.exp2(8.0 - torch.floor(torch.log2(b.abs().amax(dim=2, keepdim=True).clamp_min(1e-30))))
    return (_e4m3(b * s) / s).reshape(M, K)


ACT = {"fixed": fq_fixed, "token": fq_rows, "mx": fq_mx}
WGT = {"fixed": fq_rows, "token": fq_rows, "mx": fq_mx}


def kind_of(name: str) -> str:
    blk, mod = name.split(".", 1)
    return ("d_" if blk[0] == "d" else "s_") + mod.replace("_txt", "")


class Hooks:
    """Fake quantisation of selected GEMM kinds / attention operands of one DiTEngine (eager launch path)."""

    def __init__(self, eng):
        self.eng = eng
        self.kinds, self.recipe = set(), "fixed"
        self.attn_qk = self.attn_v = False
        self.wcache = {}
        self._orig_gemm_streams = eng._gemm_streams
        self._orig_attn = ops.attn_fwd
        eng._gemm_streams = self.gemm_streams
        ops.attn_fwd = self.attn_fwd

    def _wq(self, name: str) -> torch.Tensor:
        key = (name, self.recipe)
        t = self.wcache.get(key)
        if t is None:
            W = self.eng.w.t[name + ".w"]
            tiled = getattr(W, "lx_tiled", False)
            Wr = ops.untile_weight(W) if tiled else W
            q = WGT[self.recipe](Wr.float()).to(torch.bfloat16)
            t = ops.tile_weight(q) if tiled else q
            self.wcache[key] = t
        return t

    def gemm_streams(self, A, Cbuf, main, txt, **kw):
        if kind_of(main) not in self.kinds:
            return self._orig_gemm_streams(A, Cbuf, main, txt, **kw)
        eng = self.eng
        if kw.get("lora") is None:               # the adapters' down-projection reads the unquantised operand, as the fp8 engine path does for q/k/v
            lo, r0 = eng._lora_t(A, main, include_txt=txt is None)
            if lo is not None:
                kw["lora"] = (lo, r0)
        Aq = ACT[self.recipe](A.float()).to(torch.bfloat16)
        names = [main] + ([txt] if txt is not None else [])
        saved = {n: eng.w.t[n + ".w"] for n in names}
        try:
            for n in names:
                eng.w.t[n + ".w"] = self._wq(n)
            return self._orig_gemm_streams(Aq, Cbuf, main, txt, **kw)
        finally:
            for n in names:
                eng.w.t[n + ".w"] = saved[n]

    def attn_fwd(self, Q, K, VT, O, *, q_col, k_col, o_col, **kw):
        D = self.eng.cfg.inner_dim
        if self.attn_qk:        # q, k after RMSNorm + RoPE (qkv_prep has run: LX_QKV_FUSED=0), scale 16 as the fp8 kernel uses
            for buf, c in ((Q, q_col), (K, k_col)):
                v = buf[:, c:c + D]
                v.copy_(fq_fixed(v.float(), 16.0).to(torch.bfloat16))
        if self.attn_v:         # V^T image, unscaled as the fp8 kernel stores it
            VT.copy_(_e4m3(VT.float()).to(torch.bfloat16))
        return self._orig_attn(Q, K, VT, O, q_col=q_col, k_col=k_col, o_col=o_col, **kw)


GEMM_KINDS = ["d_qkv", "d_out", "d_ff1", "d_ff2", "s_fused", "s_out"]


@torch.no_grad()
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "fp8_ablation.json"))
    ap.add_argument("--steps", default="0,13,27")
    ap.add_argument("--quick", action="store_true", help="2 + 2 blocks (plumbing check)")
    ap.add_argument("--hw", type=int, default=32)
    a = ap.parse_args()
    os.environ["LX_QKV_FUSED"] = "0"          # q / k / V^T exist as bf16 images between the projection and the attention launch
    dev = torch.device("cuda:0")
    nl, ns = (2, 2) if a.quick else (19, 38)
    steps = 28
    cmp_steps = sorted({int(s) for s in a.steps.split(",")})
    tr, lx = build_pair(dev, nl, ns)
    eng = lx.engine
    hw, n_txt = a.hw, 512
    N = hw * hw
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

    # oracle trajectory (fp32) + the reference's own bf16 mode on the same trajectory (context: what `dtype: bfloat16` costs there)
    lat, points = lat0.clone(), []
    t0 = time.time()
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
    hooks = Hooks(eng)

    def measure(label, kinds=(), recipe="fixed", mc=None, attn_qk=False, attn_v=False):
        hooks.kinds, hooks.recipe, hooks.attn_qk, hooks.attn_v = set(kinds), recipe, attn_qk, attn_v
        m = dict(mc0); m.update(mc or {})
        errs = []
        for i, kw, want in points:
            lx.invalidate_conditioning()
            got = tranformer_forward(lx, cond, cids, None, m, return_dict=False, **kw)[0]
            errs.append(relerr(got, want))
        rec = {"label": label, "kinds": list(kinds), "recipe": recipe, "model_config": mc or {}, "attn_qk_e4m3": attn_qk, "attn_v_e4m3": attn_v,
               "relerr_per_step": [round(e, 6) for e in errs], "relerr_mean": round(float(np.mean(errs)), 6)}
        print(f"{label:58s} {rec['relerr_mean']:.4e}   {['%.3e' % e for e in errs]}", flush=True)
        return rec

    out = {"blocks": [nl, ns], "tokens": [n_txt, N, N], "steps_compared": cmp_steps,
           "rows": []}
    R = out["rows"]
    R.append(measure("bf16 (no e4m3 anywhere)"))
    for k in GEMM_KINDS:
        R.append(measure(f"e4m3 operands in {k} only, fixed scales", [k]))
    for rec in ("fixed", "token", "mx"):
        R.append(measure(f"e4m3 operands in every block GEMM, {rec} scales", GEMM_KINDS, rec))
    for k in GEMM_KINDS:
        R.append(measure(f"e4m3 everywhere except {k} (mx)", [x for x in GEMM_KINDS if x != k], "mx"))
    R.append(measure("attention: q, k rounded to e4m3 (bf16 kernel)", attn_qk=True))
    R.append(measure("attention: V rounded to e4m3 (bf16 kernel)", attn_v=True))
    R.append(measure("attention: q, k, V rounded to e4m3 (bf16 kernel)", attn_qk=True, attn_v=True))
    R.append(measure("real kernel: attn_fp8", mc={"attn_fp8": True}))
    R.append(measure("real kernel: gemm_fp8", mc={"gemm_fp8": True}))
    R.append(measure("real kernels: gemm_fp8 + attn_fp8", mc={"gemm_fp8": True, "attn_fp8": True}))
    R.append(measure("fake e4m3 GEMMs (fixed) + real attn_fp8", GEMM_KINDS, "fixed", mc={"attn_fp8": True}))
    ref_bf16 = None
    try:
        trb = tr.to(torch.bfloat16)
        errs = []
        for i, kw, want in points:
            kwb = {k: (v.to(torch.bfloat16) if k in ("hidden_states", "encoder_hidden_states", "pooled_projections") else v) for k, v in kw.items()}
            got = fr.tranformer_forward(trb, cond.to(torch.bfloat16), cids, None, mc0, **kwb)[0]
            errs.append(relerr(got.float(), want))
        ref_bf16 = float(np.mean(errs))
        pass
    except Exception as e:                      # context figure only
        ref_bf16 = f"{type(e).__name__}: {e}"
    print("reference arithmetic in torch bf16 (weights, activations, residual stream) vs fp32:", ref_bf16, flush=True)

    out["reference_bf16_mode_relerr"] = ref_bf16
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
