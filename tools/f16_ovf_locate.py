#!/usr/bin/env python3
"""Which launch of an fp16-operand forward raises the saturation counter? Wraps the ops entry points of a 1 + 1-block full-width engine,
reads the counter after every call (eager path) and prints the calls that moved it, with the largest |value| of what they wrote."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["LX_GRAPH"] = "0"
import torch
from loongx_amd import ops
from loongx_amd.flux.transformer import LxFluxTransformer, tranformer_forward
from loongx_amd.flux.weights import FluxConfig
from oracle import flux_modules as fm

dev = torch.device("cuda:0")
nl, ns = int(os.environ.get("NL", 1)), int(os.environ.get("NS", 1))
lx = LxFluxTransformer.synthetic(FluxConfig(num_layers=nl, num_single_layers=ns), dev, seed=0)
eng = lx.engine
hw, n_txt = 32, 512
N = hw * hw
g = torch.Generator(device=dev).manual_seed(4321)
r = lambda *s: torch.randn(*s, device=dev, generator=g)
lat0, cond, pe, pooled = r(1, N, 64), r(1, N, 64), r(1, n_txt, 4096) * 0.1, r(1, 768)
ids = fm.prepare_latent_image_ids(hw, hw).to(dev)
cids = ids.clone(); cids[:, 2] -= hw
log = []


def wrap(name):
    fn = getattr(ops, name)

    def w(*a, **k):
        before = int(eng.f16_ovf.item()) if eng.f16_ovf is not None else 0
        out = fn(*a, **k)
        torch.cuda.synchronize()
        after = int(eng.f16_ovf.item()) if eng.f16_ovf is not None else 0
        if after != before:
            desc = ""
            if name == "gemm":
                desc = " | ".join(f"M={p.M} N={p.N} K={p.K} epi={p.epilogue:#x}" for p in a[0])
            log.append((name, after - before, desc))
        return out
    setattr(ops, name, w)


for n in ("gemm", "ln_modulate_segs", "ln_modulate", "attn_fwd", "convert", "lora_down"):
    wrap(n)
kw = dict(hidden_states=lat0, encoder_hidden_states=pe, pooled_projections=pooled, timestep=torch.tensor([0.7], device=dev), img_ids=ids,
          txt_ids=torch.zeros(n_txt, 3, device=dev), guidance=torch.full((1,), 3.5, device=dev))
out = tranformer_forward(lx, cond, cids, None, {"union_cond_attn": True, "operands": "fp16"}, return_dict=False, **kw)[0]
print("finite", bool(torch.isfinite(out).all()), "count", eng.f16_overflow_count(reset=False))
for e in log:
    print(e)
print("max |XN16|", float(eng.XN16.float().abs().max()), "max |Y16[:, 2D:]|", float(eng.Y16[:, 2 * 3072:].float().abs().max()))
