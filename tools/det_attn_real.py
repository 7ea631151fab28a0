"""Replay the engine's own attention calls (captured operands) many times: is lx_attn_fwd deterministic on the data the model produces?"""
import os, sys, torch
os.environ["LX_GRAPH"] = "0"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from tests.test_configs_gpu import _model, T
from loongx_amd import ops
from loongx_amd.flux.condition import Condition
from loongx_amd.flux.generate import generate
B = int(os.environ.get("DET_B", "4")); hw = 32; N = hw * hw
model = _model()
eng = model.flux_pipe.transformer.engine
eng.pair_plan = False
g = torch.Generator(device="cuda").manual_seed(11)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)
x = dict(lat=r(B, N, 64), cond=r(B, N, 64), pe=r(B, T, 4096) * 0.1, pooled=r(B, 768))
caps = []
orig = ops.attn_fwd
def cap(Q, K, VT, O, **kw):
    caps.append((Q.clone(), VT.clone(), dict(kw)))
    return orig(Q, K, VT, O, **kw)
ops.attn_fwd = cap
c = Condition("subject", latents=x["cond"], latent_hw=(hw, hw), position_delta=[0, -hw])
generate(model, model.flux_pipe, conditions=[c], height=512, width=512, num_inference_steps=2, latents=x["lat"], prompt_embeds=x["pe"],
         pooled_prompt_embeds=x["pooled"], output_type="latent", model_config=model.model_config, default_lora=True, use_brain_condition=False)
ops.attn_fwd = orig
print("captured attention calls:", len(caps))
n = int(os.environ.get("DET_N", "500"))
for ci, (Y0, VT0, kw) in enumerate(caps):
    Yw = Y0.clone()
    orig(Yw, Yw, VT0, Yw, **kw)
    D = 3072
    ref = Yw[:, 2 * D:3 * D].clone()
    bad = 0
    for i in range(n):
        Yw.copy_(Y0)
        orig(Yw, Yw, VT0, Yw, **kw)
        o = Yw[:, 2 * D:3 * D]
        if not torch.equal(o.view(torch.int16), ref.view(torch.int16)):
            bad += 1
            if bad <= 3:
                nz = (o.view(torch.int16) != ref.view(torch.int16))
                rows = nz.any(-1).nonzero().flatten(); cols = nz.any(0).nonzero().flatten()
                d = (o.float() - ref.float()).nan_to_num().abs()
                print(f"  call {ci} run {i}: {int(nz.sum())} elements differ, max {float(d.max()):.3e}; rows {rows[0].item()}..{rows[-1].item()} ({len(rows)}), cols {cols[0].item()}..{cols[-1].item()} ({len(cols)})")
    print(f"attention call {ci}: {bad} of {n} replays differ", {k: v for k, v in os.environ.items() if k.startswith('LX_') and k != 'LX_GRAPH'})
