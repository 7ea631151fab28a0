"""Locate run-to-run nondeterminism in the DiT step: checksum the engine's buffers after every kernel-level op of an eager generate()
and report the first op whose checksum differs from the first run's."""
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
trace = []
def cks():
    out = []
    for name in ("X", "XN", "Y", "VT", "TLs", "mods", "cmods", "out"):
        t = getattr(eng, name, None)
        if t is None: continue
        v = t.view(torch.int16) if t.dtype == torch.bfloat16 else t.view(torch.int32)
        out.append((name, v.sum(dtype=torch.int64)))            # device scalar: no host synchronisation between the ops
    return out
def wrap(name):
    fn = getattr(ops, name)
    def w(*a, **k):
        res = fn(*a, **k)
        trace.append((name, cks()))
        return res
    setattr(ops, name, w)
for n_ in ("gemm", "attn_fwd", "ln_modulate_segs", "ln_modulate", "lora_down", "qkv_prep_segs", "linear_skinny", "linear_f32", "convert", "euler_step", "timestep_embed", "rope_table"):
    if hasattr(ops, n_): wrap(n_)
def run():
    trace.clear()
    c = Condition("subject", latents=x["cond"], latent_hw=(hw, hw), position_delta=[0, -hw])
    out = generate(model, model.flux_pipe, conditions=[c], height=512, width=512, num_inference_steps=2, latents=x["lat"], prompt_embeds=x["pe"],
                   pooled_prompt_embeds=x["pooled"], output_type="latent", model_config=model.model_config, default_lora=True, use_brain_condition=False).images.clone()
    torch.cuda.synchronize()
    return out, [(n_, [(k, int(v)) for k, v in c]) for n_, c in trace]
run()                      # warm-up: later runs start from the buffers the previous run left behind
ref, tref = run()
print("ops per generate:", len(tref))
n = int(os.environ.get("DET_N", "150")); bad = 0
for i in range(n):
    o, t = run()
    if t != tref:
        bad += 1
        for j, (a, b_) in enumerate(zip(t, tref)):
            if a != b_:
                diff = [na for (na, va), (nb, vb) in zip(a[1], b_[1]) if va != vb]
                prev = t[j - 1][0] if j else None
                if bad <= 6: print(f"  run {i}: first divergence at op #{j} = {a[0]} (previous op {prev}); buffers that differ: {diff}; final equal: {torch.equal(o, ref)}")
                break
print("diverging runs:", bad, "of", n)
