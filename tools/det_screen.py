"""Repeat the batch-16 all-modality generate() of tests/test_configs_gpu.py and count run-to-run differences (bisect with env switches)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from tests.test_configs_gpu import _model, T
from loongx_amd.flux.condition import Condition
from loongx_amd.flux.generate import generate
brain = os.environ.get("DET_BRAIN", "1") == "1"
B = int(os.environ.get("DET_B", "16")); hw = 32; N = hw * hw
model = _model()
g = torch.Generator(device="cuda").manual_seed(11)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)
x = dict(lat=r(B, N, 64), cond=r(B, N, 64), pe=r(B, T, 4096) * 0.1, pooled=r(B, 768), eeg=r(B, 4, 4096), fnirs=r(B, 6, 512), ppg=r(B, 4, 256), motion=r(B, 6, 128))
def run():
    c = Condition("subject", latents=x["cond"], latent_hw=(hw, hw), position_delta=[0, -hw])
    kw = dict(additional_condition1=x["eeg"], additional_condition2=x["fnirs"], additional_condition3=x["ppg"], additional_condition4=x["motion"]) if brain else {}
    return generate(model, model.flux_pipe, conditions=[c], height=512, width=512, num_inference_steps=2, latents=x["lat"], prompt_embeds=x["pe"],
                    pooled_prompt_embeds=x["pooled"], output_type="latent", model_config=model.model_config, default_lora=True,
                    use_brain_condition=brain, fuse_flag=True, **kw).images.clone()
model.flux_pipe.transformer.engine.pair_plan = os.environ.get("DET_PAIR", "0") == "1"
ref = run()
bad = 0
n = int(os.environ.get("DET_N", "12"))
for i in range(n):
    o = run()
    if not torch.equal(o, ref):
        d = (o - ref).abs()
        rows = (d.amax(-1) > 0).nonzero()
        bad += 1
        print(f"  run {i}: {int((d > 0).sum())} elements differ, max {float(d.max()):.3e}; batches {sorted(set(rows[:, 0].tolist()))[:8]} rows {rows[:4, 1].tolist()}")
print("mismatching runs:", bad, "of", n, {k: v for k, v in os.environ.items() if k.startswith(("LX_", "DET_"))})
