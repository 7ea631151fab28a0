"""Which op must be synchronised for the run-to-run differences to disappear? DET_SYNC=comma list of ops.* names after which the host
waits for the GPU (none = baseline)."""
import os, sys, torch
os.environ.setdefault("LX_GRAPH", "0")
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
_dummy = torch.zeros(1024, device="cuda")
MODE = os.environ.get("DET_MODE", "sync")       # sync | dummy (a tiny kernel instead of a host wait)
def wrap(name, before):
    fn = getattr(ops, name)
    def w(*a, **k):
        if before: torch.cuda.synchronize() if MODE == "sync" else _dummy.add_(1.0)
        res = fn(*a, **k)
        if not before: torch.cuda.synchronize() if MODE == "sync" else _dummy.add_(1.0)
        return res
    setattr(ops, name, w)
for n_ in [s for s in os.environ.get("DET_SYNC", "").split(",") if s]:
    wrap(n_.lstrip("^"), n_.startswith("^"))
def run():
    c = Condition("subject", latents=x["cond"], latent_hw=(hw, hw), position_delta=[0, -hw])
    return generate(model, model.flux_pipe, conditions=[c], height=512, width=512, num_inference_steps=2, latents=x["lat"], prompt_embeds=x["pe"],
                    pooled_prompt_embeds=x["pooled"], output_type="latent", model_config=model.model_config, default_lora=True, use_brain_condition=False).images.clone()
run(); ref = run()
n = int(os.environ.get("DET_N", "300")); bad = 0
prev = ref; same_prev = 0; shown = 0
for i in range(n):
    o = run()
    if torch.equal(o, prev): same_prev += 1
    if not torch.equal(o, ref):
        bad += 1
        if shown < 4:
            shown += 1
            d = (o - ref).abs(); bs = sorted(set((d.amax(-1) > 0).nonzero()[:, 0].tolist()))
            print(f"  run {i}: {int((d > 0).sum())} differ, max {float(d.max()):.3e}, batches {bs}, equals previous run: {torch.equal(o, prev)}")
    prev = o
print("runs equal to their predecessor:", same_prev, "of", n)
print("mismatching runs:", bad, "of", n, "sync:", os.environ.get("DET_SYNC", ""), "graph:", os.environ.get("LX_GRAPH"))
