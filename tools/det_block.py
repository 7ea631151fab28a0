"""Bisect the step: run one piece of the engine's flow repeatedly from the same X and compare what it leaves behind."""
import os, sys, torch
os.environ["LX_GRAPH"] = "0"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from tests.test_configs_gpu import _model, T
from oracle import flux_modules as fm
B = int(os.environ.get("DET_B", "4")); hw = 32; N = hw * hw
model = _model()
eng = model.flux_pipe.transformer.engine
eng.pair_plan = os.environ.get("DET_PAIR", "0") == "1"
mc = {"union_cond_attn": True}
for k_ in os.environ.get("DET_MC", "").split(","):
    if k_: mc[k_] = True
g = torch.Generator(device="cuda").manual_seed(11)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)
pe, pooled, cond, lat = r(B, T, 4096) * 0.1, r(B, 768), r(B, N, 64), r(B, N, 64)
ids = fm.prepare_latent_image_ids(hw, hw).cuda(); cids = ids.clone(); cids[:, 2] -= hw
eng.set_conditioning(pe, pooled, torch.full((B,), 3.5, device="cuda"), torch.zeros(T, 3, device="cuda"), ids, cond, cids, c_t=0.0, model_config=mc)
eng.embed_step_inputs(lat, torch.full((B,), 0.7, device="cuda"))
X0 = eng.X.clone()
D = 3072
def piece(name):
    if name == "double": eng.double_block(0); return eng.X
    if name == "single": eng.single_block(0); return eng.X
    if name == "single_last": eng.single_block(0, image_out_only=True); return eng.rows(eng.X, "img")
    if name == "both": eng.double_block(0); eng.single_block(0, image_out_only=True); return eng.rows(eng.X, "img")
    raise SystemExit(name)
n = int(os.environ.get("DET_N", "600"))
for name in os.environ.get("DET_PIECES", "double,single,single_last,both").split(","):
    eng.X.copy_(X0); ref = piece(name).clone()
    bad = 0
    for i in range(n):
        eng.X.copy_(X0)
        o = piece(name)
        if not torch.equal(o, ref):
            bad += 1
            if bad <= 3:
                d = (o - ref).abs(); rows = (d.amax(-1) > 0).nonzero().flatten(); cols = (d.amax(0) > 0).nonzero().flatten()
                print(f"  {name} run {i}: {int((d > 0).sum())} differ, max {float(d.max()):.3e}; rows {rows[0].item()}..{rows[-1].item()} ({len(rows)}), cols {cols[0].item()}..{cols[-1].item()} ({len(cols)})")
    print(f"piece {name}: {bad} of {n} runs differ", {k: v for k, v in os.environ.items() if (k.startswith('LX_') and k != 'LX_GRAPH') or k.startswith('DET_')})
