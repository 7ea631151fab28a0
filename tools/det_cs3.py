"""Run-to-run determinism of the CS3 encoders + DGF fusion at batch 16 (cross-wave LDS carries in the S4 scan, rank-count masks)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from loongx_amd.train.model import OminiModel, synthetic_cs3_state_dict
B = int(os.environ.get("DET_B", "16"))
m = OminiModel.from_pipe(None, synthetic_cs3_state_dict(0), {}, "cuda")
g = torch.Generator(device="cuda").manual_seed(3)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)
x = dict(eeg=r(B, 4, 4096), fnirs=r(B, 6, 512), ppg=r(B, 4, 256), motion=r(B, 6, 128), pe=r(B, 512, 4096) * 0.1, pooled=r(B, 768))
def run():
    e = m.eeg_projection(x["eeg"]); p = m.ppg_projection(x["ppg"]); f = m.fnirs_projection(x["fnirs"]); mo = m.motion_projection(x["motion"])
    fe = m.fuse_eeg(e, p); ff = m.fuse_fnirs(f, mo)
    a = m.duan_norm_prompt(x["pe"], fe) if hasattr(m, "duan_norm_prompt") else fe
    return torch.cat([t.float().flatten() for t in (e, p, f, mo, fe, ff, a)])
ref = run().clone()
n = int(os.environ.get("DET_N", "200")); bad = 0
for i in range(n):
    if not torch.equal(run(), ref): bad += 1
print("CS3 + DGF runs differing:", bad, "of", n)
