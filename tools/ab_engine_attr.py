"""In-box A/B of one DiTEngine attribute at BASELINE configs[1]'s shape: alternating blocks of denoise steps with the attribute at two
values on ONE engine (graphs are keyed by the attributes that change plans), the minimum and median step time per arm.
    python tools/ab_engine_attr.py ln_lora 1 0 [--rounds 6] [--steps 28] [--operands fp16]"""
import argparse
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from loongx_amd.flux.engine import DiTEngine
from loongx_amd.flux.weights import FluxConfig, synthetic_weights

ap = argparse.ArgumentParser()
ap.add_argument("attr")
ap.add_argument("a", type=int)
ap.add_argument("b", type=int)
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--steps", type=int, default=28)
ap.add_argument("--operands", default="bf16")
a = ap.parse_args()
dev = "cuda"
cfg = FluxConfig()
eng = DiTEngine(synthetic_weights(cfg, dev), dev)
B, T, hw = 1, 512, 32
N = hw * hw
g = torch.Generator(device=dev).manual_seed(0)
lat, cond = torch.randn(B, N, 64, device=dev, generator=g), torch.randn(B, N, 64, device=dev, generator=g)
pe, pooled = torch.randn(B, T, 4096, device=dev, generator=g) * 0.1, torch.randn(B, 768, device=dev, generator=g)
ids = torch.zeros(hw, hw, 3, device=dev)
ids[..., 1] = torch.arange(hw, device=dev)[:, None]
ids[..., 2] = torch.arange(hw, device=dev)[None, :]
img_ids = ids.reshape(-1, 3)
cond_ids = img_ids.clone()
cond_ids[:, 2] -= hw
eng.set_conditioning(pe, pooled, torch.full((B,), 3.5, device=dev), torch.zeros(T, 3, device=dev), img_ids, cond, cond_ids,
                     model_config={"operands": a.operands})
ts = torch.full((B,), 0.5, device=dev)
res = {a.a: [], a.b: []}
outs = {}
for val in (a.a, a.b):                      # warm both plans (eager pass + capture)
    setattr(eng, a.attr, (int if a.attr in ('TL_SPLIT',) else bool)(val))
    for _ in range(3):
        outs[val] = eng.forward(lat, ts).float().clone()
torch.cuda.synchronize()
for r in range(a.rounds):
    for val in ((a.a, a.b) if r % 2 == 0 else (a.b, a.a)):
        setattr(eng, a.attr, (int if a.attr in ('TL_SPLIT',) else bool)(val))
        eng.graphs.clear() if a.attr in ('TL_SPLIT',) else None      # (attributes the graph key does not know)
        for _ in range(2):
            eng.forward(lat, ts)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            eng.forward(lat, ts)
        torch.cuda.synchronize()
        res[val].append((time.perf_counter() - t0) / a.steps * 1e3)
rel = float((outs[a.a] - outs[a.b]).norm() / outs[a.b].norm())
for val in (a.a, a.b):
    print(f"{a.attr}={val}: min {min(res[val]):.3f} ms/step, median {statistics.median(res[val]):.3f}  ({[round(x, 2) for x in res[val]]})")
print(f"median ratio {statistics.median(res[a.a]) / statistics.median(res[a.b]):.4f}, min ratio {min(res[a.a]) / min(res[a.b]):.4f}; outputs differ by {rel:.2e}")
