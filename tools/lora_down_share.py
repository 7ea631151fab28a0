"""What the lx_lora_down launches cost the denoise step on its critical path: the full-shape DiT step timed as shipped and with the
launches removed (ops.lora_down replaced by a no-op before the step graph is captured: the consumer GEMMs read a stale t, the numbers
are WRONG, only the clock is of interest), alternating, same box. An upper bound on what any batching / fusion of those launches
could buy -- each one's input is the output of the kernel in front of it, so they cannot simply be grouped per block."""
import argparse
import time

import torch

from loongx_amd import ops
from loongx_amd.flux.engine import DiTEngine
from loongx_amd.flux.weights import FluxConfig, synthetic_weights

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=28)
ap.add_argument("--rounds", type=int, default=3)
a = ap.parse_args()
dev = "cuda"
cfg = FluxConfig()
w = synthetic_weights(cfg, dev)
B, T, hw = 1, 512, 32
N = hw * hw
g = torch.Generator(device=dev).manual_seed(0)
lat = torch.randn(B, N, 64, device=dev, generator=g)
cond = torch.randn(B, N, 64, device=dev, generator=g)
pe = torch.randn(B, T, 4096, device=dev, generator=g) * 0.1
pooled = torch.randn(B, 768, device=dev, generator=g)
ids = torch.zeros(hw, hw, 3, device=dev)
ids[..., 1] = torch.arange(hw, device=dev)[:, None]
ids[..., 2] = torch.arange(hw, device=dev)[None, :]
img_ids = ids.reshape(-1, 3)
cond_ids = img_ids.clone()
cond_ids[:, 2] -= hw
real = ops.lora_down
calls = [0]


def counted(*args, **kw):
    calls[0] += 1
    return real(*args, **kw)


def build(skip):
    ops.lora_down = (lambda *args, **kw: None) if skip else counted
    eng = DiTEngine(w, dev)
    eng.set_conditioning(pe, pooled, torch.full((B,), 3.5, device=dev), torch.zeros(T, 3, device=dev), img_ids, cond, cond_ids, model_config={})
    ts = torch.full((B,), 0.5, device=dev)
    for _ in range(3):
        eng.forward(lat, ts)
    torch.cuda.synchronize()
    return eng, ts


def timed(eng, ts):
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(a.steps):
        eng.forward(lat, ts)
    torch.cuda.synchronize()
    return (time.time() - t0) / a.steps * 1e3


calls[0] = 0
e0, ts0 = build(False)
per_step = calls[0] / 3
e1, ts1 = build(True)
ops.lora_down = real
print(f"lx_lora_down launches per step: {per_step:.0f}")
for r in range(a.rounds):
    t_a, t_b = timed(e0, ts0), timed(e1, ts1)
    print(f"round {r}: shipped {t_a:.3f} ms / step   without the launches {t_b:.3f} ms / step   delta {t_a - t_b:+.3f} ms ({(t_a - t_b) / t_a * 100:+.2f} %)")
